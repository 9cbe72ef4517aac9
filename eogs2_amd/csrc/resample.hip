// resample.hip — fused virtual-camera resample: uv = cam2virt (u,v,alt), bilinear grid sample (align_corners=True,
// zeros padding) of the virtual render, out-of-view fill; forward and backward.
// Reference semantics: src/gaussiansplatting/gaussian_renderer/renderer_cc_shadow.py:32-50 on top of
// torch.nn.functional.grid_sample (ATen grid_sampler_2d: unnormalize ((x+1)/2)(size-1), floor corner, weights from
// the opposite corner, taps outside the image contribute zero and receive no gradient).
// One lane per output pixel; neighbouring lanes sample neighbouring taps, so the gathers coalesce in L2. HBM-bound.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"

namespace {


struct Taps {
  int x0, y0;           // north-west corner
  float wx1, wy1;       // fractional parts: weight of the east / south neighbours
  bool in_x0, in_x1, in_y0, in_y1;
};

__device__ inline Taps make_taps(float u, float v, int Wv, int Hv) {
  Taps t;
  const float ix = (u + 1.f) * 0.5f * (float)(Wv - 1), iy = (v + 1.f) * 0.5f * (float)(Hv - 1);
  const float fx = floorf(ix), fy = floorf(iy);
  t.wx1 = ix - fx; t.wy1 = iy - fy;
  // clamp before the int conversion: far-away coordinates only need to end up out of bounds
  t.x0 = (int)fminf(fmaxf(fx, -2.f), (float)Wv + 1.f);
  t.y0 = (int)fminf(fmaxf(fy, -2.f), (float)Hv + 1.f);
  t.in_x0 = t.x0 >= 0 && t.x0 < Wv; t.in_x1 = t.x0 + 1 >= 0 && t.x0 + 1 < Wv;
  t.in_y0 = t.y0 >= 0 && t.y0 < Hv; t.in_y1 = t.y0 + 1 >= 0 && t.y0 + 1 < Hv;
  return t;
}

__global__ __launch_bounds__(BLK) void resample_fwd_kernel(int C, int Hv, int Wv, int HW, int n_out,
                                                           const float* __restrict__ vr, const float* __restrict__ uva,
                                                           const float* __restrict__ M, int fill_channel, float fill_value,
                                                           float* __restrict__ sample, float* __restrict__ uv) {
  const int p = blockIdx.x * BLK + threadIdx.x;
  if (p >= HW) return;
  const float a = uva[3 * (size_t)p], b = uva[3 * (size_t)p + 1], c = uva[3 * (size_t)p + 2];
  const float u = M[0] * a + M[1] * b + M[2] * c, v = M[3] * a + M[4] * b + M[5] * c;
  reinterpret_cast<float2*>(uv)[p] = make_float2(u, v);
  const Taps t = make_taps(u, v, Wv, Hv);
  const float wnw = (1.f - t.wx1) * (1.f - t.wy1), wne = t.wx1 * (1.f - t.wy1), wsw = (1.f - t.wx1) * t.wy1, wse = t.wx1 * t.wy1;
  const size_t plane = (size_t)Hv * Wv;
  const size_t o00 = (size_t)t.y0 * Wv + t.x0;
  const bool outside = fabsf(u) > 1.f || fabsf(v) > 1.f;
  for (int ch = 0; ch < n_out; ch++) {
    const float* src = vr + ch * plane;
    float s = 0.f;
    if (t.in_y0 && t.in_x0) s += src[o00] * wnw;
    if (t.in_y0 && t.in_x1) s += src[o00 + 1] * wne;
    if (t.in_y1 && t.in_x0) s += src[o00 + Wv] * wsw;
    if (t.in_y1 && t.in_x1) s += src[o00 + Wv + 1] * wse;
    if (ch == fill_channel && outside) s = fill_value;
    sample[(size_t)ch * HW + p] = s;
  }
}

__global__ __launch_bounds__(BLK) void resample_bwd_kernel(int C, int Hv, int Wv, int HW, int n_out,
                                                           const float* __restrict__ vr, const float* __restrict__ uva,
                                                           const float* __restrict__ M, int fill_channel,
                                                           const float* __restrict__ gs, const float* __restrict__ guv,
                                                           float* __restrict__ gvr, float* __restrict__ guva) {
  const int p = blockIdx.x * BLK + threadIdx.x;
  if (p >= HW) return;
  const float a = uva[3 * (size_t)p], b = uva[3 * (size_t)p + 1], c = uva[3 * (size_t)p + 2];
  const float u = M[0] * a + M[1] * b + M[2] * c, v = M[3] * a + M[4] * b + M[5] * c;
  const Taps t = make_taps(u, v, Wv, Hv);
  const float wx0 = 1.f - t.wx1, wy0 = 1.f - t.wy1;
  const size_t plane = (size_t)Hv * Wv;
  const size_t o00 = (size_t)t.y0 * Wv + t.x0;
  const bool outside = fabsf(u) > 1.f || fabsf(v) > 1.f;
  float gix = 0.f, giy = 0.f;
  for (int ch = 0; ch < n_out; ch++) {
    float g = gs[(size_t)ch * HW + p];
    if (ch == fill_channel && outside) g = 0.f;  // the value was overwritten by a constant
    const float* src = vr + ch * plane;
    float* dst = gvr + ch * plane;
    if (t.in_y0 && t.in_x0) {
      const float val = src[o00];
      unsafeAtomicAdd(dst + o00, g * wx0 * wy0);
      gix -= val * wy0 * g; giy -= val * wx0 * g;
    }
    if (t.in_y0 && t.in_x1) {
      const float val = src[o00 + 1];
      unsafeAtomicAdd(dst + o00 + 1, g * t.wx1 * wy0);
      gix += val * wy0 * g; giy -= val * t.wx1 * g;
    }
    if (t.in_y1 && t.in_x0) {
      const float val = src[o00 + Wv];
      unsafeAtomicAdd(dst + o00 + Wv, g * wx0 * t.wy1);
      gix -= val * t.wy1 * g; giy += val * wx0 * g;
    }
    if (t.in_y1 && t.in_x1) {
      const float val = src[o00 + Wv + 1];
      unsafeAtomicAdd(dst + o00 + Wv + 1, g * t.wx1 * t.wy1);
      gix += val * t.wy1 * g; giy += val * t.wx1 * g;
    }
  }
  // d(ix)/du = (Wv-1)/2 (align_corners=True)
  float gu = gix * (0.5f * (float)(Wv - 1)), gv = giy * (0.5f * (float)(Hv - 1));
  if (guv) {
    const float2 e = reinterpret_cast<const float2*>(guv)[p];
    gu += e.x; gv += e.y;
  }
  guva[3 * (size_t)p] = M[0] * gu + M[3] * gv;
  guva[3 * (size_t)p + 1] = M[1] * gu + M[4] * gv;
  guva[3 * (size_t)p + 2] = M[2] * gu + M[5] * gv;
}

}  // namespace

void launch_resample_fwd(int C, int Hv, int Wv, int H, int W, int n_out, const float* vr, const float* uva,
                         const float* M, int fill_channel, float fill_value, float* sample, float* uv, hipStream_t s) {
  const int HW = H * W;
  hipLaunchKernelGGL(resample_fwd_kernel, dim3((HW + BLK - 1) / BLK), dim3(BLK), 0, s, C, Hv, Wv, HW, n_out, vr, uva, M,
                     fill_channel, fill_value, sample, uv);
}

void launch_resample_bwd(int C, int Hv, int Wv, int H, int W, int n_out, const float* vr, const float* uva,
                         const float* M, int fill_channel, const float* gs, const float* guv, float* gvr, float* guva,
                         hipStream_t s) {
  const int HW = H * W;
  (void)hipMemsetAsync(gvr, 0, (size_t)C * Hv * Wv * sizeof(float), s);
  hipLaunchKernelGGL(resample_bwd_kernel, dim3((HW + BLK - 1) / BLK), dim3(BLK), 0, s, C, Hv, Wv, HW, n_out, vr, uva, M,
                     fill_channel, gs, guv, gvr, guva);
}
