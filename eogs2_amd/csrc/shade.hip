// shade.hip — the per-pixel image chain between the raw render and the scalar losses (include/eogs_shade.h, SURVEY.md
// §8 row f2): camera render pipeline (colour correction + shadow), masked resample losses, translucent-shadow regulariser.
// Reference semantics: scene/cameras/affine_cameras.py:33-40,303-348; loss/shadow.py:7-17,37-51; loss/main_loss.py:83-96,
// 151-164 (all under src/gaussiansplatting/).
//
// All of it is elementwise work over H x W planes plus a handful of sums: HBM-bound by construction. One lane per pixel in
// a grid-stride loop (dword-coalesced planar loads), nothing is saved between forward and backward (the few values
// backward needs are recomputed from the inputs), and every sum is reduced per workgroup and then by one small kernel in
// a fixed order — no atomics, bitwise reproducible.
#include "common.h"

namespace {

constexpr int ST = 256;        // threads per workgroup
constexpr int SMAXBLK = 1024;  // workgroups per launch (4 per CU); partial sums live in the caller's workspace
constexpr int SK = 16;         // floats per workgroup partial (15 used by shade_bwd)

__device__ inline float wg_sum(float v, float* s_red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[w] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

inline int shade_blocks(int64_t n) {
  const int64_t b = (n + ST - 1) / ST;
  return (int)(b < 1 ? 1 : (b > SMAXBLK ? SMAXBLK : b));
}

// sums the per-workgroup partials [nblk][SK] column by column in a fixed order: out[k] = sum_b partial[b][k], k < K
template <int MODE>  // 0: plain sums (shade_bwd); 1: mloss {S_alt/N, S_rgb/N, N}; 2: tshadow -S/n
__global__ __launch_bounds__(ST) void shade_reduce_kernel(const float* __restrict__ partial, int nblk, int K, float scale,
                                                          float* __restrict__ out) {
  __shared__ float s_red[4];
  float tot[SK];
  for (int k = 0; k < K; k++) {
    float a = 0.f;
    for (int i = threadIdx.x; i < nblk; i += ST) a += partial[(size_t)i * SK + k];
    tot[k] = wg_sum(a, s_red);
  }
  if (threadIdx.x != 0) return;
  if (MODE == 0) {
    for (int k = 0; k < K; k++) out[k] = tot[k];
  } else if (MODE == 1) {
    const float n = tot[2];
    out[0] = n > 0.f ? tot[0] / n : 0.f;
    out[1] = n > 0.f ? tot[1] / n : 0.f;
    out[2] = n;
  } else {
    out[0] = -tot[0] * scale;
  }
}

// ---- camera render pipeline ----------------------------------------------------------------------------------------
struct ShadeParams {
  float M[12];
  float ins[3];
};

__global__ __launch_bounds__(ST) void shade_fwd_kernel(int64_t n, const float* __restrict__ raw, const float* __restrict__ alt_diff,
                                                       const float* __restrict__ Mp, const float* __restrict__ insp,
                                                       float* __restrict__ cc, float* __restrict__ shaded,
                                                       float* __restrict__ shadow) {
  float M[12], ins[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 12; i++) M[i] = Mp[i];
  if (alt_diff) {
#pragma unroll
    for (int i = 0; i < 3; i++) ins[i] = insp[i];
  }
  for (int64_t p = (int64_t)blockIdx.x * ST + threadIdx.x; p < n; p += (int64_t)gridDim.x * ST) {
    const float r0 = raw[p], r1 = raw[n + p], r2 = raw[2 * n + p];
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] = M[4 * k] * r0 + M[4 * k + 1] * r1 + M[4 * k + 2] * r2 + M[4 * k + 3];
    if (cc) {
#pragma unroll
      for (int k = 0; k < 3; k++) cc[k * n + p] = c[k];
    }
    if (alt_diff) {
      const float s = expf(0.4f * fminf(alt_diff[p], 0.f));
      shadow[p] = s;
#pragma unroll
      for (int k = 0; k < 3; k++) shaded[k * n + p] = s * c[k] + ((1.f - s) * ins[k]) * c[k];
    } else {
#pragma unroll
      for (int k = 0; k < 3; k++) shaded[k * n + p] = c[k];
    }
  }
}

// shaded_c = cc_c f_c, f_c = s + (1 - s) ins_c:
//   d/dcc_c = f_c g_c (+ g_cc_c);  d/ds = sum_c g_c cc_c (1 - ins_c) (+ g_shadow);  d/dins_c = sum_p g_c (1 - s) cc_c
//   d/dalt_diff = d/ds * 0.4 s [alt_diff <= 0]   (torch.clamp(max=0) passes the gradient at equality)
//   d/draw_k = sum_c M[c][k] d/dcc_c;  d/dM[c][k] = sum_p d/dcc_c raw_k;  d/dM[c][3] = sum_p d/dcc_c
__global__ __launch_bounds__(ST) void shade_bwd_kernel(int64_t n, const float* __restrict__ raw, const float* __restrict__ alt_diff,
                                                       const float* __restrict__ Mp, const float* __restrict__ insp,
                                                       const float* __restrict__ g_shaded, const float* __restrict__ g_cc,
                                                       const float* __restrict__ g_shadow, float* __restrict__ g_raw,
                                                       float* __restrict__ g_alt, float* __restrict__ partial) {
  __shared__ float s_red[4];
  float M[12], ins[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 12; i++) M[i] = Mp[i];
  if (alt_diff) {
#pragma unroll
    for (int i = 0; i < 3; i++) ins[i] = insp[i];
  }
  float acc[15];
#pragma unroll
  for (int i = 0; i < 15; i++) acc[i] = 0.f;
  for (int64_t p = (int64_t)blockIdx.x * ST + threadIdx.x; p < n; p += (int64_t)gridDim.x * ST) {
    const float r[3] = {raw[p], raw[n + p], raw[2 * n + p]};
    float c[3], g[3], gcc[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      c[k] = M[4 * k] * r[0] + M[4 * k + 1] * r[1] + M[4 * k + 2] * r[2] + M[4 * k + 3];
      g[k] = g_shaded[k * n + p];
    }
    if (alt_diff) {
      const float d = alt_diff[p];
      const float s = expf(0.4f * fminf(d, 0.f));
      float gs = g_shadow ? g_shadow[p] : 0.f;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        gcc[k] = g[k] * (s + (1.f - s) * ins[k]);
        gs += g[k] * c[k] * (1.f - ins[k]);
        acc[12 + k] += g[k] * (1.f - s) * c[k];
      }
      g_alt[p] = d <= 0.f ? gs * 0.4f * s : 0.f;
    } else {
#pragma unroll
      for (int k = 0; k < 3; k++) gcc[k] = g[k];
    }
    if (g_cc) {
#pragma unroll
      for (int k = 0; k < 3; k++) gcc[k] += g_cc[k * n + p];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
      g_raw[k * n + p] = M[k] * gcc[0] + M[4 + k] * gcc[1] + M[8 + k] * gcc[2];
      acc[4 * k] += gcc[k] * r[0];
      acc[4 * k + 1] += gcc[k] * r[1];
      acc[4 * k + 2] += gcc[k] * r[2];
      acc[4 * k + 3] += gcc[k];
    }
  }
#pragma unroll
  for (int i = 0; i < 15; i++) {
    const float t = wg_sum(acc[i], s_red);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.x * SK + i] = t;
  }
}

// ---- masked resample losses ----------------------------------------------------------------------------------------
__device__ inline bool mloss_mask(int mode, float d, float2 uv) {
  const bool alt_ok = mode == EOGS_MLOSS_SUN ? d > -1e-2f : fabsf(d) < 0.30f;
  return alt_ok && fabsf(uv.x) < 1.f && fabsf(uv.y) < 1.f;
}

__global__ __launch_bounds__(ST) void mloss_fwd_kernel(int64_t n, int mode, const float* __restrict__ alt_diff,
                                                       const float* __restrict__ a, const float* __restrict__ b,
                                                       const float2* __restrict__ uv, float* __restrict__ partial) {
  __shared__ float s_red[4];
  float s_alt = 0.f, s_rgb = 0.f, cnt = 0.f;
  for (int64_t p = (int64_t)blockIdx.x * ST + threadIdx.x; p < n; p += (int64_t)gridDim.x * ST) {
    const float d = alt_diff[p];
    if (mloss_mask(mode, d, uv[p])) {
      s_alt += fabsf(d);
      s_rgb += fabsf(a[p] - b[p]) + fabsf(a[n + p] - b[n + p]) + fabsf(a[2 * n + p] - b[2 * n + p]);
      cnt += 1.f;  // at most 2^24 / gridDim.x per lane: exact
    }
  }
  const float t0 = wg_sum(s_alt, s_red), t1 = wg_sum(s_rgb, s_red), t2 = wg_sum(cnt, s_red);
  if (threadIdx.x == 0) {
    partial[(size_t)blockIdx.x * SK] = t0;
    partial[(size_t)blockIdx.x * SK + 1] = t1;
    partial[(size_t)blockIdx.x * SK + 2] = t2;
  }
}

__device__ inline float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(ST) void mloss_bwd_kernel(int64_t n, int mode, const float* __restrict__ alt_diff,
                                                       const float* __restrict__ a, const float* __restrict__ b,
                                                       const float2* __restrict__ uv, const float* __restrict__ out,
                                                       const float* __restrict__ upstream, float* __restrict__ g_alt,
                                                       float* __restrict__ g_a, float* __restrict__ g_b) {
  const float cnt = out[2];
  const float w_alt = cnt > 0.f ? upstream[0] / cnt : 0.f, w_rgb = cnt > 0.f ? upstream[1] / cnt : 0.f;
  for (int64_t p = (int64_t)blockIdx.x * ST + threadIdx.x; p < n; p += (int64_t)gridDim.x * ST) {
    const float d = alt_diff[p];
    const bool m = mloss_mask(mode, d, uv[p]);
    g_alt[p] = m ? w_alt * sgn(d) : 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float g = m ? w_rgb * sgn(a[k * n + p] - b[k * n + p]) : 0.f;
      g_a[k * n + p] = g;
      if (g_b) g_b[k * n + p] = -g;
    }
  }
}

// ---- translucent-shadow regulariser ------------------------------------------------------------------------------
__global__ __launch_bounds__(ST) void tshadow_fwd_kernel(int64_t n, const float* __restrict__ a, float* __restrict__ partial) {
  __shared__ float s_red[4];
  float s = 0.f;
  for (int64_t p = (int64_t)blockIdx.x * ST + threadIdx.x; p < n; p += (int64_t)gridDim.x * ST) {
    const float x = a[p];
    const float b = fminf(fmaxf(x, 0.05f), 0.95f);
    s += x * log2f(b) + (1.f - x) * log2f(1.f - b);
  }
  const float t = wg_sum(s, s_red);
  if (threadIdx.x == 0) partial[(size_t)blockIdx.x * SK] = t;
}

__global__ __launch_bounds__(ST) void tshadow_bwd_kernel(int64_t n, const float* __restrict__ a, const float* __restrict__ upstream,
                                                         float* __restrict__ g_a) {
  const float w = -upstream[0] / (float)n;
  const float inv_ln2 = 1.4426950408889634f;
  for (int64_t p = (int64_t)blockIdx.x * ST + threadIdx.x; p < n; p += (int64_t)gridDim.x * ST) {
    const float x = a[p];
    const float b = fminf(fmaxf(x, 0.05f), 0.95f);
    float d = log2f(b) - log2f(1.f - b);
    if (x >= 0.05f && x <= 0.95f) d += (x / b - (1.f - x) / (1.f - b)) * inv_ln2;
    g_a[p] = w * d;
  }
}

}  // namespace

size_t shade_ws_bytes() { return (size_t)SMAXBLK * SK * sizeof(float) + 256; }

void launch_shade_fwd(int H, int W, const float* raw, const float* alt_diff, const float* M, const float* inshadow, float* cc,
                      float* shaded, float* shadow, hipStream_t s) {
  const int64_t n = (int64_t)H * W;
  hipLaunchKernelGGL(shade_fwd_kernel, dim3(shade_blocks(n)), dim3(ST), 0, s, n, raw, alt_diff, M, inshadow, cc, shaded, shadow);
}

void launch_shade_bwd(int H, int W, const float* raw, const float* alt_diff, const float* M, const float* inshadow,
                      const float* g_shaded, const float* g_cc, const float* g_shadow, float* g_raw, float* g_alt,
                      float* g_params, void* ws, hipStream_t s) {
  const int64_t n = (int64_t)H * W;
  const int nb = shade_blocks(n);
  float* partial = reinterpret_cast<float*>(ws_base(ws));
  hipLaunchKernelGGL(shade_bwd_kernel, dim3(nb), dim3(ST), 0, s, n, raw, alt_diff, M, inshadow, g_shaded, g_cc, g_shadow, g_raw,
                     g_alt, partial);
  hipLaunchKernelGGL(shade_reduce_kernel<0>, dim3(1), dim3(ST), 0, s, partial, nb, 15, 1.f, g_params);
}

void launch_mloss_fwd(int H, int W, int mode, const float* alt_diff, const float* a, const float* b, const float* uv, float* out,
                      void* ws, hipStream_t s) {
  const int64_t n = (int64_t)H * W;
  const int nb = shade_blocks(n);
  float* partial = reinterpret_cast<float*>(ws_base(ws));
  hipLaunchKernelGGL(mloss_fwd_kernel, dim3(nb), dim3(ST), 0, s, n, mode, alt_diff, a, b, reinterpret_cast<const float2*>(uv), partial);
  hipLaunchKernelGGL(shade_reduce_kernel<1>, dim3(1), dim3(ST), 0, s, partial, nb, 3, 1.f, out);
}

void launch_mloss_bwd(int H, int W, int mode, const float* alt_diff, const float* a, const float* b, const float* uv,
                      const float* out, const float* upstream, float* g_alt, float* g_a, float* g_b, hipStream_t s) {
  const int64_t n = (int64_t)H * W;
  hipLaunchKernelGGL(mloss_bwd_kernel, dim3(shade_blocks(n)), dim3(ST), 0, s, n, mode, alt_diff, a, b,
                     reinterpret_cast<const float2*>(uv), out, upstream, g_alt, g_a, g_b);
}

void launch_tshadow_fwd(int64_t n, const float* a, float* out, void* ws, hipStream_t s) {
  const int nb = shade_blocks(n);
  float* partial = reinterpret_cast<float*>(ws_base(ws));
  hipLaunchKernelGGL(tshadow_fwd_kernel, dim3(nb), dim3(ST), 0, s, n, a, partial);
  hipLaunchKernelGGL(shade_reduce_kernel<2>, dim3(1), dim3(ST), 0, s, partial, nb, 1, 1.f / (float)n, out);
}

void launch_tshadow_bwd(int64_t n, const float* a, const float* upstream, float* g_a, hipStream_t s) {
  hipLaunchKernelGGL(tshadow_bwd_kernel, dim3(shade_blocks(n)), dim3(ST), 0, s, n, a, upstream, g_a);
}
