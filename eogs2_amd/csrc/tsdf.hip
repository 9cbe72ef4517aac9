// tsdf.hip — TSDF integration of one altitude image into the voxel volume (include/eogs_tsdf.h).
// Reference semantics: src/gaussiansplatting/tsdf.py:325-368 (sample_sdf), :459-520 (integrate, update_tsdf).
//
// One lane per voxel, z fastest (the volumes' contiguous axis): a wave covers 64 consecutive altitudes of one (x, y)
// column pair, which project to almost the same pixel under the near-nadir affine cameras, so the four bilinear taps are
// served by L1/L2; HBM traffic is the 8 B/voxel read + 8 B/voxel conditional write of the two volumes.
#include "common.h"

namespace {

struct TsdfAffine {
  float A[9], b[3], Ai[9], Aib[3];
};

__device__ inline float tap(const float* __restrict__ img, int H, int W, int x, int y) {
  return (x >= 0 && x < W && y >= 0 && y < H) ? img[(size_t)y * W + x] : 0.f;  // grid_sample padding_mode="zeros"
}

__global__ __launch_bounds__(256) void tsdf_integrate_kernel(int nx, int ny, int nz, const float* __restrict__ ax,
                                                             const float* __restrict__ ay, const float* __restrict__ az,
                                                             const float* __restrict__ affine, float scale, float trunc,
                                                             int H, int W, const float* __restrict__ alt,
                                                             const float* __restrict__ wgt, float* __restrict__ tsdf,
                                                             float* __restrict__ wvol) {
  const size_t n = (size_t)nx * ny * nz;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int iz = (int)(i % nz), iy = (int)((i / nz) % ny), ix = (int)(i / ((size_t)nz * ny));
  const float* A = affine;
  const float* b = affine + 9;
  const float* Ai = affine + 12;
  const float* Aib = affine + 21;
  const float p[3] = {ax[ix] / scale, ay[iy] / scale, az[iz] / scale};
  float v[3];
#pragma unroll
  for (int c = 0; c < 3; c++) v[c] = A[3 * c] * p[0] + A[3 * c + 1] * p[1] + A[3 * c + 2] * p[2] + b[c];
  // F.grid_sample(..., mode="bilinear", align_corners=True)
  const float fx = (v[0] + 1.f) * 0.5f * (float)(W - 1), fy = (v[1] + 1.f) * 0.5f * (float)(H - 1);
  const float x0f = floorf(fx), y0f = floorf(fy);
  const float tx = fx - x0f, ty = fy - y0f;
  float a_s = 0.f, w_s = 0.f;
  if (fx > -2.f && fy > -2.f && fx < (float)W + 1.f && fy < (float)H + 1.f) {  // int conversion is safe
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
    a_s = tap(alt, H, W, x0, y0) * w00 + tap(alt, H, W, x0 + 1, y0) * w01 + tap(alt, H, W, x0, y0 + 1) * w10 +
          tap(alt, H, W, x0 + 1, y0 + 1) * w11;
    w_s = tap(wgt, H, W, x0, y0) * w00 + tap(wgt, H, W, x0 + 1, y0) * w01 + tap(wgt, H, W, x0, y0 + 1) * w10 +
          tap(wgt, H, W, x0 + 1, y0 + 1) * w11;
  }
  const bool valid = fabsf(v[0]) <= 1.f && fabsf(v[1]) <= 1.f;
  const float vn[3] = {v[0], v[1], a_s};
  float d2 = 0.f;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float q = Ai[3 * c] * vn[0] + Ai[3 * c + 1] * vn[1] + Ai[3 * c + 2] * vn[2] - Aib[c];
    d2 += (q - p[c]) * (q - p[c]);
  }
  const float dz = v[2] - a_s;
  const float sgn = dz > 0.f ? 1.f : (dz < 0.f ? -1.f : 0.f);
  const float sdf = sqrtf(d2) * sgn * scale;
  if (valid && sdf >= -trunc) {
    const float t_new = fminf(1.f, sdf / trunc);
    const float w_old = wvol[i], t_old = tsdf[i];
    const float w_new = w_old + w_s;
    tsdf[i] = (w_old * t_old + w_s * t_new) / w_new;
    wvol[i] = w_new;
  }
}

}  // namespace

void launch_tsdf_integrate(int nx, int ny, int nz, const float* ax, const float* ay, const float* az, const float* affine,
                           float scale, float trunc, int H, int W, const float* alt, const float* wgt, float* tsdf,
                           float* wvol, hipStream_t s) {
  const size_t n = (size_t)nx * ny * nz;
  hipLaunchKernelGGL(tsdf_integrate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, nx, ny, nz, ax, ay, az, affine,
                     scale, trunc, H, W, alt, wgt, tsdf, wvol);
}
