// loss.hip — fused photometric loss: mean|x-y| and mean SSIM(x,y) with an 11x11 Gaussian window, forward and backward
// (SURVEY.md §8 row f2). Reference semantics: src/gaussiansplatting/utils/loss_utils.py:18-19 (l1_loss), :26-42 (window),
// :45-85 (ssim/_ssim: five depthwise conv2d with zero padding 5, C1 = 0.01^2, C2 = 0.03^2), image_utils.py:27-28 (lphotom).
//
// One workgroup = one 32x32 output tile of one plane. The 42x42 input patch (zero outside the image, like conv2d's
// padding) is staged in LDS once; the separable window runs as a horizontal pass LDS->LDS and a vertical pass
// LDS->registers, so each input pixel is fetched from HBM/L2 once per tile instead of 121 times. Forward keeps the three
// partial-derivative maps dS/dmu1, dS/dE[x^2], dS/dE[xy]; backward is the same separable window applied to those maps
// (the zero-padded correlation with a symmetric window is its own adjoint):
//     dS_sum/dx(p) = W*Dm (p) + 2 x(p) W*D11 (p) + y(p) W*D12 (p)
// HBM-bound by design: forward reads 8 and writes 12 B/pixel, backward reads 20 and writes 4 B/pixel.
// Sums are reduced per workgroup, then in a fixed order by one small kernel: bitwise reproducible, no atomics.
#include "common.h"

// The library is built with -ffp-contract=off for the rasterizer's sake (its parity is against an fp32 restatement of the
// reference's arithmetic). The loss has no such twin — its oracle is float64, its tolerance relative — and its window sums are
// half of its instructions: multiply-adds may fuse here.
#pragma clang fp contract(fast)

namespace {

constexpr int LT = 32;             // output tile width
#ifndef LOSS_LTY
#define LOSS_LTY 16
#endif
constexpr int LTY = LOSS_LTY;      // output tile height (round 4: 16 rows — 26 KB of LDS, six workgroups per CU instead of three)
constexpr int HALO = LOSS_WIN / 2; // 5
constexpr int LW = LT + 2 * HALO;  // 42 staged columns
constexpr int LH = LTY + 2 * HALO; // staged rows
constexpr int RPT = LTY / 8;       // adjacent output rows per thread (32 x 8 threads)
constexpr int HG = 4;              // adjacent output columns per thread in the horizontal pass
static_assert(LTY % 8 == 0 && LT % HG == 0 && LH * (LT / HG) <= 256, "tile shape against 256 threads");
constexpr int LTHREADS = 256;

__device__ inline float block_sum(float v, float* s_red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[w] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// ---------------------------------------------------------------------------------------------------------------
template <bool WITH_L1, bool WITH_SSIM>
__global__ __launch_bounds__(LTHREADS) void loss_fwd_kernel(int H, int W, const float* __restrict__ img,
                                                            const float* __restrict__ gt, LossWindow win,
                                                            float* __restrict__ maps, size_t map_stride,
                                                            float* __restrict__ partial) {
  __shared__ float s_x[LH][LW + 1];
  __shared__ float s_y[LH][LW + 1];
  __shared__ float s_h[WITH_SSIM ? 5 : 1][WITH_SSIM ? LH : 1][LT + 1];
  __shared__ float s_red[4];
  const int t = threadIdx.x;
  const int plane = blockIdx.z;
  const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LTY;
  const size_t pbase = (size_t)plane * H * W;

  {
    // all of the patch's loads in flight before the first one is stored (seven dependent round trips otherwise)
    constexpr int NE = (LH * LW + LTHREADS - 1) / LTHREADS;
    float a[NE], b[NE];
#pragma unroll
    for (int i = 0; i < NE; i++) {
      const int e = t + i * LTHREADS, r = e / LW, c = e - r * LW;
      const int gy = y0 + r - HALO, gx = x0 + c - HALO;
      a[i] = 0.f; b[i] = 0.f;
      if (e < LH * LW && gy >= 0 && gy < H && gx >= 0 && gx < W) {
        a[i] = img[pbase + (size_t)gy * W + gx];
        b[i] = gt[pbase + (size_t)gy * W + gx];
      }
    }
#pragma unroll
    for (int i = 0; i < NE; i++) {
      const int e = t + i * LTHREADS, r = e / LW, c = e - r * LW;
      if (e < LH * LW) { s_x[r][c] = a[i]; s_y[r][c] = b[i]; }
    }
  }
  __syncthreads();

  if (WITH_SSIM) {
    // horizontal pass: 42 rows x 32 columns x 5 moments. A thread takes EIGHT adjacent columns of one row: their windows
    // overlap, so it reads 18 + 18 staged values once instead of 8 x (11 + 11) (round 4: the kernel was bound by its LDS
    // reads — 86 k dwords per tile, 28 k now; every output's own sum keeps its order, k ascending)
    if (t < LH * (LT / HG)) {
      const int r = t / (LT / HG), c0 = (t % (LT / HG)) * HG;
      float xa[HG + LOSS_WIN - 1], ya[HG + LOSS_WIN - 1];
#pragma unroll
      for (int k = 0; k < HG + LOSS_WIN - 1; k++) { xa[k] = s_x[r][c0 + k]; ya[k] = s_y[r][c0 + k]; }
#pragma unroll
      for (int o = 0; o < HG; o++) {
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < LOSS_WIN; k++) {
          const float a = xa[o + k], b = ya[o + k], w = win.w[k];
          const float wa = w * a, wb = w * b;
          m1 += wa; m2 += wb; e11 += wa * a; e22 += wb * b; e12 += wa * b;
        }
        const int c = c0 + o;
        s_h[0][r][c] = m1; s_h[1][r][c] = m2; s_h[2][r][c] = e11; s_h[3][r][c] = e22; s_h[4][r][c] = e12;
      }
    }
    __syncthreads();
  }

  const int tx = t & (LT - 1), ty = t >> 5;  // 32 x 8 threads, four ADJACENT rows each: 14 column values serve four windows
  float l1 = 0.f, ss = 0.f;
  float vs[WITH_SSIM ? 5 : 1][RPT];
  if (WITH_SSIM) {
#pragma unroll
    for (int m = 0; m < 5; m++) {
      float col[RPT + LOSS_WIN - 1];
#pragma unroll
      for (int i = 0; i < RPT + LOSS_WIN - 1; i++) col[i] = s_h[m][RPT * ty + i][tx];
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < LOSS_WIN; k++) acc += win.w[k] * col[j + k];
        vs[m][j] = acc;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < RPT; j++) {
    const int r = RPT * ty + j;
    const int gy = y0 + r, gx = x0 + tx;
    const bool inside = gy < H && gx < W;
    const float a = s_x[r + HALO][tx + HALO], b = s_y[r + HALO][tx + HALO];
    if (WITH_L1 && inside) l1 += fabsf(a - b);
    if (WITH_SSIM) {
      const float mu1 = vs[0][j], mu2 = vs[1][j], e11 = vs[2][j], e22 = vs[3][j], e12 = vs[4][j];
      // _ssim (loss_utils.py:58-80)
      const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
      const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
      const float s11 = e11 - mu1_sq, s22 = e22 - mu2_sq, s12 = e12 - mu12;
      const float A1 = 2.f * mu12 + C1, A2 = 2.f * s12 + C2;
      const float B1 = mu1_sq + mu2_sq + C1, B2 = s11 + s22 + C2;
      const float inv = 1.f / (B1 * B2);
      const float S = A1 * A2 * inv;
      if (inside) {
        ss += S;
        // derivatives with respect to the three window sums that depend on x: mu1, E[x^2], E[xy]
        const float d11 = -S / B2;         // dS/dsigma1_sq
        const float d12 = 2.f * A1 * inv;  // dS/dsigma12
        const float dm = 2.f * mu2 * A2 * inv - 2.f * mu1 * S / B1 - 2.f * mu1 * d11 - mu2 * d12;
        const size_t o = pbase + (size_t)gy * W + gx;
        maps[o] = dm;
        maps[map_stride + o] = d11;
        maps[2 * map_stride + o] = d12;
      }
    }
  }
  const size_t blk = ((size_t)plane * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  const float l1s = WITH_L1 ? block_sum(l1, s_red) : 0.f;
  const float sss = WITH_SSIM ? block_sum(ss, s_red) : 0.f;
  if (t == 0) {
    partial[2 * blk] = l1s;
    partial[2 * blk + 1] = sss;
  }
}

// one workgroup per plane sums that plane's tile partials in a fixed order; the last step (plane 0's workgroup is not
// special: a second launch) combines the planes
__global__ __launch_bounds__(LTHREADS) void loss_plane_reduce_kernel(const float* __restrict__ partial, int tiles,
                                                                     float* __restrict__ plane_tmp) {
  __shared__ float s_red[4];
  const int plane = blockIdx.x;
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < tiles; i += LTHREADS) {
    a += partial[2 * ((size_t)plane * tiles + i)];
    b += partial[2 * ((size_t)plane * tiles + i) + 1];
  }
  a = block_sum(a, s_red);
  b = block_sum(b, s_red);
  if (threadIdx.x == 0) {
    plane_tmp[2 * plane] = a;
    plane_tmp[2 * plane + 1] = b;
  }
}

__global__ __launch_bounds__(64) void loss_finalize_kernel(const float* __restrict__ plane_tmp, int planes, float inv_n,
                                                           float w_l1, float w_ssim, float bias,
                                                           float* __restrict__ out, float* __restrict__ plane_sums) {
  if (threadIdx.x != 0) return;
  float a = 0.f, b = 0.f;
  for (int p = 0; p < planes; p++) {
    a += plane_tmp[2 * p];
    b += plane_tmp[2 * p + 1];
    if (plane_sums) {
      plane_sums[2 * p] = plane_tmp[2 * p];
      plane_sums[2 * p + 1] = plane_tmp[2 * p + 1];
    }
  }
  const float l1m = a * inv_n, sm = b * inv_n;
  out[0] = w_l1 * l1m + w_ssim * sm + bias;
  out[1] = l1m;
  out[2] = sm;
}

// ---------------------------------------------------------------------------------------------------------------
template <bool WITH_L1, bool WITH_SSIM>
__global__ __launch_bounds__(LTHREADS) void loss_bwd_kernel(int H, int W, const float* __restrict__ img,
                                                            const float* __restrict__ gt, LossWindow win,
                                                            const float* __restrict__ maps, size_t map_stride,
                                                            float w_l1, float w_ssim, float inv_n,
                                                            const float* __restrict__ upstream,
                                                            const float* __restrict__ plane_grad,
                                                            float* __restrict__ dimg) {
  __shared__ float s_m[WITH_SSIM ? 3 : 1][WITH_SSIM ? LH : 1][LW + 1];
  __shared__ float s_h[WITH_SSIM ? 3 : 1][WITH_SSIM ? LH : 1][LT + 1];
  const int t = threadIdx.x;
  const int plane = blockIdx.z;
  const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LTY;
  const size_t pbase = (size_t)plane * H * W;
  // weights of sum|x-y| and sum SSIM of this plane in the loss
  float g_l1, g_ss;
  if (plane_grad) {
    g_l1 = plane_grad[2 * plane];
    g_ss = plane_grad[2 * plane + 1];
  } else {
    // out[0] = w_l1*l1_mean + w_ssim*ssim_mean + bias, out[1] = l1_mean, out[2] = ssim_mean
    const float u0 = upstream ? upstream[0] : 1.f, u1 = upstream ? upstream[1] : 0.f, u2 = upstream ? upstream[2] : 0.f;
    g_l1 = (u0 * w_l1 + u1) * inv_n;
    g_ss = (u0 * w_ssim + u2) * inv_n;
  }

  if (WITH_SSIM) {
    {
      constexpr int NE = (LH * LW + LTHREADS - 1) / LTHREADS;
      float a[NE], b[NE], d[NE];
#pragma unroll
      for (int i = 0; i < NE; i++) {
        const int e = t + i * LTHREADS, r = e / LW, c = e - r * LW;
        const int gy = y0 + r - HALO, gx = x0 + c - HALO;
        a[i] = 0.f; b[i] = 0.f; d[i] = 0.f;
        if (e < LH * LW && gy >= 0 && gy < H && gx >= 0 && gx < W) {
          const size_t o = pbase + (size_t)gy * W + gx;
          a[i] = maps[o];
          b[i] = maps[map_stride + o];
          d[i] = maps[2 * map_stride + o];
        }
      }
#pragma unroll
      for (int i = 0; i < NE; i++) {
        const int e = t + i * LTHREADS, r = e / LW, c = e - r * LW;
        if (e < LH * LW) { s_m[0][r][c] = a[i]; s_m[1][r][c] = b[i]; s_m[2][r][c] = d[i]; }
      }
    }
    __syncthreads();
    // (eight adjacent columns per thread, as in the forward)
    if (t < LH * (LT / HG)) {
      const int r = t / (LT / HG), c0 = (t % (LT / HG)) * HG;
#pragma unroll
      for (int m = 0; m < 3; m++) {
        float v[HG + LOSS_WIN - 1];
#pragma unroll
        for (int k = 0; k < HG + LOSS_WIN - 1; k++) v[k] = s_m[m][r][c0 + k];
#pragma unroll
        for (int o = 0; o < HG; o++) {
          float a = 0.f;
#pragma unroll
          for (int k = 0; k < LOSS_WIN; k++) a += win.w[k] * v[o + k];
          s_h[m][r][c0 + o] = a;
        }
      }
    }
    __syncthreads();
  }

  const int tx = t & (LT - 1), ty = t >> 5;  // four adjacent rows per thread
  float vs[WITH_SSIM ? 3 : 1][RPT];
  if (WITH_SSIM) {
#pragma unroll
    for (int m = 0; m < 3; m++) {
      float col[RPT + LOSS_WIN - 1];
#pragma unroll
      for (int i = 0; i < RPT + LOSS_WIN - 1; i++) col[i] = s_h[m][RPT * ty + i][tx];
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < LOSS_WIN; k++) acc += win.w[k] * col[j + k];
        vs[m][j] = acc;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < RPT; j++) {
    const int r = RPT * ty + j;
    const int gy = y0 + r, gx = x0 + tx;
    if (gy >= H || gx >= W) continue;
    const size_t o = pbase + (size_t)gy * W + gx;
    const float x = img[o], y = gt[o];
    float g = 0.f;
    if (WITH_L1) {
      const float d = x - y;
      g += g_l1 * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    }
    if (WITH_SSIM) {
      const float cm = vs[0][j], c11 = vs[1][j], c12 = vs[2][j];
      g += g_ss * (cm + 2.f * x * c11 + y * c12);
    }
    dimg[o] = g;
  }
}

}  // namespace

LossWS loss_layout(char* base, int planes, int H, int W, unsigned mode) {
  LossWS w;
  const size_t tiles = (size_t)((W + LT - 1) / LT) * ((H + LTY - 1) / LTY);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += (bytes + 255) & ~(size_t)255;
    return p;
  };
  w.tiles = (int)tiles;
  w.partial = (float*)take((size_t)planes * tiles * 2 * sizeof(float));
  w.plane_tmp = (float*)take((size_t)planes * 2 * sizeof(float));
  w.map_stride = (size_t)planes * H * W;
  w.maps = (mode & EOGS_LOSS_SSIM) ? (float*)take(3 * w.map_stride * sizeof(float)) : nullptr;
  w.bytes = off + 256;
  return w;
}

LossWindow loss_window() {
  // gaussian(11, 1.5) (loss_utils.py:26-33): exp in double, stored as fp32, normalised by the fp32 sum
  LossWindow win;
  float g[LOSS_WIN], sum = 0.f;
  for (int i = 0; i < LOSS_WIN; i++) {
    const double d = (double)(i - LOSS_WIN / 2);
    g[i] = (float)exp(-(d * d) / (2.0 * 1.5 * 1.5));
    sum += g[i];
  }
  for (int i = 0; i < LOSS_WIN; i++) win.w[i] = g[i] / sum;
  return win;
}

void launch_loss_fwd(const LossWS& w, int planes, int H, int W, const float* img, const float* gt, unsigned mode,
                     float w_l1, float w_ssim, float bias, float* out, float* plane_sums, hipStream_t s) {
  const dim3 grid((W + LT - 1) / LT, (H + LTY - 1) / LTY, planes);
  const LossWindow win = loss_window();
  const bool l1 = mode & EOGS_LOSS_L1, ss = mode & EOGS_LOSS_SSIM;
  auto* kern = ss ? (l1 ? loss_fwd_kernel<true, true> : loss_fwd_kernel<false, true>) : loss_fwd_kernel<true, false>;
  hipLaunchKernelGGL(kern, grid, dim3(LTHREADS), 0, s, H, W, img, gt, win, w.maps, w.map_stride, w.partial);
  hipLaunchKernelGGL(loss_plane_reduce_kernel, dim3(planes), dim3(LTHREADS), 0, s, w.partial, w.tiles, w.plane_tmp);
  const float inv_n = (float)(1.0 / ((double)planes * H * W));
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, s, w.plane_tmp, planes, inv_n, w_l1, w_ssim, bias, out,
                     plane_sums);
}

void launch_loss_bwd(const LossWS& w, int planes, int H, int W, const float* img, const float* gt, unsigned mode,
                     float w_l1, float w_ssim, const float* upstream, const float* plane_grad, float* dimg,
                     hipStream_t s) {
  const dim3 grid((W + LT - 1) / LT, (H + LTY - 1) / LTY, planes);
  const LossWindow win = loss_window();
  const float inv_n = (float)(1.0 / ((double)planes * H * W));
  const bool l1 = mode & EOGS_LOSS_L1, ss = mode & EOGS_LOSS_SSIM;
  auto* kern = ss ? (l1 ? loss_bwd_kernel<true, true> : loss_bwd_kernel<false, true>) : loss_bwd_kernel<true, false>;
  hipLaunchKernelGGL(kern, grid, dim3(LTHREADS), 0, s, H, W, img, gt, win, w.maps, w.map_stride, w_l1, w_ssim, inv_n,
                     upstream, plane_grad, dimg);
}
