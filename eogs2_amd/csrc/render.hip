// render.hip — per-tile alpha blending, forward and backward.
//
// One 256-thread workgroup (4 wave64) per 16x16 tile; wave w owns the 8x8 pixel quadrant (w&1, w>>1), so a
// wave's 64 lanes are a compact pixel block (tight bound for wave-level skipping). The tile's depth-ordered
// Gaussian list is staged through LDS in batches of 256: {xy, conic+opacity, 5 colours, 1/depth} = 48 B per
// Gaussian, gathered once per tile with one Gaussian per lane; the inner loop then reads LDS at wave-uniform
// addresses (broadcast, no bank conflicts) instead of re-fetching colours from global per contributing pixel
// as the reference does (DGR/cuda_rasterizer/forward.cu:386).
//
// Forward semantics: DGR/cuda_rasterizer/forward.cu:288-411.
// Backward semantics: DGR/cuda_rasterizer/backward.cu:457-643, restructured:
//   * traversal is FRONT-TO-BACK like the forward (no T /= (1-alpha) division chain). With
//       D_final = sum_ch g_ch * out_ch (+ g_inv * out_invdepth)   [contains the T_final * bg.g term]
//       D_j     = sum_{k<=j} (g . c_k) alpha_k T_k
//     the reference's dL/dalpha_j = T_j (g.c_j - g.accum_rec_j) - T_final/(1-alpha_j) bg.g   (:586-620)
//     equals  T_j (g.c_j) - (D_final - D_j) / (1 - alpha_j): one dot product per pair instead of a
//     5-channel recurrence.
//   * no global atomics: the 12 atomicAdd per contributing (pixel,Gaussian) of the reference (:598-640) are
//     replaced by a wave reduction, a per-tile LDS accumulation and ONE 48-byte record per (tile,Gaussian)
//     pair written with plain stores; gaussian_bwd_kernel sums each Gaussian's records in fixed order
//     (bitwise reproducible gradients).
#include "common.h"

namespace {

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    uint32_t n = __shfl_xor(v, o, 64);
    v = n > v ? n : v;
  }
  return v;
}

struct PixelMap {
  int px, py;
  bool inside;
  uint32_t pix_id;
};
__device__ inline PixelMap pixel_of_thread(int tile, int gx, int W, int H) {
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  const int tx = tile % gx, ty = tile / gx;
  PixelMap m;
  m.px = tx * TILE + (w & 1) * 8 + (l & 7);
  m.py = ty * TILE + (w >> 1) * 8 + (l >> 3);
  m.inside = m.px < W && m.py < H;
  m.pix_id = (uint32_t)m.py * (uint32_t)W + (uint32_t)m.px;
  return m;
}

// Gather one Gaussian per lane into the staging arrays.
__device__ inline void stage_gaussian(uint32_t id, int t, const float2* __restrict__ means2D,
                                      const float4* __restrict__ conic_o, const float* __restrict__ depth,
                                      const float* __restrict__ colors, float2* s_xy, float4* s_co, float* s_ft) {
  s_xy[t] = means2D[id];
  s_co[t] = conic_o[id];
  const float* c = colors + (size_t)id * NCH;
#pragma unroll
  for (int ch = 0; ch < NCH; ch++) s_ft[t * NFEAT + ch] = c[ch];
  s_ft[t * NFEAT + NCH] = 1.f / depth[id];
}

}  // namespace

__global__ __launch_bounds__(BLK) void render_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx,
    const float2* __restrict__ means2D, const float4* __restrict__ conic_o, const float* __restrict__ depth,
    const float* __restrict__ colors, const float* __restrict__ bg, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, float* __restrict__ out_color, float* __restrict__ out_invdepth) {
  __shared__ float2 s_xy[BLK];
  __shared__ float4 s_co[BLK];
  __shared__ float s_ft[BLK * NFEAT];
  const int t = threadIdx.x;
  const int tile = blockIdx.x;
  const PixelMap pm = pixel_of_thread(tile, gx, W, H);
  const float pxf = (float)pm.px, pyf = (float)pm.py;
  const uint2 range = ranges[tile];

  float T = 1.0f;
  uint32_t contributor = 0, last_contributor = 0;
  float C[NCH] = {0.f, 0.f, 0.f, 0.f, 0.f};
  float invd = 0.f;
  bool done = !pm.inside;

  for (uint32_t b0 = range.x; b0 < range.y; b0 += BLK) {
    if (__syncthreads_and(done)) break;  // also fences LDS reuse
    const uint32_t k = b0 + t;
    if (k < range.y) stage_gaussian(point_list[k], t, means2D, conic_o, depth, colors, s_xy, s_co, s_ft);
    __syncthreads();
    const int nb = (int)((range.y - b0) < (uint32_t)BLK ? (range.y - b0) : (uint32_t)BLK);
    for (int j = 0; !done && j < nb; j++) {
      contributor++;
      const float2 xy = s_xy[j];
      const float4 co = s_co[j];
      const float dx = xy.x - pxf, dy = xy.y - pyf;
      const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
      if (power > 0.0f) continue;
      const float alpha = fminf(0.99f, co.w * __expf(power));
      if (alpha < 1.0f / 255.0f) continue;
      const float test_T = T * (1 - alpha);
      if (test_T < 0.0001f) {
        done = true;
        continue;
      }
      const float wgt = alpha * T;
#pragma unroll
      for (int ch = 0; ch < NCH; ch++) C[ch] += s_ft[j * NFEAT + ch] * wgt;
      invd += s_ft[j * NFEAT + NCH] * wgt;
      T = test_T;
      last_contributor = contributor;
    }
  }
  if (pm.inside) {
    const size_t HW = (size_t)H * W;
    final_T[pm.pix_id] = T;
    n_contrib[pm.pix_id] = last_contributor;
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) out_color[ch * HW + pm.pix_id] = C[ch] + T * bg[ch];
    if (out_invdepth) out_invdepth[pm.pix_id] = invd;
  }
}

void launch_render_fwd(const GeomWS& g, const BinWS& b, const ImgWS& im, int H, int W, const float* colors,
                       const float* bg, float* out_color, float* out_invdepth, hipStream_t s) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  hipLaunchKernelGGL(render_fwd_kernel, dim3(gx * gy), dim3(BLK), 0, s, im.ranges, b.point_list, W, H, gx, g.means2D,
                     g.conic_o, g.depth, colors, bg, im.final_T, im.n_contrib, out_color, out_invdepth);
}

// ------------------------------------------------------------------------------------------------------
// Backward
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLK) void render_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx,
    const float2* __restrict__ means2D, const float4* __restrict__ conic_o, const float* __restrict__ depth,
    const float* __restrict__ colors, const uint2* __restrict__ rect, const uint32_t* __restrict__ slot_base,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ out_color,
    const float* __restrict__ out_invdepth, const float* __restrict__ dL_dpix, const float* __restrict__ dL_dinv,
    float* __restrict__ records) {
  __shared__ float2 s_xy[BLK];
  __shared__ float4 s_co[BLK];
  __shared__ float s_ft[BLK * NFEAT];
  __shared__ uint32_t s_id[BLK];
  __shared__ float s_acc[BLK / 64][BLK][REC];  // per-wave partial sums of the current batch
  __shared__ uint32_t s_wmax[BLK / 64];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const int tile = blockIdx.x;
  const int tx = tile % gx, ty = tile / gx;
  const PixelMap pm = pixel_of_thread(tile, gx, W, H);
  const float pxf = (float)pm.px, pyf = (float)pm.py;
  const uint2 range = ranges[tile];
  const size_t HW = (size_t)H * W;
  const bool have_inv = dL_dinv != nullptr;

  float g[NCH] = {0.f, 0.f, 0.f, 0.f, 0.f};
  float ginv = 0.f, Dfinal = 0.f;
  uint32_t ncontrib = 0;
  if (pm.inside) {
    ncontrib = n_contrib[pm.pix_id];
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
      g[ch] = dL_dpix[ch * HW + pm.pix_id];
      Dfinal += g[ch] * out_color[ch * HW + pm.pix_id];
    }
    if (have_inv) {
      ginv = dL_dinv[pm.pix_id];
      Dfinal += ginv * out_invdepth[pm.pix_id];
    }
  }
  // Gaussians past the last contributor of every pixel of this wave / tile cannot receive gradient
  const uint32_t wave_last = wave_max_u32(ncontrib);
  if (lane == 0) s_wmax[w] = wave_last;
  __syncthreads();
  const uint32_t m01 = s_wmax[0] > s_wmax[1] ? s_wmax[0] : s_wmax[1];
  const uint32_t m23 = s_wmax[2] > s_wmax[3] ? s_wmax[2] : s_wmax[3];
  const uint32_t tile_last = m01 > m23 ? m01 : m23;
  const uint32_t list_end = range.x + tile_last < range.y ? range.x + tile_last : range.y;

  const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
  float T = 1.0f, Dacc = 0.f;

  for (uint32_t b0 = range.x; b0 < list_end; b0 += BLK) {
    __syncthreads();  // previous batch's flush has read s_id / s_acc
    const uint32_t k = b0 + t;
    const int nb = (int)((list_end - b0) < (uint32_t)BLK ? (list_end - b0) : (uint32_t)BLK);
    if (k < list_end) {
      const uint32_t id = point_list[k];
      s_id[t] = id;
      stage_gaussian(id, t, means2D, conic_o, depth, colors, s_xy, s_co, s_ft);
    }
#pragma unroll
    for (int ww = 0; ww < BLK / 64; ww++)
#pragma unroll
      for (int c = 0; c < REC; c++) s_acc[ww][t][c] = 0.f;
    __syncthreads();

    const uint32_t jbase = b0 - range.x;  // list index of the batch's first entry
    int jend = nb;
    if (wave_last < jbase + (uint32_t)nb) jend = wave_last > jbase ? (int)(wave_last - jbase) : 0;
    for (int j = 0; j < jend; j++) {
      const float2 xy = s_xy[j];
      const float4 co = s_co[j];
      const float dx = xy.x - pxf, dy = xy.y - pyf;
      const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
      const float G = __expf(power);
      const float alpha = fminf(0.99f, co.w * G);
      const bool valid = (jbase + (uint32_t)j < ncontrib) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
      if (__ballot(valid) == 0ull) continue;  // wave-uniform skip

      float c[REC - 1];
#pragma unroll
      for (int q = 0; q < REC - 1; q++) c[q] = 0.f;
      if (valid) {
        const float wgt = alpha * T;
        float gc = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) gc += g[ch] * s_ft[j * NFEAT + ch];
        if (have_inv) gc += ginv * s_ft[j * NFEAT + NCH];
        Dacc += gc * wgt;
        const float one_m = 1.f - alpha;
        const float dL_dalpha = T * gc - (Dfinal - Dacc) / one_m;
        T = T * one_m;
        const float dL_dG = co.w * dL_dalpha;  // no zeroing when alpha was clamped (backward.cu:624)
        const float gdx = G * dx, gdy = G * dy;
        const float dG_ddelx = -gdx * co.x - gdy * co.y;
        const float dG_ddely = -gdy * co.z - gdx * co.y;
        c[0] = dL_dG * dG_ddelx * ddelx_dx;
        c[1] = dL_dG * dG_ddely * ddely_dy;
        c[2] = -0.5f * gdx * dx * dL_dG;
        c[3] = -0.5f * gdx * dy * dL_dG;
        c[4] = -0.5f * gdy * dy * dL_dG;
        c[5] = G * dL_dalpha;
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) c[6 + ch] = wgt * g[ch];
      }
#pragma unroll
      for (int q = 0; q < REC - 1; q++) {
        const float v = wave_sum(c[q]);
        if (lane == 0) s_acc[w][j][q] = v;
      }
    }
    __syncthreads();
    // flush: one 48-byte record per (tile, Gaussian) pair, fixed wave order
    if (t < nb) {
      float r[REC];
#pragma unroll
      for (int q = 0; q < REC; q++) r[q] = ((s_acc[0][t][q] + s_acc[1][t][q]) + s_acc[2][t][q]) + s_acc[3][t][q];
      const uint32_t id = s_id[t];
      const uint2 rc = rect[id];
      const uint32_t x0 = rc.x & 0xFFFFu, x1 = rc.x >> 16, y0 = rc.y & 0xFFFFu;
      const uint32_t slot = slot_base[id] + ((uint32_t)ty - y0) * (x1 - x0) + ((uint32_t)tx - x0);
      float4* dst = reinterpret_cast<float4*>(records + (size_t)slot * REC);
      dst[0] = make_float4(r[0], r[1], r[2], r[3]);
      dst[1] = make_float4(r[4], r[5], r[6], r[7]);
      dst[2] = make_float4(r[8], r[9], r[10], 0.f);
    }
  }
  // list entries beyond the last contributor of the whole tile still own a record: zero it
  for (uint32_t k = list_end + t; k < range.y; k += BLK) {
    const uint32_t id = point_list[k];
    const uint2 rc = rect[id];
    const uint32_t x0 = rc.x & 0xFFFFu, x1 = rc.x >> 16, y0 = rc.y & 0xFFFFu;
    const uint32_t slot = slot_base[id] + ((uint32_t)ty - y0) * (x1 - x0) + ((uint32_t)tx - x0);
    float4* dst = reinterpret_cast<float4*>(records + (size_t)slot * REC);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    dst[0] = z; dst[1] = z; dst[2] = z;
  }
}

void launch_render_bwd(const GeomWS& g, const BinWS& b, const ImgWS& im, int H, int W, const float* colors,
                       const float* out_color, const float* out_invdepth, const float* dL_dcolor,
                       const float* dL_dinvdepth, hipStream_t s) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  hipLaunchKernelGGL(render_bwd_kernel, dim3(gx * gy), dim3(BLK), 0, s, im.ranges, b.point_list, W, H, gx, g.means2D,
                     g.conic_o, g.depth, colors, g.rect, g.slot_base, im.n_contrib, out_color, out_invdepth, dL_dcolor,
                     dL_dinvdepth, b.records);
}
