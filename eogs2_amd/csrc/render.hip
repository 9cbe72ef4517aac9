// render.hip — per-tile alpha blending, forward and backward.
//
// Work decomposition (both directions)
//   * one 256-thread workgroup (4 wave64) per 16x16 tile; wave w owns the 8x8 pixel block (w&1, w>>1);
//   * the tile's depth-ordered Gaussian list is staged through LDS in batches of 256 candidates, one
//     candidate per lane: {xy, conic+opacity, 5 colours, 1/depth} = 48 B, gathered ONCE per tile (the
//     reference re-fetches colours from global per contributing pixel, DGR/cuda_rasterizer/forward.cu:386);
//   * wave64 ballot compaction: every wave takes the batch 64 candidates at a time (one per lane, read from
//     LDS conflict-free), tests the candidate's alpha >= 1/255 ellipse against ITS 8x8 block with an exact
//     continuous minimisation of the conic over the block (conservative margin), and ballots. Only the
//     survivors are visited, in list order, by a scalar bit loop; a survivor's parameters are broadcast from
//     the owning lane with v_readlane into SGPRs, so the hot loop has no memory latency at all.
//     A culled candidate would have been skipped by every pixel of the block (alpha < 1/255, forward.cu:375),
//     so results are identical to visiting every list entry; with the reference's init opacity 0.01 the
//     alpha >= 1/255 footprint is 1.37 sigma against the 3 sigma tile rect, i.e. ~5x fewer visits.
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
//     replaced by a DPP wave reduction, a per-tile LDS accumulation and ONE 48-byte record per (tile,Gaussian)
//     pair written with plain stores; gaussian_bwd_kernel sums each Gaussian's records in fixed order
//     (bitwise reproducible gradients).
#include "common.h"

#pragma clang fp contract(fast)

namespace {

// ---- wave64 helpers ----
__device__ inline float rl(float v, int lane) {  // broadcast lane `lane` (wave-uniform) to an SGPR
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <int CTRL, int ROW_MASK>
__device__ inline float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// Sum over the 64 lanes; the total is valid in lane 63 only. GFX9 DPP: row_shr 1/2/4/8 build each row's
// inclusive scan (lane 15 of a row = row sum), row_bcast15 adds it into the next row, row_bcast31 into rows 2-3.
__device__ inline float wave_sum_lane63(float v) {
  v = dpp_add<0x111, 0xf>(v);
  v = dpp_add<0x112, 0xf>(v);
  v = dpp_add<0x114, 0xf>(v);
  v = dpp_add<0x118, 0xf>(v);
  v = dpp_add<0x142, 0xa>(v);
  v = dpp_add<0x143, 0xc>(v);
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

// Can this Gaussian reach alpha >= 1/255 anywhere in the pixel block [x0,x1] x [y0,y1]?
// alpha = o exp(-q/2) with q(d) = a dx^2 + 2 b dx dy + c dy^2 (d = centre - pixel), so alpha >= 1/255 <=> q <= tau,
// tau = 2 ln(255 o). The minimum of the convex q over the block is 0 if the centre is inside, otherwise it lies on
// the edges that face the centre: minimise q along x = clamp(gx) and along y = clamp(gy) with the free coordinate
// clamped to the block. The continuous block contains the pixel centres, so q_min(block) <= q(pixel): culling when
// q_min > tau (plus a margin far above fp32 rounding of `power`) never drops a contributing Gaussian.
// NaNs (degenerate conics) fail the comparison and are kept.
__device__ inline bool block_hit(float gx, float gy, float a, float b, float c, float o, float x0, float y0, float x1,
                                 float y1) {
  const float cx = fminf(fmaxf(gx, x0), x1), cy = fminf(fmaxf(gy, y0), y1);
  const float dxe = gx - cx, dye = gy - cy;
  // edge x = cx: free y
  const float py = fminf(fmaxf(gy + b * dxe / c, y0), y1);
  const float dy1 = gy - py;
  const float q1 = a * dxe * dxe + 2.f * b * dxe * dy1 + c * dy1 * dy1;
  // edge y = cy: free x
  const float pxs = fminf(fmaxf(gx + b * dye / a, x0), x1);
  const float dx2 = gx - pxs;
  const float q2 = a * dx2 * dx2 + 2.f * b * dx2 * dye + c * dye * dye;
  const float qmin = fminf(q1, q2);
  const float tau = 2.f * __logf(255.f * o);
  return !(qmin > tau + 1e-3f * (1.f + fabsf(tau)));
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
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const int tile = blockIdx.x;
  const PixelMap pm = pixel_of_thread(tile, gx, W, H);
  const float pxf = (float)pm.px, pyf = (float)pm.py;
  const uint2 range = ranges[tile];
  // this wave's 8x8 block in pixel coordinates (wave-uniform)
  const float bx0 = (float)((tile % gx) * TILE + (w & 1) * 8), by0 = (float)((tile / gx) * TILE + (w >> 1) * 8);
  const float bx1 = bx0 + 7.f, by1 = by0 + 7.f;

  float T = 1.0f;
  uint32_t last_contributor = 0;
  float C[NCH] = {0.f, 0.f, 0.f, 0.f, 0.f};
  float invd = 0.f;
  bool done = !pm.inside;

  for (uint32_t b0 = range.x; b0 < range.y; b0 += BLK) {
    if (__syncthreads_and(done)) break;  // also fences LDS reuse
    const uint32_t k = b0 + t;
    if (k < range.y) stage_gaussian(point_list[k], t, means2D, conic_o, depth, colors, s_xy, s_co, s_ft);
    __syncthreads();
    const int nb = (int)((range.y - b0) < (uint32_t)BLK ? (range.y - b0) : (uint32_t)BLK);
    const uint32_t jbase = b0 - range.x;
    for (int cb = 0; cb < nb; cb += 64) {
      if (__ballot(!done) == 0ull) break;  // every pixel of this wave has terminated
      const int ci = cb + lane;
      const bool cand = ci < nb;
      const float2 cxy = s_xy[cand ? ci : 0];
      const float4 cco = s_co[cand ? ci : 0];
      unsigned long long mask = __ballot(cand && block_hit(cxy.x, cxy.y, cco.x, cco.y, cco.z, cco.w, bx0, by0, bx1, by1));
      if (mask == 0ull) continue;
      float cf[NFEAT];
#pragma unroll
      for (int q = 0; q < NFEAT; q++) cf[q] = s_ft[(cand ? ci : 0) * NFEAT + q];
      while (mask) {
        const int j = __builtin_ctzll(mask);
        mask &= mask - 1ull;
        const float gxs = rl(cxy.x, j), gys = rl(cxy.y, j);
        const float ca = rl(cco.x, j), cbb = rl(cco.y, j), cc = rl(cco.z, j), op = rl(cco.w, j);
        const float dx = gxs - pxf, dy = gys - pyf;
        const float power = -0.5f * (ca * dx * dx + cc * dy * dy) - cbb * dx * dy;
        const float alpha = fminf(0.99f, op * __expf(power));
        bool valid = !done && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
        const float test_T = T * (1.f - alpha);
        const bool term = valid && test_T < 0.0001f;  // this Gaussian is NOT blended; the pixel is finished
        done = done || term;
        valid = valid && !term;
        if (__ballot(valid) == 0ull) continue;
        const float wgt = valid ? alpha * T : 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) C[ch] += rl(cf[ch], j) * wgt;
        invd += rl(cf[NCH], j) * wgt;
        T = valid ? test_T : T;
        last_contributor = valid ? jbase + (uint32_t)(cb + j) + 1u : last_contributor;
      }
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
  const float bx0 = (float)(tx * TILE + (w & 1) * 8), by0 = (float)(ty * TILE + (w >> 1) * 8);
  const float bx1 = bx0 + 7.f, by1 = by0 + 7.f;

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
    int jend = nb;                        // candidates at list index >= wave_last cannot matter to this wave
    if (wave_last < jbase + (uint32_t)nb) jend = wave_last > jbase ? (int)(wave_last - jbase) : 0;
    for (int cb = 0; cb < jend; cb += 64) {
      const int ci = cb + lane;
      const bool cand = ci < jend;
      const float2 cxy = s_xy[cand ? ci : 0];
      const float4 cco = s_co[cand ? ci : 0];
      unsigned long long mask = __ballot(cand && block_hit(cxy.x, cxy.y, cco.x, cco.y, cco.z, cco.w, bx0, by0, bx1, by1));
      if (mask == 0ull) continue;
      float cf[NFEAT];
#pragma unroll
      for (int q = 0; q < NFEAT; q++) cf[q] = s_ft[(cand ? ci : 0) * NFEAT + q];
      while (mask) {
        const int j = __builtin_ctzll(mask);
        mask &= mask - 1ull;
        const float gxs = rl(cxy.x, j), gys = rl(cxy.y, j);
        const float ca = rl(cco.x, j), cbb = rl(cco.y, j), cc = rl(cco.z, j), op = rl(cco.w, j);
        const float dx = gxs - pxf, dy = gys - pyf;
        const float power = -0.5f * (ca * dx * dx + cc * dy * dy) - cbb * dx * dy;
        const float G = __expf(power);
        const float alpha = fminf(0.99f, op * G);
        const bool valid = (jbase + (uint32_t)(cb + j) < ncontrib) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
        if (__ballot(valid) == 0ull) continue;  // wave-uniform skip

        float gc = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) gc += g[ch] * rl(cf[ch], j);
        if (have_inv) gc += ginv * rl(cf[NCH], j);
        const float wgt = valid ? alpha * T : 0.f;
        Dacc += gc * wgt;
        const float one_m = 1.f - alpha;
        const float dL_dalpha = valid ? T * gc - (Dfinal - Dacc) * __frcp_rn(one_m) : 0.f;
        T = valid ? T * one_m : T;
        const float dL_dG = op * dL_dalpha;  // no zeroing when alpha was clamped (backward.cu:624)
        const float Gs = valid ? G : 0.f;    // exp() may overflow on lanes that skip this Gaussian
        const float gdx = Gs * dx, gdy = Gs * dy;
        const float dG_ddelx = -gdx * ca - gdy * cbb;
        const float dG_ddely = -gdy * cc - gdx * cbb;
        float c[REC - 1];
        c[0] = dL_dG * dG_ddelx * ddelx_dx;
        c[1] = dL_dG * dG_ddely * ddely_dy;
        c[2] = -0.5f * gdx * dx * dL_dG;
        c[3] = -0.5f * gdx * dy * dL_dG;
        c[4] = -0.5f * gdy * dy * dL_dG;
        c[5] = Gs * dL_dalpha;
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) c[6 + ch] = wgt * g[ch];
        float* acc = &s_acc[w][cb + j][0];
#pragma unroll
        for (int q = 0; q < REC - 1; q++) {
          const float v = wave_sum_lane63(c[q]);
          if (lane == 63) acc[q] = v;
        }
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

// ---- self test of the wave64 primitives (diagnostics; returns mismatching lanes in out[0]) ----
__global__ void selftest_kernel(uint32_t* out) {
  const int lane = threadIdx.x & 63;
  const float v = (float)((lane * 37 + 11) % 101) - 50.f;
  float ref = v;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) ref += __shfl_xor(ref, o, 64);
  const float got = wave_sum_lane63(v);
  const float b = rl(v, 17);
  uint32_t bad = 0;
  if (lane == 63 && got != ref) bad |= 1u;
  if (b != (float)((17 * 37 + 11) % 101) - 50.f) bad |= 2u;
  if (bad) atomicOr(out, bad);
}
void launch_selftest(uint32_t* out, hipStream_t s) { hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(256), 0, s, out); }
