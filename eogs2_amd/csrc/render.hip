// render.hip — alpha blending per internal 8x8 tile, forward and backward.
//
// Work decomposition (both directions)
//   * ONE wave64 per internal 8x8 tile, lane = pixel. Waves are fully independent: no LDS, no barriers, no
//     cross-wave merge. A 256-thread workgroup is just four tiles; workgroups are dealt to the 8 XCDs so that
//     each XCD (own L2) walks one contiguous band of tiles and neighbouring tiles share their Gaussians in L2.
//   * the tile's list (built by binning.hip) holds exactly the Gaussians that can reach alpha >= 1/255 inside
//     the tile, in (depth, index) order. The wave takes it 64 entries at a time: lane i gathers entry i
//     ({xy, conic+opacity, 5 colours, 1/depth} = 48 B) into registers, the NEXT chunk's gather is issued before
//     the current chunk is consumed (software pipeline), and each entry's parameters are then broadcast from
//     the owning lane with v_readlane into SGPRs — the hot loop touches no memory (the reference re-fetches
//     colours from global per contributing pixel, DGR/cuda_rasterizer/forward.cu:386).
//
// Forward semantics: DGR/cuda_rasterizer/forward.cu:288-411.
// Backward semantics: DGR/cuda_rasterizer/backward.cu:457-643, restructured:
//   * traversal is FRONT-TO-BACK like the forward (no T /= (1-alpha) division chain). With
//       D_final = sum_ch g_ch * out_ch (+ g_inv * out_invdepth)   [contains the T_final * bg.g term]
//       D_j     = sum_{k<=j} (g . c_k) alpha_k T_k
//     the reference's dL/dalpha_j = T_j (g.c_j - g.accum_rec_j) - T_final/(1-alpha_j) bg.g   (:586-620)
//     equals  T_j (g.c_j) - (D_final - D_j) / (1 - alpha_j): one dot product per pair instead of a
//     5-channel recurrence.
//   * no atomics: the 12 atomicAdd per contributing (pixel,Gaussian) of the reference (:598-640) become a DPP
//     wave reduction and ONE 48-byte record per (tile,Gaussian) pair, written with plain stores by the only
//     wave that owns the pair; gaussian_bwd_kernel sums each Gaussian's records in fixed order (bitwise
//     reproducible gradients).
#include "common.h"

#pragma clang fp contract(fast)

namespace {

// ---- wave64 helpers ----
__device__ inline float rl(float v, int lane) {  // broadcast lane `lane` (wave-uniform) to an SGPR
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ inline uint32_t rlu(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }

// ---- wave64 sum of 11 values at once: 6 DPP steps x 11 registers = 66 v_add_f32_dpp ----
// GFX9 DPP reduction: row_shr 1/2/4/8 (bound_ctrl: out-of-row sources read 0) build each 16-lane row's inclusive
// scan (lane 15 of a row = row sum); row_bcast:15 (rows 1,3) adds the previous row's sum; row_bcast:31 (rows 2,3)
// adds lane 31's. The totals are valid in LANE 63 ONLY. Written as inline asm because hipcc materialises
// "old = 0" moves around the masked steps (3 instructions per step). An asm statement is opaque to the hazard
// recogniser: a DPP read of a VGPR written by the previous VALU instruction needs 2 wait states, so each block
// starts with s_nop 1; inside a block the 11 chains are independent and every register is re-read 11
// instructions after it was written.
#define DPP_STEP11(CTRL)                                                                                         \
  asm volatile("s_nop 1\n\t"                                                                                     \
               "v_add_f32_dpp %0, %0, %0 " CTRL "\n\t"                                                            \
               "v_add_f32_dpp %1, %1, %1 " CTRL "\n\t"                                                            \
               "v_add_f32_dpp %2, %2, %2 " CTRL "\n\t"                                                            \
               "v_add_f32_dpp %3, %3, %3 " CTRL "\n\t"                                                            \
               "v_add_f32_dpp %4, %4, %4 " CTRL "\n\t"                                                            \
               "v_add_f32_dpp %5, %5, %5 " CTRL "\n\t"                                                            \
               "v_add_f32_dpp %6, %6, %6 " CTRL "\n\t"                                                            \
               "v_add_f32_dpp %7, %7, %7 " CTRL "\n\t"                                                            \
               "v_add_f32_dpp %8, %8, %8 " CTRL "\n\t"                                                            \
               "v_add_f32_dpp %9, %9, %9 " CTRL "\n\t"                                                            \
               "v_add_f32_dpp %10, %10, %10 " CTRL                                                                \
               : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), \
                 "+v"(c[8]), "+v"(c[9]), "+v"(c[10]))
__device__ inline void wave_sum11_lane63(float (&c)[REC]) {
  DPP_STEP11("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
  DPP_STEP11("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1");
  DPP_STEP11("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1");
  DPP_STEP11("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1");
  DPP_STEP11("row_bcast:15 row_mask:0xa bank_mask:0xf");
  DPP_STEP11("row_bcast:31 row_mask:0xc bank_mask:0xf");
}
// compiler-scheduled single-value form (self test reference)
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
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

// XCD-aware tile of this wave: workgroup b runs on XCD b % 8 (round-robin dispatch; speed only, never
// correctness), so XCD x gets the contiguous run of tile groups [x*per, (x+1)*per). gridDim.x is a multiple of 8.
__device__ inline int tile_of_wave() {
  const int per = gridDim.x >> 3;
  const int grp = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  // the wave index is uniform across the wave: tell the compiler, so tile, list range and loop control live in SGPRs
  return grp * (BLK / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

// One list entry held by a lane.
struct Cand {
  float2 xy;
  float4 co;
  float ft[NFEAT];
  uint32_t slot;
};

__device__ inline Cand load_cand(uint32_t k, uint32_t end, const uint32_t* __restrict__ point_list,
                                 const uint32_t* __restrict__ gid, const float2* __restrict__ means2D,
                                 const float4* __restrict__ conic_o, const float* __restrict__ depth,
                                 const float* __restrict__ colors) {
  Cand c;
  c.xy = make_float2(0.f, 0.f);
  c.co = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int q = 0; q < NFEAT; q++) c.ft[q] = 0.f;
  c.slot = 0;
  if (k < end) {
    c.slot = point_list[k];
    const uint32_t id = gid[c.slot];
    c.xy = means2D[id];
    c.co = conic_o[id];
    const float* f = colors + (size_t)id * NCH;
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) c.ft[ch] = f[ch];
    c.ft[NCH] = 1.f / depth[id];
  }
  return c;
}

}  // namespace

__global__ __launch_bounds__(BLK) void render_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const uint32_t* __restrict__ gid,
    int W, int H, int gsx, int ntiles, const float2* __restrict__ means2D, const float4* __restrict__ conic_o,
    const float* __restrict__ depth, const float* __restrict__ colors, const float* __restrict__ bg,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
    float* __restrict__ out_invdepth) {
  const int lane = threadIdx.x & 63;
  const int tile = tile_of_wave();
  if (tile >= ntiles) return;  // wave-uniform; waves never synchronise with each other
  const int px = (tile % gsx) * SUB + (lane & 7), py = (tile / gsx) * SUB + (lane >> 3);
  const bool inside = px < W && py < H;
  const uint32_t pix_id = (uint32_t)py * (uint32_t)W + (uint32_t)px;
  const float pxf = (float)px, pyf = (float)py;
  const uint2 range = ranges[tile];

  float T = 1.0f;
  uint32_t last_contributor = 0;
  float C[NCH] = {0.f, 0.f, 0.f, 0.f, 0.f};
  float invd = 0.f;
  bool done = !inside;

  Cand cur = load_cand(range.x + lane, range.y, point_list, gid, means2D, conic_o, depth, colors);
  for (uint32_t c0 = range.x; c0 < range.y; c0 += 64) {
    const Cand nxt = load_cand(c0 + 64 + lane, range.y, point_list, gid, means2D, conic_o, depth, colors);
    if (__ballot(!done) == 0ull) break;  // every pixel of the tile has terminated
    const int n = (int)((range.y - c0) < 64u ? (range.y - c0) : 64u);
    const uint32_t jbase = c0 - range.x;
    for (int j = 0; j < n; j++) {
      const float gxs = rl(cur.xy.x, j), gys = rl(cur.xy.y, j);
      const float ca = rl(cur.co.x, j), cb = rl(cur.co.y, j), cc = rl(cur.co.z, j), op = rl(cur.co.w, j);
      const float dx = gxs - pxf, dy = gys - pyf;
      const float power = -0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
      const float alpha = fminf(0.99f, op * __expf(power));
      bool valid = !done && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
      const float test_T = T * (1.f - alpha);
      const bool term = valid && test_T < 0.0001f;  // this Gaussian is NOT blended; the pixel is finished
      done = done || term;
      valid = valid && !term;
      if (__ballot(valid) == 0ull) continue;
      const float wgt = valid ? alpha * T : 0.f;
#pragma unroll
      for (int ch = 0; ch < NCH; ch++) C[ch] += rl(cur.ft[ch], j) * wgt;
      invd += rl(cur.ft[NCH], j) * wgt;
      T = valid ? test_T : T;
      last_contributor = valid ? jbase + (uint32_t)j + 1u : last_contributor;
    }
    cur = nxt;
  }
  if (inside) {
    const size_t HW = (size_t)H * W;
    final_T[pix_id] = T;
    n_contrib[pix_id] = last_contributor;
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) out_color[ch * HW + pix_id] = C[ch] + T * bg[ch];
    if (out_invdepth) out_invdepth[pix_id] = invd;
  }
}

static inline uint32_t render_grid(int ntiles) {
  const uint32_t groups = ceil_div_u32((uint64_t)ntiles, BLK / 64);
  return ((groups + 7u) / 8u) * 8u;  // multiple of 8 for the XCD band mapping
}

void launch_render_fwd(const GeomWS& g, const BinWS& b, const ImgWS& im, int H, int W, const float* colors,
                       const float* bg, float* out_color, float* out_invdepth, hipStream_t s) {
  const int gsx = (W + SUB - 1) / SUB, gsy = (H + SUB - 1) / SUB, ntiles = gsx * gsy;
  hipLaunchKernelGGL(render_fwd_kernel, dim3(render_grid(ntiles)), dim3(BLK), 0, s, im.ranges, b.point_list, b.gid, W, H,
                     gsx, ntiles, g.means2D, g.conic_o, g.depth, colors, bg, im.final_T, im.n_contrib, out_color,
                     out_invdepth);
}

// ------------------------------------------------------------------------------------------------------
// Backward
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLK) void render_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const uint32_t* __restrict__ gid,
    int W, int H, int gsx, int ntiles, const float2* __restrict__ means2D, const float4* __restrict__ conic_o,
    const float* __restrict__ depth, const float* __restrict__ colors, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ out_color, const float* __restrict__ out_invdepth, const float* __restrict__ dL_dpix,
    const float* __restrict__ dL_dinv, float* __restrict__ records) {
  const int lane = threadIdx.x & 63;
  const int tile = tile_of_wave();
  if (tile >= ntiles) return;
  const int px = (tile % gsx) * SUB + (lane & 7), py = (tile / gsx) * SUB + (lane >> 3);
  const bool inside = px < W && py < H;
  const uint32_t pix_id = (uint32_t)py * (uint32_t)W + (uint32_t)px;
  const float pxf = (float)px, pyf = (float)py;
  const uint2 range = ranges[tile];
  const size_t HW = (size_t)H * W;
  const bool have_inv = dL_dinv != nullptr;

  float g[NCH] = {0.f, 0.f, 0.f, 0.f, 0.f};
  float ginv = 0.f, Dfinal = 0.f;
  uint32_t ncontrib = 0;
  if (inside) {
    ncontrib = n_contrib[pix_id];
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
      g[ch] = dL_dpix[ch * HW + pix_id];
      Dfinal += g[ch] * out_color[ch * HW + pix_id];
    }
    if (have_inv) {
      ginv = dL_dinv[pix_id];
      Dfinal += ginv * out_invdepth[pix_id];
    }
  }
  // list entries past the last contributor of every pixel of the tile receive no gradient
  const uint32_t tile_last = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(ncontrib));

  const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
  float T = 1.0f, Dacc = 0.f;

  Cand cur = load_cand(range.x + lane, range.y, point_list, gid, means2D, conic_o, depth, colors);
  for (uint32_t c0 = range.x; c0 < range.y; c0 += 64) {
    const Cand nxt = load_cand(c0 + 64 + lane, range.y, point_list, gid, means2D, conic_o, depth, colors);
    const int n = (int)((range.y - c0) < 64u ? (range.y - c0) : 64u);
    const uint32_t jbase = c0 - range.x;
    int jn = n;  // entries of this chunk that can still matter
    if (tile_last < jbase + (uint32_t)n) jn = tile_last > jbase ? (int)(tile_last - jbase) : 0;
    unsigned long long written = 0ull;
    for (int j = 0; j < jn; j++) {
      const float gxs = rl(cur.xy.x, j), gys = rl(cur.xy.y, j);
      const float ca = rl(cur.co.x, j), cb = rl(cur.co.y, j), cc = rl(cur.co.z, j), op = rl(cur.co.w, j);
      const float dx = gxs - pxf, dy = gys - pyf;
      const float power = -0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
      const float G = __expf(power);
      const float alpha = fminf(0.99f, op * G);
      const bool valid = (jbase + (uint32_t)j < ncontrib) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
      if (__ballot(valid) == 0ull) continue;  // wave-uniform skip

      float gc = 0.f;
#pragma unroll
      for (int ch = 0; ch < NCH; ch++) gc += g[ch] * rl(cur.ft[ch], j);
      if (have_inv) gc += ginv * rl(cur.ft[NCH], j);
      const float wgt = valid ? alpha * T : 0.f;
      Dacc += gc * wgt;
      const float one_m = 1.f - alpha;
      const float dL_dalpha = valid ? T * gc - (Dfinal - Dacc) * __builtin_amdgcn_rcpf(one_m) : 0.f;
      T = valid ? T * one_m : T;
      const float dL_dG = op * dL_dalpha;  // no zeroing when alpha was clamped (backward.cu:624)
      const float Gs = valid ? G : 0.f;    // exp() may overflow on lanes that skip this Gaussian
      const float gdx = Gs * dx, gdy = Gs * dy;
      const float dG_ddelx = -gdx * ca - gdy * cb;
      const float dG_ddely = -gdy * cc - gdx * cb;
      float c[REC];
      c[0] = dL_dG * dG_ddelx * ddelx_dx;
      c[1] = dL_dG * dG_ddely * ddely_dy;
      c[2] = -0.5f * gdx * dx * dL_dG;
      c[3] = -0.5f * gdx * dy * dL_dG;
      c[4] = -0.5f * gdy * dy * dL_dG;
      c[5] = Gs * dL_dalpha;
#pragma unroll
      for (int ch = 0; ch < NCH; ch++) c[6 + ch] = wgt * g[ch];
      c[REC - 1] = 0.f;
      wave_sum11_lane63(c);
      const uint32_t slot = rlu(cur.slot, j);
      if (lane == 63) {
        float4* dst = reinterpret_cast<float4*>(records + (size_t)slot * REC);
        dst[0] = make_float4(c[0], c[1], c[2], c[3]);
        dst[1] = make_float4(c[4], c[5], c[6], c[7]);
        dst[2] = make_float4(c[8], c[9], c[10], 0.f);
      }
      written |= 1ull << j;
    }
    // every pair owns a record: entries that reached no pixel get zeros
    if (lane < n && !((written >> lane) & 1ull)) {
      float4* dst = reinterpret_cast<float4*>(records + (size_t)cur.slot * REC);
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      dst[0] = z; dst[1] = z; dst[2] = z;
    }
    cur = nxt;
  }
}

void launch_render_bwd(const GeomWS& g, const BinWS& b, const ImgWS& im, int H, int W, const float* colors,
                       const float* out_color, const float* out_invdepth, const float* dL_dcolor,
                       const float* dL_dinvdepth, hipStream_t s) {
  const int gsx = (W + SUB - 1) / SUB, gsy = (H + SUB - 1) / SUB, ntiles = gsx * gsy;
  hipLaunchKernelGGL(render_bwd_kernel, dim3(render_grid(ntiles)), dim3(BLK), 0, s, im.ranges, b.point_list, b.gid, W, H,
                     gsx, ntiles, g.means2D, g.conic_o, g.depth, colors, im.n_contrib, out_color, out_invdepth,
                     dL_dcolor, dL_dinvdepth, b.records);
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
  float c[REC];
#pragma unroll
  for (int q = 0; q < REC; q++) c[q] = v * (float)(q + 1) + (float)q * 0.25f;  // freshly written VGPRs (hazard case)
  wave_sum11_lane63(c);
#pragma unroll
  for (int q = 0; q < REC - 1; q++) {
    float r = v * (float)(q + 1) + (float)q * 0.25f;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) r += __shfl_xor(r, o, 64);
    if (lane == 63 && c[q] != r) bad |= 4u;
  }
  if (b != (float)((17 * 37 + 11) % 101) - 50.f) bad |= 2u;
  if (bad) atomicOr(out, bad);
}
void launch_selftest(uint32_t* out, hipStream_t s) { hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(256), 0, s, out); }
