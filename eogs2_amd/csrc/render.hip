// render.hip — alpha blending per internal 16x8 tile, forward and backward.
//
// Work decomposition (both directions)
//   * ONE wave64 per internal 16x8 tile; lane l owns the two horizontally adjacent pixels
//     (2*(l&7), 2*(l&7)+1) of tile row l>>3 and evaluates them with PACKED fp32 (v_pk_fma/mul/add_f32): plain
//     fp32 VALU issues a wave64 instruction in 4 cycles on gfx950 (measured: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU),
//     packed fp32 does two pixels in the same 4 cycles, and every per-Gaussian cost (parameter broadcast, wave
//     reduction) is paid once per 128 pixels. Waves are fully independent: no LDS, no barriers, no cross-wave
//     merge. A 256-thread workgroup is just four tiles; workgroups are dealt to the 8 XCDs so that each XCD (own
//     L2) walks one contiguous band of tiles and neighbouring tiles share their Gaussians in L2.
//   * the tile's list (built by binning.hip) holds exactly the Gaussians that can reach alpha >= 1/255 inside
//     the tile, in (depth, index) order. The wave takes it 64 entries at a time: lane i gathers entry i
//     ({xy, conic+opacity, 5 colours, 1/depth} = 48 B) into registers, the NEXT chunk's gather is issued before
//     the current chunk is consumed (software pipeline), and each entry's parameters are then broadcast from
//     the owning lane with v_readlane into SGPRs — the hot loop touches no memory (the reference re-fetches
//     colours from global per contributing pixel, DGR/cuda_rasterizer/forward.cu:386).
//   * the conic is pre-scaled by log2(e) at gather time, so alpha = o * 2^p with p = (A dx - B dy) dx + C dy^2,
//     A = -a log2e / 2, B = b log2e, C = -c log2e / 2: two packed FMAs and one v_exp_f32 per pixel.
//
// Forward semantics: DGR/cuda_rasterizer/forward.cu:288-411.
// Backward semantics: DGR/cuda_rasterizer/backward.cu:457-643, restructured:
//   * traversal is FRONT-TO-BACK like the forward (no T /= (1-alpha) division chain). With
//       D_final = sum_ch g_ch * out_ch (+ g_inv * out_invdepth)   [contains the T_final * bg.g term]
//       D_j     = sum_{k<=j} (g . c_k) alpha_k T_k
//     the reference's dL/dalpha_j = T_j (g.c_j - g.accum_rec_j) - T_final/(1-alpha_j) bg.g   (:586-620)
//     equals  T_j (g.c_j) - (D_final - D_j) / (1 - alpha_j): one dot product per pair instead of a
//     5-channel recurrence.
//   * per (tile, Gaussian) the position/conic/opacity gradients are linear in six moments of v = G dL/dalpha:
//       M = sum_pixels v * {1, dx, dy, dx^2, dx dy, dy^2}
//     (:624-640: dL/dmean2D = o (W/2, H/2) * (-(a M_dx + b M_dy), -(c M_dy + b M_dx)), dL/dconic = -o/2 (M_dxdx,
//     M_dxdy, M_dydy), dL/dopacity = M_1), so a lane adds its two pixels, the wave reduces 6 moments + 5 colour
//     sums with DPP, and lane 63 applies the per-Gaussian factors once.
//   * no atomics: the 12 atomicAdd per contributing (pixel,Gaussian) of the reference (:598-640) become that
//     DPP reduction and ONE 48-byte record per (tile,Gaussian) pair, written with plain stores by the only wave
//     that owns the pair; gaussian_bwd_kernel sums each Gaussian's records in fixed order (bitwise
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

typedef float f2 __attribute__((ext_vector_type(2)));
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

__device__ inline f2 splat(float v) { return f2{v, v}; }
__device__ inline f2 exp2_2(f2 p) { return f2{__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)}; }
__device__ inline f2 sel(bool c0, bool c1, f2 a, f2 b) { return f2{c0 ? a.x : b.x, c1 ? a.y : b.y}; }

// One list entry held by a lane (conic pre-scaled by log2 e, see header).
struct Cand {
  float gx, gy, A, B, C, op;
  float ft[NFEAT];
  uint32_t slot;
};

__device__ inline Cand load_cand(uint32_t k, uint32_t end, const uint32_t* __restrict__ point_list,
                                 const uint32_t* __restrict__ gid, const float2* __restrict__ means2D,
                                 const float4* __restrict__ conic_o, const float* __restrict__ depth,
                                 const float* __restrict__ colors) {
  Cand c;
  c.gx = c.gy = c.A = c.B = c.C = c.op = 0.f;
#pragma unroll
  for (int q = 0; q < NFEAT; q++) c.ft[q] = 0.f;
  c.slot = 0;
  if (k < end) {
    c.slot = point_list[k];
    const uint32_t id = gid[c.slot];
    const float2 xy = means2D[id];
    const float4 co = conic_o[id];
    c.gx = xy.x; c.gy = xy.y;
    c.A = co.x * (-0.5f * LOG2E); c.B = co.y * LOG2E; c.C = co.z * (-0.5f * LOG2E); c.op = co.w;
    const float* f = colors + (size_t)id * NCH;
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) c.ft[ch] = f[ch];
    c.ft[NCH] = 1.f / depth[id];
  }
  return c;
}

// the two pixels of a lane
struct Pix {
  int px0, py;
  bool in0, in1;
  uint32_t id0;
};
__device__ inline Pix pixels_of_lane(int tile, int gsx, int W, int H) {
  const int lane = threadIdx.x & 63;
  Pix p;
  p.px0 = (tile % gsx) * SUBX + 2 * (lane & 7);
  p.py = (tile / gsx) * SUBY + (lane >> 3);
  p.in0 = p.px0 < W && p.py < H;
  p.in1 = p.px0 + 1 < W && p.py < H;
  p.id0 = (uint32_t)p.py * (uint32_t)W + (uint32_t)p.px0;
  return p;
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
  const Pix pm = pixels_of_lane(tile, gsx, W, H);
  const f2 pxf = f2{(float)pm.px0, (float)(pm.px0 + 1)};
  const float pyf = (float)pm.py;
  const uint2 range = ranges[tile];

  f2 T = splat(1.0f);
  uint32_t last0 = 0, last1 = 0;
  f2 C[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ch++) C[ch] = splat(0.f);
  f2 invd = splat(0.f);
  bool done0 = !pm.in0, done1 = !pm.in1;

  Cand cur = load_cand(range.x + lane, range.y, point_list, gid, means2D, conic_o, depth, colors);
  for (uint32_t c0 = range.x; c0 < range.y; c0 += 64) {
    const Cand nxt = load_cand(c0 + 64 + lane, range.y, point_list, gid, means2D, conic_o, depth, colors);
    if (__ballot(!(done0 && done1)) == 0ull) break;  // every pixel of the tile has terminated
    const int n = (int)((range.y - c0) < 64u ? (range.y - c0) : 64u);
    const uint32_t jbase = c0 - range.x;
    for (int j = 0; j < n; j++) {
      const float gxs = rl(cur.gx, j), gys = rl(cur.gy, j);
      const float A = rl(cur.A, j), B = rl(cur.B, j), Cq = rl(cur.C, j), op = rl(cur.op, j);
      const f2 dx = gxs - pxf;
      const float dy = gys - pyf;
      const float e1 = B * dy, e0 = Cq * dy * dy;
      const f2 p = (A * dx - e1) * dx + e0;  // log2 of the Gaussian falloff
      const f2 alpha = __builtin_elementwise_min(op * exp2_2(p), splat(0.99f));
      bool v0 = !done0 && !(p.x > 0.0f) && !(alpha.x < 1.0f / 255.0f);
      bool v1 = !done1 && !(p.y > 0.0f) && !(alpha.y < 1.0f / 255.0f);
      const f2 test_T = T * (1.f - alpha);
      const bool t0 = v0 && test_T.x < 0.0001f, t1 = v1 && test_T.y < 0.0001f;  // not blended; pixel finished
      done0 = done0 || t0; done1 = done1 || t1;
      v0 = v0 && !t0; v1 = v1 && !t1;
      if (__ballot(v0 || v1) == 0ull) continue;
      const f2 wgt = sel(v0, v1, alpha * T, splat(0.f));
#pragma unroll
      for (int ch = 0; ch < NCH; ch++) C[ch] += rl(cur.ft[ch], j) * wgt;
      invd += rl(cur.ft[NCH], j) * wgt;
      T = sel(v0, v1, test_T, T);
      const uint32_t idx = jbase + (uint32_t)j + 1u;
      last0 = v0 ? idx : last0;
      last1 = v1 ? idx : last1;
    }
    cur = nxt;
  }
  const size_t HW = (size_t)H * W;
  if (pm.in0) {
    final_T[pm.id0] = T.x;
    n_contrib[pm.id0] = last0;
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) out_color[ch * HW + pm.id0] = C[ch].x + T.x * bg[ch];
    if (out_invdepth) out_invdepth[pm.id0] = invd.x;
  }
  if (pm.in1) {
    final_T[pm.id0 + 1] = T.y;
    n_contrib[pm.id0 + 1] = last1;
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) out_color[ch * HW + pm.id0 + 1] = C[ch].y + T.y * bg[ch];
    if (out_invdepth) out_invdepth[pm.id0 + 1] = invd.y;
  }
}

static inline uint32_t render_grid(int ntiles) {
  const uint32_t groups = ceil_div_u32((uint64_t)ntiles, BLK / 64);
  return ((groups + 7u) / 8u) * 8u;  // multiple of 8 for the XCD band mapping
}

void launch_render_fwd(const GeomWS& g, const BinWS& b, const ImgWS& im, int H, int W, const float* colors,
                       const float* bg, float* out_color, float* out_invdepth, hipStream_t s) {
  const int gsx = (W + SUBX - 1) / SUBX, gsy = (H + SUBY - 1) / SUBY, ntiles = gsx * gsy;
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
  const Pix pm = pixels_of_lane(tile, gsx, W, H);
  const f2 pxf = f2{(float)pm.px0, (float)(pm.px0 + 1)};
  const float pyf = (float)pm.py;
  const uint2 range = ranges[tile];
  const size_t HW = (size_t)H * W;
  const bool have_inv = dL_dinv != nullptr;

  f2 g[NCH];
  f2 ginv = splat(0.f), Dfinal = splat(0.f);
  uint32_t nc0 = 0, nc1 = 0;
#pragma unroll
  for (int ch = 0; ch < NCH; ch++) g[ch] = splat(0.f);
  if (pm.in0) {
    nc0 = n_contrib[pm.id0];
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
      g[ch].x = dL_dpix[ch * HW + pm.id0];
      Dfinal.x += g[ch].x * out_color[ch * HW + pm.id0];
    }
    if (have_inv) {
      ginv.x = dL_dinv[pm.id0];
      Dfinal.x += ginv.x * out_invdepth[pm.id0];
    }
  }
  if (pm.in1) {
    nc1 = n_contrib[pm.id0 + 1];
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
      g[ch].y = dL_dpix[ch * HW + pm.id0 + 1];
      Dfinal.y += g[ch].y * out_color[ch * HW + pm.id0 + 1];
    }
    if (have_inv) {
      ginv.y = dL_dinv[pm.id0 + 1];
      Dfinal.y += ginv.y * out_invdepth[pm.id0 + 1];
    }
  }
  // list entries past the last contributor of every pixel of the tile receive no gradient
  const uint32_t tile_last = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(nc0 > nc1 ? nc0 : nc1));

  const float kx = LN2 * 0.5f * W, ky = LN2 * 0.5f * H;  // d(pixel)/d(ndc) and the ln2 of the log2-domain conic
  f2 T = splat(1.0f), Dacc = splat(0.f);

  Cand cur = load_cand(range.x + lane, range.y, point_list, gid, means2D, conic_o, depth, colors);
  for (uint32_t c0 = range.x; c0 < range.y; c0 += 64) {
    const Cand nxt = load_cand(c0 + 64 + lane, range.y, point_list, gid, means2D, conic_o, depth, colors);
    const int n = (int)((range.y - c0) < 64u ? (range.y - c0) : 64u);
    const uint32_t jbase = c0 - range.x;
    int jn = n;  // entries of this chunk that can still matter
    if (tile_last < jbase + (uint32_t)n) jn = tile_last > jbase ? (int)(tile_last - jbase) : 0;
    unsigned long long written = 0ull;
    for (int j = 0; j < jn; j++) {
      const float gxs = rl(cur.gx, j), gys = rl(cur.gy, j);
      const float A = rl(cur.A, j), B = rl(cur.B, j), Cq = rl(cur.C, j), op = rl(cur.op, j);
      const f2 dx = gxs - pxf;
      const float dy = gys - pyf;
      const float e1 = B * dy, e0 = Cq * dy * dy;
      const f2 p = (A * dx - e1) * dx + e0;
      const f2 G = exp2_2(p);
      const f2 alpha = __builtin_elementwise_min(op * G, splat(0.99f));
      const uint32_t li = jbase + (uint32_t)j;
      const bool v0 = (li < nc0) && !(p.x > 0.0f) && !(alpha.x < 1.0f / 255.0f);
      const bool v1 = (li < nc1) && !(p.y > 0.0f) && !(alpha.y < 1.0f / 255.0f);
      if (__ballot(v0 || v1) == 0ull) continue;  // wave-uniform skip

      f2 gc = g[0] * rl(cur.ft[0], j);
#pragma unroll
      for (int ch = 1; ch < NCH; ch++) gc += g[ch] * rl(cur.ft[ch], j);
      if (have_inv) gc += ginv * rl(cur.ft[NCH], j);
      const f2 wgt = sel(v0, v1, alpha * T, splat(0.f));
      Dacc += gc * wgt;
      const f2 one_m = 1.f - alpha;
      const f2 rc = f2{__builtin_amdgcn_rcpf(one_m.x), __builtin_amdgcn_rcpf(one_m.y)};
      const f2 dLda = sel(v0, v1, T * gc - (Dfinal - Dacc) * rc, splat(0.f));
      T = sel(v0, v1, T * one_m, T);
      // v = G dL/dalpha (exp2 may overflow on pixels that skip this Gaussian: select, do not multiply by 0);
      // no zeroing when alpha was clamped (backward.cu:624)
      const f2 v = sel(v0, v1, G * dLda, splat(0.f));
      const f2 vdx = v * dx, vdxdx = vdx * dx;
      float c[REC];
      c[0] = v.x + v.y;          // M_1
      c[1] = vdx.x + vdx.y;      // M_dx
      c[2] = dy * c[0];          // M_dy
      c[3] = vdxdx.x + vdxdx.y;  // M_dxdx
      c[4] = dy * c[1];          // M_dxdy
      c[5] = dy * c[2];          // M_dydy
#pragma unroll
      for (int ch = 0; ch < NCH; ch++) {
        const f2 wg = wgt * g[ch];
        c[6 + ch] = wg.x + wg.y;
      }
      c[REC - 1] = 0.f;
      wave_sum11_lane63(c);
      const uint32_t slot = rlu(cur.slot, j);
      if (lane == 63) {
        // -a = 2A/log2e, -b = -B/log2e, -c = 2C/log2e
        const float m2x = op * kx * (2.f * A * c[1] - B * c[2]);
        const float m2y = op * ky * (2.f * Cq * c[2] - B * c[1]);
        const float ho = -0.5f * op;
        float4* dst = reinterpret_cast<float4*>(records + (size_t)slot * REC);
        dst[0] = make_float4(m2x, m2y, ho * c[3], ho * c[4]);
        dst[1] = make_float4(ho * c[5], c[0], c[6], c[7]);
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
  const int gsx = (W + SUBX - 1) / SUBX, gsy = (H + SUBY - 1) / SUBY, ntiles = gsx * gsy;
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
