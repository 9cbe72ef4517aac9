// common.h — shared definitions for the gfx950 rasterizer library (host + device).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

#include "eogs_rast.h"
#include "eogs_loss.h"
#include "eogs_optim.h"
#include "eogs_resample.h"
#include "eogs_knn.h"
#include "eogs_shade.h"
#include "eogs_tsdf.h"

#define NCH EOGS_RAST_CHANNELS  // 5 feature channels
#define TILE EOGS_RAST_TILE     // 16x16 pixel tiles: the reference's binning granularity (tile rect, radii)
#define SUBX 8                  // internal tile = SUBX x SUBY pixels = one wave64 (PPL pixels per lane, side by side);
#define SUBY 8                  // lists are built per internal tile. SUBX, SUBY divide TILE.
#define PPL (SUBX * SUBY / 64)  // pixels per lane
#define FX (TILE / SUBX)        // internal tiles per 16-px tile, horizontally / vertically
#define FY (TILE / SUBY)
#define MASK_MAX_SUBTILES 64    // Gaussians whose 16-px rect spans <= 64 internal tiles carry an exact hit mask
#define BLK 256                 // threads per workgroup everywhere (4 wave64)
#define NFEAT 6                 // staged per-Gaussian features: 5 colours + 1/depth
#ifndef REC
#define REC_ALT 8               // ... of an altitude-only render: 32 bytes {mean2D.x, .y, conic.a, opacity | conic.b, conic.c, colour3, -}
#define REC 12                  // floats per (tile,Gaussian) gradient record: 48 bytes, three 16-byte quarters, 11 floats used
#endif                          //   [0..3]   dL/dmean2D.x, .y (NDC units), dL/dconic.a, dL/dopacity
                                //   [4..7]   dL/dconic.b, dL/dconic.c, dL/dcolour0, dL/dcolour1
                                //   [8..11]  dL/dcolour2..4, -
                                // (64-byte slots with zero padding in rounds 1-2: a quarter of the backward's record traffic)
static_assert(REC == 12, "record quarters");
// Where quarter q (16 bytes) of the record in slot `slot` lives, in float4 units: record-major (a record's 48 bytes side by
// side). -DEOGS_REC_SOA=1 builds the quarter-major alternative — three planes of `plane` = capacity slots — tried in round 4 on
// the idea that gaussian_bwd's lanes (one Gaussian each, its records 48 x tiles-per-Gaussian bytes from the neighbour lane's)
// would coalesce better 16 x tiles apart: measured WORSE (gaussian_bwd 0.094 -> 0.111 ms, render_bwd 0.288 -> 0.300: a lane's
// three quarters then sit in three cache lines instead of one or two; profiles/r04_experiments/ab_rec_soa.txt). Kept as a switch.
#ifndef EOGS_REC_SOA
#define EOGS_REC_SOA 0
#endif
__host__ __device__ inline size_t rec_q(size_t slot, int q, size_t plane, int quarters) {
  return EOGS_REC_SOA ? (size_t)q * plane + slot : slot * (size_t)quarters + (size_t)q;
}

// ---- misc[] slots (u32) in the geometry workspace ----
// Inclusive prefix sum over the 64 lanes of a wave on the DPP path: row_shr 1, 2, 4, 8 inside each row of 16 lanes, then the
// rows' totals handed on with row_bcast15 / row_bcast31 — six VALU instructions. The __shfl_up form is six DEPENDENT
// ds_bpermute_b32 (an LDS round trip each, ~100 cycles of latency): in kernels that scan between workgroup barriers
// (pblock_scan, expand, the entry scatter, block_lists) that latency is on the critical path of every wave.
__device__ inline uint32_t wave_incl_scan_u32(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);  // row_shr:8: inclusive inside each row
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast15 -> rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast31 -> rows 2 and 3
  return (uint32_t)x;
}
// Wave-wide reductions on the same path (identity 0: unsigned max, sum, or): rotations inside the rows, row_bcast15 / 31 across
// them, the total read from lane 63 into an SGPR — every lane gets it.
template <class Op>
__device__ inline uint32_t wave_reduce_u32(uint32_t v, Op op) {
  int x = (int)v;
  x = (int)op((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x121, 0xF, 0xF, false));  // row_ror:1
  x = (int)op((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x122, 0xF, 0xF, false));  // row_ror:2
  x = (int)op((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x124, 0xF, 0xF, false));  // row_ror:4
  x = (int)op((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, false));  // row_ror:8: every lane = its row
  x = (int)op((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false));  // row_bcast15 -> rows 1, 3
  x = (int)op((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false));  // row_bcast31 -> rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}
__device__ inline uint32_t wave_max_u32_dpp(uint32_t v) { return wave_reduce_u32(v, [](uint32_t a, uint32_t b) { return a > b ? a : b; }); }
__device__ inline uint32_t wave_sum_u32_dpp(uint32_t v) { return wave_reduce_u32(v, [](uint32_t a, uint32_t b) { return a + b; }); }
__device__ inline uint32_t wave_or_u32_dpp(uint32_t v) { return wave_reduce_u32(v, [](uint32_t a, uint32_t b) { return a | b; }); }
__device__ inline unsigned long long wave_incl_scan_u64(unsigned long long v) {
  // four 16-bit pieces, each scanned exactly in 32 bits (64 lanes x 65535 < 2^32), recombined with 64-bit adds
  const uint32_t a = wave_incl_scan_u32((uint32_t)(v & 0xFFFFu)), b = wave_incl_scan_u32((uint32_t)((v >> 16) & 0xFFFFu));
  const uint32_t c = wave_incl_scan_u32((uint32_t)((v >> 32) & 0xFFFFu)), d = wave_incl_scan_u32((uint32_t)(v >> 48));
  return (unsigned long long)a + ((unsigned long long)b << 16) + ((unsigned long long)c << 32) + ((unsigned long long)d << 48);
}
#define MISC_TOTAL_LO 0  // sum of tiles_touched (u64, lo/hi)
#define MISC_TOTAL_HI 1
#define MISC_ERR 2       // bit0: altitude > 200
#define MISC_TAG 3       // a constant the count scan writes with the counts (MISC_TAG_VALUE): lets a host that watches a
                         // mirrored copy arrive (eogs_rast_mirror_token) see that the FIRST 16 bytes are there as a whole
#define MISC_TAG_VALUE 0x0C0FFEE0u
#define MISC_KEY_MAX 4   // max depth key over listed Gaussians
#define MISC_KEY_NMIN 5  // max of ~key = ~min depth key (zero-initialised like the rest of misc)
#define MISC_MACRO_LO 6  // number of (macro block, Gaussian) list entries (u64, lo/hi)
#define MISC_MACRO_HI 7
#define MISC_OPW_LO 8    // sum over listed (tile, Gaussian) pairs of round(64 opacity) (u64, lo/hi): mean pair opacity
#define MISC_OPW_HI 9
// Flag-free records (quad backward on per-tile lists only). Where no tile comes near saturation every listed pair is walked
// by the backward, so render_bwd_quad_kernel writes the record of EVERY pair (zeros where nothing contributed), walks every
// chunk of a list whatever its pixels' last contributors are, and nobody writes or reads the one-byte live flags — 4.2 M
// scattered byte stores at the headline, each dirtying a 64-byte line of its own (3 % of the dominant kernel). Correct for
// any scene (a saturated tile only costs the walk over its dead entries); chosen where the mean over tiles of the listed
// pairs' summed opacity stays below NOFLAG_K (saturation needs ~9 along ONE pixel, which sees a third of its tile's list at
// a fraction of its opacity). Forced on, it still pays at a mean of 53 (1 M Gaussians at opacity 0.1: render_bwd -1.5 %,
// gaussian_bwd -1.4 %; opacity 0.05, mean 23: -3 % / -3 %) and costs +75 % at 121 (opacity 0.2, where tiles saturate):
// profiles/r05_ab_noflag.txt. 24 keeps a factor of two to where the gain ends.
// Both kernels evaluate this same expression on the same words of `misc` (written by pblock_scan_kernel).
#define NOFLAG_K 24.0f
__host__ __device__ inline bool noflag_scene(uint32_t opw_lo, uint32_t opw_hi, int W, int H) {
  const float opw = (float)opw_lo + 4294967296.0f * (float)opw_hi;  // sum over listed pairs of round(64 opacity)
  const float ntiles8 = (float)((W + 7) / 8) * (float)((H + 7) / 8);
  return opw <= 64.0f * NOFLAG_K * ntiles8;
}
#define MISC_READBACK 10 // words copied to the host by forward_prepare
#define MISC_DEPTH_PASSES 10  // 8-bit digits that cover the varying bits of the listed Gaussians' depth keys (0..4)
#define MISC_WORDS 64

// Tile schedule of the render launches (binning.hip tile_sched_body). It is computed by one workgroup that
// holds every block's pair count: images up to 4096 blocks of 32 x 32 px (2048^2); larger ones keep the band mapping.
#define SCHED_MAX_BLOCKS 4096u
// An XCD's sequence holds up to 1.25 x its fair share of the blocks (+1): the equal-work cut gives an XCD whose blocks are
// light more of them; what exceeds the capacity spills into the other XCDs' free places.
static inline uint32_t sched_capacity(uint32_t nblocks) { return (nblocks + 7u) / 8u + (nblocks + 31u) / 32u + 1u; }
static inline uint32_t sched_flags() {  // EOGS_SCHED_FLAGS: experiment switch of tile_sched_body (0x20 = no saturation cap), default 0
  static const uint32_t v = [] {
    const char* e = getenv("EOGS_SCHED_FLAGS");
    return (uint32_t)(e ? strtoul(e, nullptr, 0) : 0ul) & 0xE0u;
  }();
  return v;
}
// EOGS_TILE_SCHED=0 switches the schedule off (A/B: the XCD band mapping of rounds 1-3)
static inline bool sched_enabled() {
  static const bool v = [] {
    const char* e = getenv("EOGS_TILE_SCHED");
    return !(e && e[0] == '0');
  }();
  return v;
}

// ---- radix sort geometry ----
#define SORTP_ITEMS 8    // generic u32 key + u32 payload sort (knn.hip's Morton order): 2048 keys per workgroup
#ifndef SORTE_TILE
#define SORTE_TILE 8192
#endif
// SORTE_TILE: block sort of the list entries (16-byte items): entries per workgroup (binning.hip ES_TILE)
#define SORTE_MAXBINS 4096  // ... and the widest digit of one pass (ES_MAXBITS)
#define EXPAND_ITEMS 1   // expand: 256 Gaussians (one preprocess workgroup) per workgroup
#define MAX_BLOCKS (1u << 16)  // 32 x 32-pixel blocks per image (their id is the low half of an entry's sort key)

// ---- which internal tiles a Gaussian is listed in (GeomWS::binfo) ----
#define BK_RECT 0u
#define BK_MASK 1u
#define BK_SPANS 2u
// Per-Gaussian constants of the row-span test, computed once by preprocess and stored (GeomWS::bext): expand
// re-evaluates row_span() on the same bits, so the emitted tiles are exactly the counted ones.
// Ellipse q(u,v) = a u^2 + 2 b u v + c v^2 <= tau_m around (gx,gy), det = ac - b^2:
struct SpanParams {
  float gx, gy;  // centre (pixels)
  float ex, ey;  // half extents sqrt(tau_m c/det), sqrt(tau_m a/det)
  float boa, boc;  // b/a, b/c
  float ta, da;  // tau_m/a, det/a^2:  u(v) = -(b/a) v +- sqrt(ta - da v^2)
};
// (v_rcp_f32 / v_sqrt_f32 instead of IEEE divisions and square roots — ten instructions each —: what these constants decide is
// which tiles a Gaussian LISTS, behind margins of 1e-3 relative + 1e-2 px (row_span), five orders above the 1-ulp error of the fast
// forms; preprocess counts and expand emits with this same code on the same bits, so count and emission cannot disagree.)
__device__ inline SpanParams span_params(float gx, float gy, float a, float b, float c, float tau_m) {
  const float det = a * c - b * b;
  const float ra = __builtin_amdgcn_rcpf(a), rc = __builtin_amdgcn_rcpf(c), rdet = __builtin_amdgcn_rcpf(det);
  SpanParams p;
  p.gx = gx; p.gy = gy;
  p.ex = __builtin_amdgcn_sqrtf(tau_m * c * rdet); p.ey = __builtin_amdgcn_sqrtf(tau_m * a * rdet);
  p.boa = b * ra; p.boc = b * rc;
  p.ta = tau_m * ra; p.da = det * (ra * ra);
  return p;
}
// Internal tiles of row sy (pixel centres y in [SUBY sy, SUBY sy + SUBY-1]) whose continuous block intersects the
// ellipse: a column range [c0,c1) inside [sx0,sx1). The ellipse cut by the row band is convex, so its x-projection is
// one interval [gx+ul, gx+ur]; u_max(v) = -(b/a) v + sqrt(ta - da v^2) is concave, hence maximal at the rightmost
// point's v = -(b/c) ex clamped to the band (likewise for the left end). Same predicate as block_hit() in
// preprocess.hip, in closed form per row.
__device__ inline void row_span(const SpanParams& p, int sy, int sx0, int sx1, int& c0, int& c1) {
  c0 = c1 = sx0;
  const float v0 = (float)(sy * SUBY) - p.gy;
  const float w0 = fmaxf(v0, -p.ey), w1 = fminf(v0 + (float)(SUBY - 1), p.ey);
  if (!(w0 <= w1)) return;  // the band misses the ellipse
  const float vr = fminf(fmaxf(-p.boc * p.ex, w0), w1), vl = fminf(fmaxf(p.boc * p.ex, w0), w1);
  const float ur = -p.boa * vr + __builtin_amdgcn_sqrtf(fmaxf(p.ta - p.da * vr * vr, 0.f));
  const float ul = -p.boa * vl - __builtin_amdgcn_sqrtf(fmaxf(p.ta - p.da * vl * vl, 0.f));
  const float xr = p.gx + ur + (1e-3f * fabsf(ur) + 1e-2f), xl = p.gx + ul - (1e-3f * fabsf(ul) + 1e-2f);
  // internal tile j covers [SUBX j, SUBX j + SUBX-1]
  const float j0 = fmaxf(ceilf((xl - (float)(SUBX - 1)) * (1.f / SUBX)), (float)sx0);
  const float j1 = fminf(floorf(xr * (1.f / SUBX)) + 1.f, (float)sx1);
  if (j0 < j1) {
    c0 = (int)j0;
    c1 = (int)j1;
  }
}

// Can this Gaussian reach alpha >= 1/255 anywhere in the pixel block [x0,x1] x [y0,y1]?
// alpha = o exp(-q/2), q(d) = a dx^2 + 2 b dx dy + c dy^2 (d = centre - pixel), so alpha >= 1/255 <=> q <= tau,
// tau = 2 ln(255 o). The minimum of the convex q over the block is 0 if the centre is inside, otherwise it lies on
// the edges that face the centre: minimise q along x = clamp(gx) and along y = clamp(gy), the free coordinate
// clamped to the block. The continuous block contains the pixel centres, so q_min(block) <= q(pixel): dropping the
// block when q_min > tau (plus a margin far above the fp32 rounding of the renderer's `power`) never drops a pixel
// that would blend this Gaussian (forward.cu:374-376 skips alpha < 1/255). NaNs fail the comparison -> kept.
// b_c = b/c and b_a = b/a are per-Gaussian constants (the unclamped minimiser on an edge).
__device__ inline bool block_hit(float gx, float gy, float a, float b, float c, float b_c, float b_a, float tau_m,
                                 float x0, float y0, float x1, float y1) {
  // clamps as v_med3_f32 (x0 <= x1, y0 <= y1): one instruction each instead of v_max + v_min, and no canonicalised copies of
  // the box corners for the compiler to keep in registers (a NaN operand leaves the smaller bound, as fminf(fmaxf()) did)
  const float cx = __builtin_amdgcn_fmed3f(gx, x0, x1), cy = __builtin_amdgcn_fmed3f(gy, y0, y1);
  const float dxe = gx - cx, dye = gy - cy;
  const float py = __builtin_amdgcn_fmed3f(gy + b_c * dxe, y0, y1);  // edge x = cx, free y
  const float dy1 = gy - py;
  const float q1 = a * dxe * dxe + 2.f * b * dxe * dy1 + c * dy1 * dy1;
  const float pxs = __builtin_amdgcn_fmed3f(gx + b_a * dye, x0, x1);  // edge y = cy, free x
  const float dx2 = gx - pxs;
  const float q2 = a * dx2 * dx2 + 2.f * b * dx2 * dye + c * dye * dye;
  return !(fminf(q1, q2) > tau_m);
}

// ---- blocks: the unit of the sorted lists, chosen per forward ----
// Lists are sorted per M x M block of internal tiles. M = 1: one list per internal tile (small footprints: every list
// entry is used by the wave that reads it). M = BLOCK_BIG = 4 (32 x 32 pixels): every list entry carries an M*M-bit
// sub-mask (upper half of its 32-bit sort key) saying which internal tiles of the block the Gaussian is listed in; a
// render wave (one internal tile) scans its block's list and keeps the entries whose bit is set (ballot + prefix
// compaction into LDS), so the sort moves one entry per (block, Gaussian) instead of one per (tile, Gaussian) — 4-7x
// fewer for footprints of tens of pixels, where binning otherwise costs as much as rendering. forward_prepare picks
// the mode from the two pair counts it reads back (api.hip) and hands it on inside num_rendered.
#define BLOCK_BIG 4
#define EOGS_BLOCK_SWITCH 20  // listed internal tiles per Gaussian (average) above which a forward uses BLOCK_BIG
#define EOGS_DEPTH_SWITCH 0  // (pairs per tile) x (mean pair opacity) above which a forward uses BLOCK_BIG; 0 = never
                             // (round 3: per-tile lists cost one 8-byte item per pair to build, the criterion no longer pays)
#define MACRO_KEY_BITS 16  // M > 1: block id in the low half of the key, sub-mask in the high half
static_assert(BLOCK_BIG * BLOCK_BIG <= 16, "the sub-mask lives in the upper 16 bits of the sort key");

// Walks the macro row MY of a Gaussian's listing (kind, mask m or span constants sp, internal-tile rect
// [sx0,sx1) x [sy0,sy1)) and calls emit(MX, sub) for every block with a non-empty sub-mask, left to right.
// Used by preprocess (counting) and expand (emission): same code, same bits, same result.
template <int MACRO, class Emit>
__device__ inline void walk_macro_row(uint32_t kind, unsigned long long m, const SpanParams& sp, int sx0, int sy0, int sx1,
                                      int sy1, int MY, Emit&& emit) {
  const int sw = sx1 - sx0;
  int c0[MACRO], c1[MACRO];
  unsigned long long w[MACRO];
  int lo = 0x7FFFFFFF, hi = -1;  // columns with any hit in this macro row
#pragma unroll
  for (int dy = 0; dy < MACRO; dy++) {
    const int fy = MY * MACRO + dy;
    c0[dy] = c1[dy] = 0;
    w[dy] = 0ull;
    if (fy < sy0 || fy >= sy1) continue;
    if (kind == BK_MASK) {  // sw * sh <= 64: row fy is sw bits of m
      w[dy] = (m >> ((fy - sy0) * sw)) & (sw >= 64 ? ~0ull : ((1ull << sw) - 1ull));
      if (w[dy]) {
        lo = min(lo, sx0 + (int)__builtin_ctzll(w[dy]));
        hi = max(hi, sx0 + 63 - (int)__builtin_clzll(w[dy]));
      }
    } else {
      if (kind == BK_SPANS) row_span(sp, fy, sx0, sx1, c0[dy], c1[dy]);
      else { c0[dy] = sx0; c1[dy] = sx1; }
      if (c1[dy] > c0[dy]) {
        lo = min(lo, c0[dy]);
        hi = max(hi, c1[dy] - 1);
      }
    }
  }
  if (hi < 0) return;
  for (int MX = lo / MACRO; MX <= hi / MACRO; MX++) {
    const int X0 = MX * MACRO;
    uint32_t sub = 0;
#pragma unroll
    for (int dy = 0; dy < MACRO; dy++) {
      uint32_t nib;
      if (kind == BK_MASK) {
        const int off = X0 - sx0;
        nib = (uint32_t)((off >= 0 ? (w[dy] >> off) : (w[dy] << (-off))) & ((1ull << MACRO) - 1ull));
      } else {
        const int a = max(c0[dy], X0) - X0, b = min(c1[dy], X0 + MACRO) - X0;
        nib = b > a ? (((1u << (b - a)) - 1u) << a) : 0u;
      }
      sub |= nib << (dy * MACRO);
    }
    if (sub) emit(MX, sub);
  }
}

// What walk_macro_row() emits for a listing WITHOUT a hit mask (BK_SPANS / BK_RECT), counted without walking the blocks: `ent` =
// blocks with a non-empty sub-mask, `fine` = listed internal tiles of the macro row. Block MX gets a bit from tile row dy iff the
// row's column interval [c0, c1) meets [MX M, MX M + M), i.e. iff MX lies in [c0 / M, (c1 - 1) / M]: the blocks are the union of
// at most M such intervals (a 64-bit mask when the macro row spans at most 64 blocks — images up to 2048 px wide —, the walk
// itself otherwise), the tiles the sum of the rows' lengths. The same row_span() on the same bits as the walk: the same count, at
// ~40 instead of ~45 instructions per BLOCK of the row (a wave pays its widest footprint: preprocess_fwd_kernel).
template <int MACRO>
__device__ inline void count_macro_row(uint32_t kind, const SpanParams& sp, int sx0, int sy0, int sx1, int sy1, int MY,
                                       uint32_t& ent, uint32_t& fine) {
  int c0[MACRO], c1[MACRO];
  int lo = 0x7FFFFFFF, hi = -1;
#pragma unroll
  for (int dy = 0; dy < MACRO; dy++) {
    const int fy = MY * MACRO + dy;
    c0[dy] = c1[dy] = 0;
    if (fy < sy0 || fy >= sy1) continue;
    if (kind == BK_SPANS) row_span(sp, fy, sx0, sx1, c0[dy], c1[dy]);
    else { c0[dy] = sx0; c1[dy] = sx1; }
    if (c1[dy] > c0[dy]) {
      lo = min(lo, c0[dy]);
      hi = max(hi, c1[dy] - 1);
    }
  }
  if (hi < 0) return;
  const int B0 = lo / MACRO, B1 = hi / MACRO;
  if (B1 - B0 < 64) {
    unsigned long long m = 0ull;
#pragma unroll
    for (int dy = 0; dy < MACRO; dy++) {
      if (c1[dy] > c0[dy]) {
        const int a = c0[dy] / MACRO - B0, len = (c1[dy] - 1) / MACRO - c0[dy] / MACRO + 1;
        m |= (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << a;
        fine += (uint32_t)(c1[dy] - c0[dy]);
      }
    }
    ent += (uint32_t)__popcll(m);
  } else {
    walk_macro_row<MACRO>(kind, 0ull, sp, sx0, sy0, sx1, sy1, MY, [&](int, uint32_t sub) {
      ent++;
      fine += (uint32_t)__popc(sub);
    });
  }
}

static inline size_t ws_align(size_t x) { return (x + 255u) & ~(size_t)255u; }

template <typename T>
static inline size_t ws_carve(char* base, size_t off, T*& p, size_t count) {
  off = ws_align(off);
  p = base ? reinterpret_cast<T*>(base + off) : nullptr;
  return off + count * sizeof(T);
}

static inline uint32_t ceil_div_u32(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// Geometry workspace: everything that is O(P) and read again by backward (plus the per-block bookkeeping of one
// forward, a fixed 0.8 MB). SoA, every array 256-B aligned.
struct GeomWS {
  float4* packed;       // 4 x float4 = one 64-byte line per Gaussian, everything the render kernels gather:
                        //   {gx, gy, A, B} {C, opacity, f0, f1} {f2, f3, f4, 1/depth} {pad}
                        //   with the conic pre-scaled by log2 e: A = -a log2e/2, B = b log2e, C = -c log2e/2
  uint4* binfo;         // 2 x uint4 = 32 bytes per Gaussian in TWO PLANES of P records ([k] and [P + k]: the backward reads the second alone), everything binning needs, written once by preprocess:
                        //   [0] = {sx0 | sx1<<16, sy0 | sy1<<16 (internal-tile rect, clipped), mask lo, mask hi}
                        //   [1] = {tiles (internal tiles listed, 0 = none), lpre, depth key, kind | block entries << 2}
                        //   kind BK_MASK : mask bit (sy-sy0)*(sx1-sx0) + (sx-sx0) set <=> internal tile (sx,sy) can reach
                        //                  alpha >= 1/255 (rects of <= 64 internal tiles)
                        //   kind BK_SPANS: larger rects; the listed tiles of row sy are row_span(bext[id], sy) (below)
                        //   kind BK_RECT : every internal tile of the rect (NaN opacity only)
                        //   lpre = exclusive prefix of `tiles` inside the Gaussian's preprocess workgroup (256 Gaussians)
                        //   depth key = bits of (float)(200 - altitude), the reference's sort key (forward.cu:267)
  float4* bext;         // 2 x float4 = 32 bytes per Gaussian, written for BK_SPANS only: SpanParams
  uint32_t* pbkey;      // per preprocess workgroup: {max depth key, max ~key, list entries, sum of tiles * round(64 opacity)}
  uint32_t* pblock;     // per preprocess workgroup: listed tiles, then (pblock_scan_kernel) exclusive prefix over workgroups.
                        // record slot of (Gaussian i, its q-th tile) = pblock[i/256] + lpre[i] + q: records are laid out
                        // in Gaussian-id order, so gaussian_bwd streams them
  uint32_t* pblockE;    // the same prefix for the list entries (one per listed 32 x 32-px block): where a preprocess
                        // workgroup's entries start in the id-ordered entry array
  uint32_t* bcount;     // [MAX_BLOCKS] per 32 x 32-px block: list entries (their sum over the blocks before b = where
                        // block b's entries start in the block-sorted entry array)
  uint32_t* bpairs;     // [MAX_BLOCKS] per block: listed (internal tile, Gaussian) pairs
  uint32_t* sched;      // [16] tile schedule of the render launches (binning.hip tile_sched_body): [0..8) blocks in XCD x's sequence
  uint32_t* where;      // [3 x SCHED_MAX_BLOCKS] block b's place in that schedule: XCD << 24 | position in the XCD's sequence;
                        // behind them {list entries, pairs} of the blocks before b (two words per block: block_lists_kernel's start)
  uint32_t* misc;       // MISC_WORDS
  float* vmpart;        // backward: per-workgroup partials of the 18 camera-gradient sums [ceil(P/256)][18]
  uint32_t nblkE;
  size_t bytes;
};

static inline GeomWS geom_layout(char* base, int P) {
  GeomWS g;
  size_t n = (size_t)P, o = 0;
  g.nblkE = ceil_div_u32(n, BLK * EXPAND_ITEMS);
  o = ws_carve(base, o, g.packed, n * 4);
  o = ws_carve(base, o, g.binfo, n * 2);
  o = ws_carve(base, o, g.bext, n * 2);
  o = ws_carve(base, o, g.pblock, (size_t)g.nblkE + 1);
  o = ws_carve(base, o, g.pblockE, (size_t)g.nblkE + 1);
  o = ws_carve(base, o, g.pbkey, (size_t)g.nblkE * 4);
  o = ws_carve(base, o, g.bcount, (size_t)MAX_BLOCKS);
  o = ws_carve(base, o, g.bpairs, (size_t)MAX_BLOCKS);
  o = ws_carve(base, o, g.sched, (size_t)16);
  o = ws_carve(base, o, g.where, (size_t)3 * SCHED_MAX_BLOCKS);
  o = ws_carve(base, o, g.misc, MISC_WORDS);
  o = ws_carve(base, o, g.vmpart, (size_t)g.nblkE * 18);
  g.bytes = ws_align(o) + 256;  // slack so a base that is only 1-aligned still fits after rounding
  return g;
}

// Entry-sort workspace: the list entries (one per (32 x 32-px block, Gaussian)) of ONE forward, ping-pong, plus the radix
// histograms. An entry is 16 bytes {key = block id | sub-mask << 16, depth key, Gaussian id, first record slot}.
// Transient: dead once the forward has built its lists, so it lives in the caller's scratch buffer (shared by every
// forward on a stream) and not in a workspace autograd keeps. Its capacity is fixed before the entry count is known
// (ent_cap: six entries per Gaussian); a forward with more entries sorts inside its binning workspace instead.
struct SortWS {
  uint4* entA;
  uint4* entB;
  uint32_t* hist;    // [bins][nblk] entries per digit and workgroup -> exclusive prefix over the workgroups
  uint32_t* histp;   // [bins][nblk] listed pairs per digit and workgroup (single-pass sort)
  uint32_t* dtotal;  // [bins] (two-pass sort; a single pass writes its totals to GeomWS::bcount)
  uint32_t cap, nblk;
  size_t bytes;
};
static inline uint32_t ent_cap(int P) {
  const uint64_t c = 6ull * (uint64_t)(P < 0 ? 0 : P);
  return c < 4096ull ? 4096u : (c > 0x07FFFFFFull ? 0x07FFFFFFu : (uint32_t)c);
}
static inline SortWS sort_layout(char* base, uint32_t cap) {
  SortWS w;
  size_t o = 0;
  w.cap = cap;
  w.nblk = ceil_div_u32(cap, (uint64_t)SORTE_TILE);
  o = ws_carve(base, o, w.entA, (size_t)cap);
  o = ws_carve(base, o, w.entB, (size_t)cap);
  o = ws_carve(base, o, w.hist, (size_t)SORTE_MAXBINS * (w.nblk ? w.nblk : 1));
  o = ws_carve(base, o, w.histp, (size_t)SORTE_MAXBINS * (w.nblk ? w.nblk : 1));
  o = ws_carve(base, o, w.dtotal, SORTE_MAXBINS);
  w.bytes = ws_align(o) + 256;
  return w;
}

// Binning workspace: everything that is O(R) (R = number of (tile,Gaussian) pairs) and read again by backward.
struct BinWS {
  uint2* point_list;     // the sorted lists. block 1: per (internal tile, Gaussian) pair {Gaussian id, record slot};
                         // BLOCK_BIG: per (block, Gaussian) entry {Gaussian id, record slot of its first listed tile}
  uint32_t* sorted_keys; // BLOCK_BIG only: per entry {block id | sub-mask << 16}
  float* records;   // backward scratch: REC floats per record slot (Gaussian-id order, see GeomWS::pblock)
  uint8_t* live;    // backward scratch: 1 = the pair's record was written (dead pairs are never touched)
  uint8_t* qmask;   // block 1 only: per list entry the 4-bit mask of the tile's 4 x 4-px quads the Gaussian can reach, written by
                    // the quad forward for the chunks it walked and read by the quad backward instead of being computed again
  SortWS sort;      // entry sort of a forward whose entries did not fit the caller's scratch (nr_sorted(R) == 0)
  int block;
  uint32_t cap_slots, cap_entries;  // what the token sized these arrays for (a forward that needs more has built no lists)
  size_t bytes;
};

static inline int ceil_log2_u32(uint32_t n) {  // smallest b with (1<<b) >= n
  int b = 0;
  while (((uint64_t)1 << b) < n) b++;
  return b;
}

// num_rendered as handed across the C-ABI packs what a forward decided: the record slots (one per listed internal tile,
// backward scratch) in bits 0..30, the list entries (one per listed 32 x 32-px block) in bits 32..58, bit 59 = an altitude-only
// forward (EOGS_FLAG_ALT_ONLY: per-tile lists, the quad kernels' one-channel variants, 32-byte gradient records), bit 61 = the
// entries were sorted in the caller's scratch by forward_prepare, bit 62 = the render kernels read block lists
// (BLOCK_BIG) instead of per-tile lists, bit 31 = a block holds 2800 ... 6000 entries on average (block_lists_kernel's
// 8-item build; a property of the forward the token was counted on, carried over into capacity tokens), bit 60 = the forward's
// lists are shallow in opacity (mean list length x mean pair opacity <= GB_WIDE_DEPTH: most listed pairs are live, and the
// backward launches the build of its per-Gaussian kernel that keeps eight records in flight per lane — speed only, every build
// computes the same bits; carried over into capacity tokens. ABI 3-7 used the bit for "the back-to-front backward": every
// backward walks back to front since ABI 8, and the hint lived in a per-process table beside the token).
static inline uint32_t nr_slots(int64_t R) { return (uint32_t)((uint64_t)R & 0x7FFFFFFFull); }
static inline uint32_t nr_entries(int64_t R) { return (uint32_t)(((uint64_t)R >> 32) & 0x07FFFFFFull); }
static inline int nr_alt(int64_t R) { return (int)(((uint64_t)R >> 59) & 1ull); }  // altitude-only forward (EOGS_FLAG_ALT_ONLY)
static inline int nr_shallow(int64_t R) { return (int)(((uint64_t)R >> 60) & 1ull); }
static inline int nr_sorted(int64_t R) { return (int)(((uint64_t)R >> 61) & 1ull); }
static inline int nr_block(int64_t R) { return (((uint64_t)R >> 62) & 1ull) ? BLOCK_BIG : 1; }
static inline int nr_wide(int64_t R) { return (int)(((uint64_t)R >> 31) & 1ull); }
static inline int64_t nr_pack(uint32_t slots, uint32_t entries, int block, int sorted, int wide, int shallow, int alt = 0) {
  return (int64_t)(((uint64_t)(block > 1) << 62) | ((uint64_t)(sorted != 0) << 61) | ((uint64_t)(shallow != 0) << 60) |
                   ((uint64_t)(alt != 0) << 59) | ((uint64_t)(entries & 0x07FFFFFFu) << 32) | ((uint64_t)(wide != 0) << 31) | slots);
}
static inline uint32_t macro_grid_x(int W, int M) { return (uint32_t)(((W + SUBX - 1) / SUBX + M - 1) / M); }
static inline uint32_t macro_grid_y(int H, int M) { return (uint32_t)(((H + SUBY - 1) / SUBY + M - 1) / M); }
// the entry sort orders by block id: `passes` stable radix passes of `bits` bits each
static inline void block_sort_geometry(int H, int W, int& passes, int& bits) {
  const uint32_t nb = macro_grid_x(W, BLOCK_BIG) * macro_grid_y(H, BLOCK_BIG);
  const int tb = ceil_log2_u32(nb) < 1 ? 1 : ceil_log2_u32(nb);
  passes = tb <= 12 ? 1 : 2;  // one counting pass up to 4096 blocks (SORTE_MAXBINS)
  bits = (tb + passes - 1) / passes;
}

static inline BinWS bin_layout(char* base, int H, int W, int64_t R) {
  (void)H; (void)W;
  BinWS b;
  const size_t ne = (size_t)nr_entries(R), nslots = (size_t)nr_slots(R);
  size_t o = 0;
  b.block = nr_block(R);
  b.cap_slots = (uint32_t)nslots;
  b.cap_entries = (uint32_t)ne;
  o = ws_carve(base, o, b.point_list, b.block > 1 ? ne : nslots);
  o = ws_carve(base, o, b.sorted_keys, b.block > 1 ? ne : (size_t)0);
  o = ws_carve(base, o, b.records, nslots * REC);
  o = ws_carve(base, o, b.live, nslots);
  o = ws_carve(base, o, b.qmask, b.block > 1 ? (size_t)0 : nslots);
  b.sort = sort_layout(nullptr, 0);
  if (!nr_sorted(R) && ne) {
    o = ws_align(o);
    b.sort = sort_layout(base ? base + o : nullptr, (uint32_t)ne);
    o += b.sort.bytes;
  }
  b.bytes = ws_align(o) + 256;
  return b;
}

// Image workspace: O(H*W) + O(tiles).
struct ImgWS {
  uint2* ranges;       // per block (M x M internal tiles, M chosen per forward) [start,end) into point_list
  float* final_T;      // transmittance after the last blended Gaussian
  uint32_t* n_contrib; // entries at positions >= this — positions in the list the render wave walks: the tile's, or its block's —
                       // take no part at the pixel: 1 + the position of its last blended entry (tile / block forward kernels)
                       // or the position of its stop entry, 0xFFFFFFFF if it never stopped (quad forward)
  uint4* desc;         // one descriptor per dispatched render workgroup (nullptr without a tile schedule): XCD x's i-th workgroup
                       // reads desc[x * 16 sched_lg + i] = {tile tx | ty << 16 (0xFFFFFFFF: outside the image), list begin, list end, -},
                       // written by block_lists_kernel at its block's place in the schedule (GeomWS::where)
  uint32_t sched_lg;   // capacity of one XCD's sequence in blocks: the render grid is 8 x 16 x sched_lg workgroups
  size_t bytes;
};

static inline ImgWS img_layout(char* base, int H, int W) {
  ImgWS im;
  size_t n = (size_t)H * W, o = 0;
  size_t T = (size_t)macro_grid_x(W, 1) * macro_grid_y(H, 1);  // enough for either block size
  const uint32_t nblocks = macro_grid_x(W, BLOCK_BIG) * macro_grid_y(H, BLOCK_BIG);
  o = ws_carve(base, o, im.ranges, T);
  o = ws_carve(base, o, im.final_T, n);
  o = ws_carve(base, o, im.n_contrib, n);
  im.desc = nullptr;
  im.sched_lg = 0;
  if (nblocks <= SCHED_MAX_BLOCKS && sched_enabled()) {
    im.sched_lg = sched_capacity(nblocks);
    o = ws_carve(base, o, im.desc, (size_t)8 * 16 * im.sched_lg);
  }
  im.bytes = ws_align(o) + 256;
  return im;
}

static inline char* ws_base(const void* p) {  // round the caller's pointer up to 256 B
  return reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 255u) & ~(uintptr_t)255u);
}

// ---- launch entry points implemented in the .hip files (host functions) ----
struct FwdPrepArgs {
  int P, H, W;
  const float *means3D, *scales, *rotations, *cov3D_precomp, *opacities, *colors, *viewmatrix;
  float scale_modifier;
  bool antialiasing;
  int* radii;
  bool raw;  // EOGS_FLAG_RAW_PARAMS
  const float* alt_affine;
};
void launch_preprocess_fwd(const FwdPrepArgs& a, const GeomWS& g, hipStream_t s);
// exclusive scans of the per-workgroup pair / entry counts + the totals and depth-key range the host reads back (misc)
void launch_pblock_scan(const GeomWS& g, int P, hipStream_t s);
// expand the list entries in Gaussian-id order, sort them by block id (stable), find every block's range and pair count.
// Reads the entry count from g.misc on the device: does nothing when it exceeds w.cap (so it can be queued before the host
// knows the count).
void launch_entry_sort(const GeomWS& g, const SortWS& w, int P, int H, int W, hipStream_t s);
// which of w.entA / w.entB holds the block-sorted entries after launch_entry_sort
static inline bool entry_sort_result_in_A(int H, int W) {
  int passes, bits;
  block_sort_geometry(H, W, passes, bits);
  return (passes & 1) == 0;
}
// per block: depth order (stable radix on the depth keys, misc[MISC_DEPTH_PASSES] 8-bit digits) and, with per-tile lists,
// the split of the block's entries into its internal tiles' lists; writes b.point_list (+ b.sorted_keys), im.ranges, clears
// b.live
void launch_block_lists(const GeomWS& g, const SortWS& w, const BinWS& b, const ImgWS& im, int P, int H, int W, int64_t R,
                        hipStream_t s);
void launch_render_fwd(const GeomWS& g, const BinWS& b, const ImgWS& im, int P, int H, int W, int64_t R,
                       const float* bg, float* out_color, float* out_invdepth, hipStream_t s);
void launch_render_bwd(const GeomWS& g, const BinWS& b, const ImgWS& im, int P, int H, int W, int64_t R,
                       const float* dL_dcolor, const float* dL_dinvdepth, const float* bg, bool raw, hipStream_t s);
// which render kernel a forward / backward runs: 0 = one list per tile, 1 = block lists, 2 = quad sub-lists (render.hip)
int render_fwd_variant(int block, int64_t R, int P);
int render_bwd_variant(int block, int64_t R, int P);  // 0 tile, 1 block, 2 quad, 6 quad of an altitude-only render
int render_bwd_noflag_ok(int block, int64_t R, int P);  // that backward writes flag-free records: 0 no, 1 where noflag_scene() holds, 3 always
struct GaussBwdArgs {
  int P, H, W;
  const float *means3D, *scales, *rotations, *cov3D_precomp, *opacities, *viewmatrix, *projmatrix;
  const int* radii;
  float scale_modifier;
  bool antialiasing;
  float *dL_dmeans2D, *dL_dcolors, *dL_dopacity, *dL_dmeans3D, *dL_dcov3D, *dL_dscales, *dL_drotations;
  float *dL_dT_sum, *dL_dvm_mean;
  bool raw;  // EOGS_FLAG_RAW_PARAMS
  const float* alt_affine;
  float* dL_dcolors_lead;  // second destination of the colour gradient's first lead_cols columns (or NULL)
  int lead_cols;
  bool alt_only;           // the records are those of an altitude-only render (REC_ALT)
  int noflag_ok;           // render_bwd_noflag_ok(): the records are flag-free (bit 0) where the scene allows / always (bit 1)
  int wide;                // records in flight per lane: gaussian_bwd_kernel's WIDE (0, 1, 2), gaussian_bwd_wide()
};
// Which gaussian_bwd_kernel variant a backward of (R, P) launches (token bit 60: the forward's lists are shallow in opacity).
// EOGS_GB_WIDE=0|1|2 forces one.
#define GB_WIDE_DEPTH 256.0f
int gaussian_bwd_wide(int64_t R, int P);
// per-Gaussian backward over rows [p_begin, p_end) (p_begin a multiple of BLK); the camera sums are finished by the call
// whose p_end == P
void launch_gaussian_bwd(const GaussBwdArgs& a, const GeomWS& g, const BinWS& b, int p_begin, int p_end, hipStream_t s);
void launch_selftest(uint32_t* out, hipStream_t s);

// ---- photometric loss (loss.hip, include/eogs_loss.h) ----
#define LOSS_WIN EOGS_LOSS_WINDOW
struct LossWindow {
  float w[LOSS_WIN];
};
struct LossWS {
  float* partial;    // [planes][tiles][2] per-workgroup {sum|x-y|, sum SSIM}
  float* plane_tmp;  // [planes][2]
  float* maps;       // [3][planes][H][W] dSSIM/d{mu1, E[x^2], E[xy]} (EOGS_LOSS_SSIM only)
  size_t map_stride;
  int tiles;
  size_t bytes;
};
LossWS loss_layout(char* base, int planes, int H, int W, unsigned mode);
void launch_loss_fwd(const LossWS& w, int planes, int H, int W, const float* img, const float* gt, unsigned mode,
                     float w_l1, float w_ssim, float bias, float* out, float* plane_sums, hipStream_t s);
void launch_loss_bwd(const LossWS& w, int planes, int H, int W, const float* img, const float* gt, unsigned mode,
                     float w_l1, float w_ssim, const float* upstream, const float* plane_grad, float* dimg,
                     hipStream_t s);

// ---- optimizer / compaction (optim.hip, include/eogs_optim.h) ----
#define EOGS_COMPACT_MAX_TENSORS 24  // tensors per compaction launch (more are split over launches)
struct CompactWS {
  uint32_t* blk;  // [nblk + 1] kept rows per 256-row workgroup -> exclusive prefix, total at [nblk]
  uint32_t nblk;
  size_t bytes;
};
CompactWS compact_layout(char* base, int64_t n_rows);
int launch_adam(int n, const eogs_adam_tensor* tensors, double beta1, double beta2, double eps, int64_t step, hipStream_t s);
int launch_sum_into(int n, const eogs_sum_tensor* tensors, int nsrc, hipStream_t s);
void launch_pack_columns(int64_t rows, int n, const eogs_pack_tensor* tensors, float* packed, int packed_cols, int unpack,
                         hipStream_t s);
void launch_compact_plan(const CompactWS& w, int64_t n_rows, const uint8_t* keep, hipStream_t s);
void launch_compact_apply(const CompactWS& w, int64_t n_rows, const uint8_t* keep, int n_tensors, const void* const* src,
                          void* const* dst, const int* row_bytes, hipStream_t s);

// ---- virtual-camera resample (resample.hip, include/eogs_resample.h) ----
void launch_resample_fwd(int C, int Hv, int Wv, int H, int W, int n_out, const float* vr, const float* uva,
                         const float* M, int fill_channel, float fill_value, float* sample, float* uv, hipStream_t s);
size_t resample_bwd_ws_bytes(int H, int W);
void launch_resample_bwd(int C, int Hv, int Wv, int H, int W, int n_out, const float* vr, const float* uva,
                         const float* M, int fill_channel, const float* gs, const float* guv, float* gvr, float* guva,
                         void* ws, hipStream_t s);

// ---- image chain after the raw render (shade.hip, include/eogs_shade.h) ----
size_t shade_ws_bytes();
void launch_shade_fwd(int H, int W, const float* raw, const float* alt_diff, const float* M, const float* inshadow, float* cc,
                      float* shaded, float* shadow, hipStream_t s);
void launch_shade_bwd(int H, int W, const float* raw, const float* alt_diff, const float* M, const float* inshadow,
                      const float* g_shaded, const float* g_cc, const float* g_shadow, float* g_raw, float* g_alt,
                      float* g_params, void* ws, hipStream_t s);
void launch_mloss_fwd(int H, int W, int mode, const float* alt_diff, const float* a, const float* b, const float* uv, float* out,
                      void* ws, hipStream_t s);
void launch_mloss_bwd(int H, int W, int mode, const float* alt_diff, const float* a, const float* b, const float* uv,
                      const float* out, const float* upstream, float* g_alt, float* g_a, float* g_b, hipStream_t s);
void launch_tshadow_fwd(int64_t n, const float* a, float* out, void* ws, hipStream_t s);
void launch_tshadow_bwd(int64_t n, const float* a, const float* upstream, float* g_a, hipStream_t s);

// ---- TSDF integration (tsdf.hip, include/eogs_tsdf.h) ----
void launch_tsdf_integrate(int nx, int ny, int nz, const float* ax, const float* ay, const float* az, const float* affine,
                           float scale, float trunc, int H, int W, const float* alt, const float* wgt, float* tsdf,
                           float* wvol, hipStream_t s);

// ---- 3-nearest-neighbour statistic (knn.hip, include/eogs_knn.h) ----
struct KnnWS {
  uint32_t *keyA, *keyB, *valA, *valB, *hist, *dtotal;  // Morton sort ping-pong + radix histograms
  float4* sorted;                                       // points in Morton order
  float* boxes;                                         // (nbox + 1) x {lo[3], hi[3]}; the last one is the scene box
  uint32_t nblk, nbox;
  size_t bytes;
};
KnnWS knn_layout(char* base, int P);
void launch_knn(const KnnWS& w, int P, const float* pts, float* out, hipStream_t s);
// `passes` stable 8-bit LSD passes over 32-bit keys with a 32-bit payload, ping-ponging A -> B -> A ... (binning.hip)
void launch_sort_u32(uint32_t* keyA, uint32_t* valA, uint32_t* keyB, uint32_t* valB, uint32_t n, int passes, uint32_t* hist,
                     uint32_t nblk, uint32_t* dtotal, hipStream_t s);
