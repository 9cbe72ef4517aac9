// render.hip — alpha blending per internal 8x8 tile, forward and backward.
//
// Work decomposition (both directions)
//   * ONE wave64 per internal 8x8 tile, lane = pixel. Waves are fully independent: no barriers, no cross-wave
//     merge; each wave owns a private slab of LDS. A 256-thread workgroup is just four tiles; workgroups are dealt
//     to the 8 XCDs so that each XCD (own L2) walks one contiguous band of tiles and neighbouring tiles share
//     their Gaussians in L2.
//   * the tile's list (built by binning.hip) holds exactly the Gaussians that can reach alpha >= 1/255 inside
//     the tile, in (depth, index) order. The wave takes it 64 entries at a time: lane i gathers entry i
//     ({xy, conic+opacity, 5 colours, 1/depth} = 48 B) from HBM/L2 (the gather of the NEXT chunk is in flight
//     while the current one is consumed), parks it in the wave's LDS slab, and the hot loop reads each entry back
//     at a wave-uniform address (LDS broadcast) one entry ahead of its use. Operands therefore arrive in VGPRs:
//     on gfx950 a VALU op with VGPR operands issues in ~2.4 cycles per wave64, with an SGPR operand in ~4.2, a
//     v_readlane with a variable lane in ~8 and v_pk_fma_f32 in ~10 (tools/ubench.hip), so LDS-broadcast beats both
//     the readlane broadcast and packed math here. (The reference re-fetches colours from global memory per
//     contributing pixel, DGR/cuda_rasterizer/forward.cu:386.)
//   * the conic is pre-scaled by log2(e) at gather time, so alpha = o * 2^p with p = (A dx - B dy) dx + C dy^2,
//     A = -a log2e / 2, B = b log2e, C = -c log2e / 2: three FMAs and one v_exp_f32 per pixel.
//
// Forward semantics: DGR/cuda_rasterizer/forward.cu:288-411.
// Backward semantics: DGR/cuda_rasterizer/backward.cu:457-643, restructured:
//   * traversal is BACK TO FRONT like the reference's (:536-643): last contributor first, T recovered from the forward's final
//     transmittance by T_j = T_{j+1} / (1 - alpha_j) (:573), the background term from T_final (:617-620). The colour behind a
//     Gaussian — the reference's five-channel accum_rec recursion (:586-599) — is carried as ONE number per pixel, its projection
//     on the pixel's upstream gradient g (a constant of the pixel):
//       a_j = g . accum_rec_j,   a_j = a_{j+1} + alpha_{j+1} (g.c_{j+1} - a_{j+1}),   a_last = 0
//       dL/dalpha_j = T_j (g.c_j - a_j) - T_final / (1 - alpha_j) (bg . g)
//     and the background term is the same recursion started at a_last = bg . g instead of 0 (the background is the colour behind
//     the last contributor: T_final / (1 - alpha_j) = T_j x the transmittance of everything behind j, which is exactly the weight
//     the recursion leaves of its initial value at j), so the kernels evaluate dL/dalpha_j = T_j (g.c_j - a_j) alone —
//     the same sum, at one dot product, one reciprocal and five multiply-adds per pair; with bg = 0 the same bits. (Rounds 1-5 walked
//     front to back and took the sum behind a Gaussian as "rendered total minus running prefix": the same instruction count, but
//     an absolute error of an ulp of the TOTAL in a quantity that, deep under opaque Gaussians, is orders of magnitude smaller;
//     forwards of image-sized Gaussians had to be switched to a second, slower kernel by a tuned threshold. DESIGN.md 5.)
//   * the 12 atomicAdd per contributing (pixel,Gaussian) of the reference (:598-640) are replaced by a
//     transposition through LDS: the pixel-parallel pass only produces two numbers per (pixel, Gaussian),
//     u = alpha T and v = G dL/dalpha; every 8 surviving Gaussians the wave switches to lanes = (Gaussian,
//     pixel row) and accumulates the six moments of v and the five colour sums of u serially in registers
//     (see transpose_round). No cross-lane reduction tree, no atomics, and ONE record (11 floats in 48 bytes) per
//     (tile,Gaussian) pair written with plain stores by the only wave that owns the pair (plus a 1-byte live flag:
//     pairs behind every pixel's last contributor are never gathered nor written); gaussian_bwd_kernel sums each
//     Gaussian's live records in fixed order (bitwise reproducible gradients).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#pragma clang fp contract(fast)

// Threads per render workgroup. A workgroup is nothing but RBLK / 64 independent tiles (no barriers, no shared data), and its
// LDS and wave slots are only released when its LAST wave ends, while tiles have uneven lists: one tile = one wave = one
// workgroup measured render_bwd 0.355 -> 0.329 ms, render_fwd 0.151 -> 0.144 ms, the block-list kernels -10 % / -3 %
// against four tiles per workgroup. Every LDS address is a compile-time constant as well.
#ifndef EOGS_RENDER_BLK
#define EOGS_RENDER_BLK 64
#endif
#define RBLK EOGS_RENDER_BLK
static_assert(SUBX == 8 && SUBY == 8 && PPL == 1, "render kernels are written for 8x8 internal tiles, one pixel per lane");

namespace {

#define LN2 0.6931471805599453f
#define ENT 12   // floats per staged list entry: gx gy A B | C op f0 f1 | f2 f3 f4 1/depth
#define FWD_CAP 96   // slab entries, forward: up to FWD_MIN-1 waiting + 64 appended
#define FWD_MIN 32   // forward processes the slab once it holds this many entries (or the list is exhausted)
#define KSURV 8  // survivors per transposition round (backward)
// u/v matrices [survivor k][pixel p] in LDS, laid out as two half-matrices (pixels 0..31 / 32..63) with row stride UV_ROW = 32
// and 268 floats between the halves. The pixel-parallel writes (lane = p, fixed k) hit 32 consecutive banks per half-wave.
// The transposed reads (lane = 8k + o reads pixels 8o .. 8o + 7 of row k) are two ds_read_b128 per matrix: a b128 read is
// served in four groups of 16 lanes over 64 banks (MI355X_MICROARCH.md, LDS table), and with this stride and half distance
// the 16 lanes of every group fall into 16 different 16-byte slots — 4 LDS cycles per instruction, 16 per round for u and v
// where rounds 1-4 (row stride 33, four ds_read2_b32 per matrix) took 32 (the render backward keeps the LDS array busy for
// 80 % of its cycles: every cycle less there counts).
#define UV_ROW 32
#define UV_HALF 268
#define UV_SIZE (2 * UV_HALF)
#define UV_PITCH 576  // >= UV_SIZE, multiple of 64 dwords
__device__ inline int uv_index(int k, int p) { return (p >> 5) * UV_HALF + k * UV_ROW + (p & 31); }
__device__ inline void uv_row8(const float* row, float (&x)[8]) {  // eight consecutive pixels of a row (16-byte aligned)
  const float4 a = *reinterpret_cast<const float4*>(row), b = *reinterpret_cast<const float4*>(row + 4);
  x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
}

// ---- wave64 helpers ----
__device__ inline uint32_t wave_max_u32(uint32_t v) { return wave_max_u32_dpp(v); }  // (common.h: DPP path, no LDS round trips)
// wave-uniform sum on the DPP path: four rotations inside each row of 16 lanes, the row totals handed on by row_bcast15 /
// row_bcast31, lane 63 read back — seven VALU instructions and no LDS round trip (__shfl_xor is a ds_bpermute_b32: six
// dependent ones cost the quad forward 4 % when this sum sat in its chunk loop, profiles/r05_ab_plain_trips.txt)
__device__ inline float wave_sum_f32(float v) {
  auto dpp = [](float x, auto ctrl, auto rows) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, decltype(rows)::value, 0xF, true));
  };
  v += dpp(v, std::integral_constant<int, 0x121>{}, std::integral_constant<int, 0xF>{});  // row_ror:1
  v += dpp(v, std::integral_constant<int, 0x122>{}, std::integral_constant<int, 0xF>{});  // row_ror:2
  v += dpp(v, std::integral_constant<int, 0x124>{}, std::integral_constant<int, 0xF>{});  // row_ror:4
  v += dpp(v, std::integral_constant<int, 0x128>{}, std::integral_constant<int, 0xF>{});  // row_ror:8: every lane holds its row's sum
  v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});  // row_bcast15 into rows 1 and 3
  v += dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});  // row_bcast31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// The quad forward's plain chunks (render_fwd_quad_kernel): while the bound on T — the product of (1 - opacity) over everything
// a tile has listed so far — stays above 2^-13 = 1.22e-4 no pixel can reach the reference's stop test T (1 - alpha) < 1e-4
// (forward.cu:378-382): 22 % of margin against the rounding of a product of a few thousand fp32 factors (relative error
// < 1e-3) and of the bound's own fp32 sum.
#define PLAIN_LOG2_T (-13.0f)
// LDS produced and consumed by the SAME wave: the hardware executes a wave's LDS instructions in order, so only
// the compiler must be kept from reordering them.
__device__ inline void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// XCD-aware tile of this wave: workgroup b runs on XCD b % 8 (round-robin dispatch; speed only, never correctness).
//   * with a tile schedule (binning.hip tile_sched_body): XCD x walks ITS sequence of 32 x 32-px blocks, which
//     the schedule cut so that the eight sequences hold equal WORK (listed pairs), long blocks first and the near-empty ones
//     last; the 16 tiles of a block are 16 consecutive workgroups of its XCD. The workgroup's descriptor — written by
//     block_lists_kernel at the block's place in the schedule — holds the tile AND its list range: one 16-byte load where
//     rounds 1-3 computed the tile and loaded ranges[tile]. Workgroups past an XCD's count, or on tiles outside the image, leave.
//   * without one (images beyond 4096 blocks, or a forward that lists nothing): XCD x gets the contiguous run of tile groups
//     [x*per, (x+1)*per); gridDim.x is a multiple of 8; `range` is left for the caller to load.
__device__ inline int tile_of_wave(const uint4* __restrict__ desc, const uint32_t* __restrict__ sched, int lg16, int gsx, uint2& range) {
  if (RBLK == 64 && desc != nullptr) {
    const uint32_t x = blockIdx.x & 7u, i = blockIdx.x >> 3;
    const uint32_t cnt = sched[x];                          // (independent loads: one round trip)
    const uint4 d = desc[(size_t)x * (uint32_t)lg16 + i];
    if ((i >> 4) >= cnt || d.x == 0xFFFFFFFFu) return 0x7FFFFFFF;
    range = make_uint2(d.y, d.z);
    return (int)(d.x >> 16) * gsx + (int)(d.x & 0xFFFFu);
  }
  const int per = gridDim.x >> 3;
  const int grp = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  // the wave index is uniform across the wave: tell the compiler, so tile, list range and loop control live in SGPRs
  return grp * (RBLK / 64) + (RBLK == 64 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)));
}

// -DEOGS_WAVE_TRACE: per tile {start, end} of its wave on the constant-rate clock, the shader-clock ticks between them, and where
// it ran (HW_ID, XCC_ID): tools/wave_trace.py turns them into resident waves over time, per-XCD work and finish times, slot
// idle gaps, ramp and tail. Diagnostics only (profiles/r04_wave_trace.txt).
#ifdef EOGS_WAVE_TRACE
#define WTRACE_TILES 65536
__device__ unsigned long long g_wave_trace[2][WTRACE_TILES][4];
#define WTRACE_BEGIN const unsigned long long wt_r0 = __builtin_amdgcn_s_memrealtime(), wt_c0 = __builtin_amdgcn_s_memtime()
#define WTRACE_END(dir, tile)                                                                                                \
  do {                                                                                                                         \
    if ((threadIdx.x & 63) == 0 && (tile) < WTRACE_TILES) {                                                                    \
      g_wave_trace[dir][tile][0] = wt_r0;                                                                                      \
      g_wave_trace[dir][tile][1] = __builtin_amdgcn_s_memrealtime();                                                           \
      g_wave_trace[dir][tile][2] = __builtin_amdgcn_s_memtime() - wt_c0;                                                       \
      g_wave_trace[dir][tile][3] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32); \
    }                                                                                                                          \
  } while (0)
#else
#define WTRACE_BEGIN
#define WTRACE_END(dir, tile)
#endif

// One list entry as gathered by a lane: the Gaussian's 64-byte render record written by preprocess_fwd_kernel
// (conic pre-scaled by log2 e: alpha = o 2^p, p = (A dx - B dy) dx + C dy^2).
struct Cand {
  float4 q0;  // gx gy A B
  float4 q1;  // C op f0 f1
  float4 q2;  // f2 f3 f4 1/depth
  uint32_t slot;
  uint32_t qm;  // (quad backward only) the forward's quad mask of the entry, see peek_cand_q
  bool hit;  // the entry lists this wave's internal tile
};

// List entries are fetched in two pipelined stages so that no dependent global load is ever waited for inside a chunk:
//   peek   (two chunks ahead): lane i reads entry k's sort key (block id | sub-mask << 16) and payload
//          {Gaussian id, first record slot} — coalesced, unconditional;
//   gather (one chunk ahead) : if the sub-mask lists this wave's internal tile (bit `sub`), the Gaussian's 64-byte render
//          record is gathered; the record slot of (entry, internal tile) is the entry's first slot plus the number of
//          listed tiles before this one.
struct Peek {
  uint32_t key;
  uint32_t qraw;  // peek_cand_q: the byte loaded from BinWS::qmask, untouched (see there)
  uint2 e;
};
template <int MACRO>
__device__ inline Peek peek_cand(uint32_t k, uint32_t end, const uint32_t* __restrict__ keys,
                                 const uint2* __restrict__ point_list) {
  Peek p;
  p.key = 0u;  // empty sub-mask: never a hit
  p.qraw = 0u;
  p.e = make_uint2(0u, 0u);
  if (k < end) {
    p.key = MACRO > 1 ? keys[k] : 1u;  // block size 1: every entry of the tile's own list is a hit
    // (two 4-byte loads, not one 8-byte load: the gather extends e.x to a 64-bit offset in the register PAIR the 8-byte load
    // wrote, so the compiler moved e.y out of that pair right behind the load — a use, hence an s_waitcnt there, which with the
    // in-order counter also waits for the next chunk's three gather loads issued just before: a memory round trip per chunk in
    // the block-list kernels, found in the ISA)
    const uint32_t* pl = reinterpret_cast<const uint32_t*>(point_list + k);
    p.e.x = pl[0];
    asm volatile("" ::: "memory");  // (keeps the two loads apart: merged again they are the 8-byte load)
    p.e.y = pl[1];
  }
  return p;
}
// The quad backward's peek on per-tile lists: the entry's in-range flag (bit 4) and, where the quad FORWARD walked the chunk, the 4-bit
// quad mask it left in `qmask` (BinWS::qmask) — the backward's chunk set-up then needs no quad_mask() of its own (four block_hit
// tests, a v_log and two v_rcp per entry: half of the set-up's instructions).
__device__ inline Peek peek_cand_q(uint32_t k, uint32_t begin, uint32_t end, const uint8_t* __restrict__ qmask, const uint2* __restrict__ point_list) {
  // (The loaded byte is NOT combined with the flag here: `0x10 | qmask[k]` is a use, and the compiler waits for the load on the
  // spot — s_waitcnt vmcnt(0), which also waits for the three record loads of the next chunk's gather issued just before it: one
  // full memory round trip per chunk in the middle of the chunk set-up, found in the ISA (render_bwd -2.9 % without it). The
  // byte is first looked at a chunk later, in gather_cand. Taking "in range" as an argument of the gather instead of carrying
  // the flag — no scratch, 8 bytes with it — measured 1 % slower at the headline and 1 % faster at trained opacities.)
  Peek p;
  p.key = 0u;
  p.qraw = 0u;
  p.e = make_uint2(0u, 0u);
  if (k >= begin && k < end) {  // (the backward walks from the list's end: an index may also fall in front of the list)
    p.key = 0x10u;
    if (qmask) p.qraw = (uint32_t)qmask[k];
    p.e = point_list[k];
  }
  return p;
}
template <int MACRO>
__device__ inline Cand gather_cand(const Peek& p, uint32_t sub, const float4* __restrict__ packed) {
  Cand c;
  c.q0 = c.q1 = c.q2 = make_float4(0.f, 0.f, 0.f, 0.f);
  c.slot = 0;
  c.qm = MACRO > 1 ? 0u : ((p.key | p.qraw) & 0xFu);
  const uint32_t mask = MACRO > 1 ? p.key >> MACRO_KEY_BITS : p.key;  // block size 1: peek_cand's in-range flag
  c.hit = MACRO > 1 ? ((mask >> sub) & 1u) != 0u : mask != 0u;
  if (c.hit) {
    c.slot = MACRO > 1 ? p.e.y + (uint32_t)__popc(mask & ((1u << sub) - 1u)) : p.e.y;
    const float4* r = packed + 4 * (size_t)p.e.x;  // one 64-byte line per list entry
    c.q0 = r[0]; c.q1 = r[1]; c.q2 = r[2];
  }
  return c;
}

// The hit lanes park their entries, compacted in list order, in the wave's LDS slab (rank = number of hit lanes below);
// afterwards any lane can read any entry at a wave-uniform address (LDS broadcast, no bank conflicts) and gets the
// values in VGPRs: VALU ops on VGPR operands issue in ~2.4 cycles, the same ops on SGPR operands (v_readlane
// broadcast) in ~4.2, and a v_readlane with a variable lane costs ~8 (measured, tools/ubench.hip).
// Entries are appended behind the `fill` entries already waiting in the slab. Returns the number appended (wave-uniform).
__device__ inline int park(float* slab, uint32_t* sslot, int lane, const Cand& c, int fill) {
  const unsigned long long hm = __builtin_amdgcn_ballot_w64(c.hit);
  if (c.hit) {
    const int rank = fill + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
    float4* d = reinterpret_cast<float4*>(slab + rank * ENT);
    d[0] = c.q0; d[1] = c.q1; d[2] = c.q2;
    if (sslot) sslot[rank] = c.slot;
  }
  return (int)__popcll(hm);
}
struct Ent {
  float4 q0, q1, q2;
};
// p = (A dx - B dy) dx + C dy^2 with ONE fixed association and explicit fused steps, so that every kernel (forward and
// backward, tile and quad variants, both unrolled copies of the entry loops) rounds it identically: left to
// `fp contract(fast)`, the compiler picked different contractions in different copies, and a pixel within an ulp of the
// alpha = 1/255 or p = 0 threshold could blend an entry in the forward and skip it in the backward.
__device__ inline float power_of(const Ent& e, float dx, float dy) {
  const float t = __builtin_fmaf(e.q0.z, dx, -(e.q0.w * dy));
  return __builtin_fmaf(dy, e.q1.x * dy, dx * t);
}
__device__ inline Ent fetch(const float* slab, int j) {
  const float4* s = reinterpret_cast<const float4*>(slab + j * ENT);
  Ent e;
  e.q0 = s[0]; e.q1 = s[1]; e.q2 = s[2];
  return e;
}

}  // namespace

template <int MACRO>
__global__ __launch_bounds__(RBLK) void render_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ keys, const uint2* __restrict__ point_list, int W, int H, int gsx, int ntiles, int gmx, const uint4* __restrict__ desc, const uint32_t* __restrict__ sched, int lg16,
    const float4* __restrict__ packed, const float* __restrict__ bg,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
    float* __restrict__ out_invdepth, int opts) {
  __shared__ __attribute__((aligned(16))) float s_slab[RBLK / 64][FWD_CAP * ENT];
  __shared__ uint32_t s_pos[RBLK / 64][FWD_CAP];  // list position of every slab entry (what n_contrib counts)
  const int lane = threadIdx.x & 63;
  uint2 range = make_uint2(0u, 0u);
  const int tile = tile_of_wave(desc, sched, lg16, gsx, range);
  if (tile >= ntiles) return;  // wave-uniform; waves never synchronise with each other
  float* slab = s_slab[RBLK == 64 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
  uint32_t* spos = s_pos[RBLK == 64 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
  const int px = (tile % gsx) * SUBX + (lane & 7), py = (tile / gsx) * SUBY + (lane >> 3);
  const bool inside = px < W && py < H;
  const uint32_t pix_id = (uint32_t)py * (uint32_t)W + (uint32_t)px;
  const float pxf = (float)px, pyf = (float)py;
  const int ftx = tile % gsx, fty = tile / gsx;
  if (desc == nullptr) range = ranges[(fty / MACRO) * gmx + ftx / MACRO];  // the macro block's list
  const uint32_t sub = (uint32_t)((fty % MACRO) * MACRO + ftx % MACRO);

  float T = 1.0f;
  uint32_t last_contributor = 0;
  float C[NCH] = {0.f, 0.f, 0.f, 0.f, 0.f};
  float invd = 0.f;
  bool done = !inside;

  // What this kernel leaves in n_contrib: 1 + the position, IN THE LIST THE WAVE WALKS (the tile's own list, or its block's list
  // with the other tiles' entries in between), of the pixel's last blended entry; 0 = none. The backward of the same list
  // granularity reads it as "entries at positions >= n_contrib take no part" and starts its back-to-front walk there.
  int fill = 0;        // entries waiting in the slab
  Cand nxt = gather_cand<MACRO>(peek_cand<MACRO>(range.x + lane, range.y, keys, point_list), sub, packed);
  Peek pk = peek_cand<MACRO>(range.x + 64 + lane, range.y, keys, point_list);
  for (uint32_t c0 = range.x; c0 < range.y; c0 += 64) {
    wave_lds_sync();  // previous chunk's reads are done
    nxt.slot = c0 - range.x + (uint32_t)lane;  // (the forward has no use for the record slot: the entry's list position travels in its place)
    fill += park(slab, spos, lane, nxt, fill);
    nxt = gather_cand<MACRO>(pk, sub, packed);                                  // chunk c0+64: in flight during this chunk
    pk = peek_cand<MACRO>(c0 + 128 + lane, range.y, keys, point_list);  // chunk c0+128
    wave_lds_sync();
    if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;  // every pixel of the tile has terminated
    // a block's list holds the entries of all its internal tiles: keep appending until the inner loop is worth entering
    if (fill < FWD_MIN && c0 + 64 < range.y) continue;
    const int n = fill;
    fill = 0;
    // one list entry against this lane's pixel; returns nothing, all state is captured by reference
    int last_pos = -1;  // slab position of this pixel's last blended entry in this batch
    auto blend = [&](const Ent& e, int j) {
      const float dx = e.q0.x - pxf, dy = e.q0.y - pyf;
      const float p = power_of(e, dx, dy);
      const float alpha = fminf(e.q1.y * __builtin_amdgcn_exp2f(p), 0.99f);
      bool valid = !done && !(p > 0.0f) && !(alpha < 1.0f / 255.0f);
      const float test_T = T * (1.f - alpha);
      // test_T >= 0 and finite: its bits order like the value, and ONE compare serves `stop` and `!stop`
      const bool stop = valid && __float_as_uint(test_T) < __float_as_uint(0.0001f);
      done = done || stop;    // this Gaussian is NOT blended; the pixel is finished
      valid = valid != stop;  // (stop implies valid)
      const float wgt = valid ? alpha * T : 0.f;
      // wave-uniform skip of entries no pixel blends; wgt > 0 <=> valid (alpha >= 1/255, T >= 1e-4), and a ballot of a
      // COMPARE is the compare's own lane mask (a ballot of a boolean costs a v_cndmask + v_cmp)
      if (__builtin_amdgcn_ballot_w64(wgt > 0.f) == 0ull) return;
      C[0] += e.q1.z * wgt; C[1] += e.q1.w * wgt;
      C[2] += e.q2.x * wgt; C[3] += e.q2.y * wgt; C[4] += e.q2.z * wgt;
      invd += e.q2.w * wgt;
      T = valid ? test_T : T;
      last_pos = valid ? j : last_pos;
    };
    // software pipeline over two register sets: each entry's broadcast read is issued one entry ahead of its use
    // and lands directly in the set that has just been consumed (no register rotation)
    Ent ea = fetch(slab, 0);
    int j = 0;
    for (; j + 1 < n; j += 2) {
      const Ent eb = fetch(slab, j + 1);
      blend(ea, j);
      ea = fetch(slab, j + 2 < n ? j + 2 : j + 1);
      blend(eb, j + 1);
    }
    if (j < n) blend(ea, j);
    if (last_pos >= 0) last_contributor = spos[last_pos] + 1u;
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


// ------------------------------------------------------------------------------------------------------
// Quad sub-lists
// ------------------------------------------------------------------------------------------------------
// At 8x8 granularity only ~20-40 % of a list entry's 64 pixel evaluations pass the alpha >= 1/255 test (a footprint of
// 1.4-3 sigma is smaller than the tile it touches). The wave therefore splits its tile into four 4x4 QUADS, one per
// DPP row of 16 lanes (lane = 16 q + r, pixel (4 (q&1) + (r&3), 4 (q>>1) + (r>>2)) of the tile), and each gathering
// lane tests its entry against the four quads with the same exact predicate binning uses for tiles (block_hit, common.h:
// a superset of the pixels that can blend the entry, so results are unchanged). The entries stay ONCE in the wave's LDS
// slab; per quad, a byte list of slab positions is built by ballot + prefix count. In the hot loop every quad walks its
// OWN sub-list: one ds_read_u8 (4 distinct addresses per wave) and the entry's broadcast reads at 4 distinct addresses.
// Trip count = the longest sub-list (~0.6 of the tile's list at 1M Gaussians / 1024^2), list traffic from HBM unchanged.
namespace {

#define FQ_CAP 64  // slab entries of the quad forward: per-tile lists only, a chunk is processed as soon as it is parked
#define QCAP 68   // dwords per quad sub-list (>= FQ_CAP + 3: padding and the pipelined over-read); with the dummy slab entry
                  // the forward wave needs 4208 B of LDS: 32 waves per CU
// Waves per SIMD the quad forward is compiled for: at 8 (64 VGPRs) the two entry register sets spill (48 B of scratch per lane,
// written and re-read every chunk) and the kernel takes 0.142 ms; at 6 (76 VGPRs, no scratch) 0.132 ms; 5 the same, 4 slower.
#ifndef EOGS_FW
#define EOGS_FW 6
#endif

__device__ inline void quad_pixel(int lane, int& ox, int& oy) {
  const int q = lane >> 4, r = lane & 15;
  ox = 4 * (q & 1) + (r & 3);
  oy = 4 * (q >> 1) + (r >> 2);
}

// 4-bit mask of the quads of the tile at (bx0, by0) this entry can reach with alpha >= 1/255. In the log2 domain of the
// render record: alpha = o 2^p, p = -(a dx^2 + 2 b dx dy + c dy^2) with a = -A, b = B/2, c = -C, so
// alpha >= 1/255 <=> a dx^2 + 2 b dx dy + c dy^2 <= log2(255 o). Anything not provably a positive-definite conic with a
// finite threshold keeps all four quads.
__device__ inline uint32_t quad_mask(const Cand& c, float bx0, float by0) {
  const float a = -c.q0.z, b = 0.5f * c.q0.w, cc = -c.q1.x;
  const float thr = __builtin_amdgcn_logf(255.f * c.q1.y);
  const float thr_m = thr + 1e-3f * (1.f + fabsf(thr));
  const bool pd = a > 0.f && cc > 0.f && a * cc - b * b > 0.f && thr_m < 3.0e38f && thr_m > -3.0e38f;
  if (!pd) return 0xFu;
  const float b_c = b * __builtin_amdgcn_rcpf(cc), b_a = b * __builtin_amdgcn_rcpf(a);
  uint32_t m = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const float x0 = bx0 + 4.f * (float)(q & 1), y0 = by0 + 4.f * (float)(q >> 1);
    if (block_hit(c.q0.x, c.q0.y, a, b, cc, b_c, b_a, thr_m, x0, y0, x0 + 3.f, y0 + 3.f)) m |= 1u << q;
  }
  return m;
}

// Appends this chunk's hit entries to the four sub-lists, as the BYTE OFFSETS of their slab entries (the hot loop's ds_read
// takes them as they come). `pos` = the lane's slab position (valid where c.hit), qm = its quad mask. nq[] are wave-uniform.
__device__ inline void quad_append(uint32_t* sidx, int lane, bool hit, int pos, uint32_t qm, int (&nq)[4]) {
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const bool in = hit && ((qm >> q) & 1u);
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(in);
    if (in) {
      const int rank = nq[q] + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
      sidx[q * QCAP + rank] = (uint32_t)(pos * (4 * ENT));
    }
    nq[q] += (int)__popcll(bal);
  }
}

}  // namespace

// ALT: an altitude-only render (EOGS_FLAG_ALT_ONLY): only feature channel 3 is blended and stored — out_color is ONE plane
// f32[H, W], out_invdepth is not written — and a trip reads 28 of the entry's 48 bytes.
// INVD = false: the caller passed no inverse-depth image (out_invdepth == NULL: the reference's render() drops that output,
// gaussian_renderer/renderer.py:101,126) — its multiply-add per evaluated (pixel, entry) is left out, one of 22.
template <int MACRO, bool ALT, bool INVD = true>
__global__ __launch_bounds__(RBLK) __attribute__((amdgpu_waves_per_eu(EOGS_FW, EOGS_FW))) void render_fwd_quad_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ keys, const uint2* __restrict__ point_list, int W, int H, int gsx, int ntiles, int gmx, const uint4* __restrict__ desc, const uint32_t* __restrict__ sched, int lg16,
    const float4* __restrict__ packed, const float* __restrict__ bg,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
    float* __restrict__ out_invdepth, int opts) {
  static_assert(MACRO == 1, "quad sub-lists run on per-tile lists: every chunk of 64 list entries is processed at once");
  // slab position FQ_CAP holds a DUMMY entry (opacity 0: alpha = 0 fails the 1/255 test at every pixel); sub-lists
  // shorter than the longest one are padded with it, so the hot loop needs no "is my quad still active" test
  __shared__ __attribute__((aligned(16))) float s_slab[RBLK / 64][(FQ_CAP + 1) * ENT];
  __shared__ __attribute__((aligned(16))) uint32_t s_idx[RBLK / 64][4 * QCAP];
  const int lane = threadIdx.x & 63;
  uint2 range = make_uint2(0u, 0u);
  const int tile = tile_of_wave(desc, sched, lg16, gsx, range);
  if (tile >= ntiles) return;  // wave-uniform; waves never synchronise with each other
  WTRACE_BEGIN;
  const int w = RBLK == 64 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* slab = s_slab[w];
  uint32_t* sidx = s_idx[w];
  if (lane < ENT) slab[FQ_CAP * ENT + lane] = 0.f;
  int ox, oy;
  quad_pixel(lane, ox, oy);
  const int tx0 = (tile % gsx) * SUBX, ty0 = (tile / gsx) * SUBY;
  const int px = tx0 + ox, py = ty0 + oy;
  const bool inside = px < W && py < H;
  const uint32_t pix_id = (uint32_t)py * (uint32_t)W + (uint32_t)px;
  const float pxf = (float)px, pyf = (float)py;
  const float bx0 = (float)tx0, by0 = (float)ty0;
  if (desc == nullptr) range = ranges[tile];
  const int myq = lane >> 4;
  const uint32_t* myidx = sidx + myq * QCAP;
  // stale elements are read (never used) by the pipelined loop: make them valid slab offsets
  for (int t = lane; t < 4 * QCAP; t += 64) sidx[t] = 0u;
  static_assert(QCAP >= FQ_CAP + 3, "sub-list sizing");

  float T = 1.0f;
  // What this kernel leaves in n_contrib: the number of list entries BEFORE the entry at which the pixel stopped (that entry
  // is not blended: forward.cu:378-382), or 0xFFFFFFFF when it never stopped. The backward kernels read it as "entries at
  // positions >= n_contrib take no part"; everything between a pixel's last blended entry (the reference's n_contrib - 1 and
  // what the one-list-per-tile forward kernels store) and its stop entry fails the alpha test on its own, so any value in that
  // range gives the same gradients bit for bit — and this one needs no select per trip in the plain chunks, and tells the
  // backward's plain chunks that a pixel which never stopped contributes down to the end of the list.
  uint32_t stop_at = 0xFFFFFFFFu;
  float C[NCH] = {0.f, 0.f, 0.f, 0.f, 0.f};
  float invd = 0.f;
  bool done = !inside;

  uint32_t jbase = 0;  // entries of THIS tile processed so far: list positions are counted over the tile's own entries
  Cand nxt = gather_cand<1>(peek_cand<1>(range.x + lane, range.y, keys, point_list), 0u, packed);
  Peek pk = peek_cand<1>(range.x + 64 + lane, range.y, keys, point_list);
  uint32_t c0 = range.x;
  float lsum = 0.f;  // log2 of the bound on every pixel's T behind the chunks taken so far (wave-uniform)
  // The chunk loop exists twice, one after the other. PLAIN: the tile's first chunks, as long as no pixel of the tile can reach
  // the stop test inside the chunk (`lsum`) and no entry of it has o > 0.99 (then o 2^p <= o: the clamp never binds). The stop
  // test, the clamp and the `done` mask are no-ops there and are left out: three instructions of the 4-cycle class (v_min,
  // v_cmp, v_cndmask) and three mask operations of a trip's 22; what is computed is computed by the same expressions as in the
  // general form, bit for bit. The first chunk that does not qualify, and every chunk after it, takes the general loop.
  // (Two loops in sequence, not two trip loops inside one chunk loop: that form kept both variants' registers alive together
  // and spilled 72 bytes per lane; profiles/r05_ab_plain_trips_first.txt.)
  auto chunks = [&](auto plainc) {
    constexpr bool PLAIN = decltype(plainc)::value;
    for (; c0 < range.y; c0 += 64) {
      if constexpr (PLAIN) {
        // alpha <= o for every pixel: T >= prod (1 - min(o, 0.99)) over everything listed up to and including this chunk
        const float l = lsum + wave_sum_f32(nxt.hit ? __builtin_amdgcn_logf(1.f - fminf(nxt.q1.y, 0.99f)) : 0.f);
        if (!(l > PLAIN_LOG2_T) || __builtin_amdgcn_ballot_w64(nxt.hit && nxt.q1.y > 0.99f) != 0ull) return;
        lsum = l;
      }
      wave_lds_sync();  // previous chunk's reads are done
      int nq[4] = {0, 0, 0, 0};
      {
        // per-tile lists: the in-range entries are lanes 0..fill-1, parked at their own lane index
        const uint32_t qm = nxt.hit ? quad_mask(nxt, bx0, by0) : 0u;
        // (`keys` is unused on per-tile lists: this launch receives BinWS::qmask through it — one byte per list entry, written
        // coalesced, for the quad backward's chunk set-up: peek_cand_q)
        if (nxt.hit && keys != nullptr) const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(keys))[c0 + (uint32_t)lane] = (uint8_t)qm;
        quad_append(sidx, lane, nxt.hit, lane, qm, nq);
      }
      const int fill = park(slab, nullptr, lane, nxt, 0);
      nxt = gather_cand<1>(pk, 0u, packed);                           // chunk c0+64: in flight during this chunk
      pk = peek_cand<1>(c0 + 128 + lane, range.y, keys, point_list);  // chunk c0+128
      wave_lds_sync();
      if (!PLAIN && __builtin_amdgcn_ballot_w64(!done) == 0ull) {  // every pixel of the tile has terminated
        c0 = range.y;
        return;
      }
      const int nmax = max(max(nq[0], nq[1]), max(nq[2], nq[3]));
      // pad the shorter sub-lists up to nmax (+2 for the pipelined over-read) with the dummy entry
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (lane <= nmax + 1 - nq[q]) sidx[q * QCAP + nq[q] + lane] = (uint32_t)(FQ_CAP * 4 * ENT);
      wave_lds_sync();
      // one entry of this lane's quad sub-list against the lane's pixel
      // Every trip evaluates four different entries, each picked because it reaches its quad: a trip in which no pixel of
      // the wave blends is rare, so there is no wave-uniform early-out here (the ballot it needs costs two VALU
      // instructions per trip). test_T >= 0 and finite, so its bits order like the value: ONE integer compare serves both
      // `term` and `!term` (the float compare is emitted twice, once per polarity, for NaN's sake).
      uint32_t stop_off = 0xFFFFFFFFu;  // slab byte offset of the entry at which this pixel stopped, if in this chunk
      auto fetch_off = [&](uint32_t off) {
        const float4* e4 = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(slab) + off);
        Ent e;
        e.q0 = e4[0];
        if (ALT) {  // C, opacity and the altitude feature: 12 of the other 32 bytes
          const float2 co = *reinterpret_cast<const float2*>(e4 + 1);
          e.q1 = make_float4(co.x, co.y, 0.f, 0.f);
          e.q2 = make_float4(0.f, reinterpret_cast<const float*>(e4 + 2)[1], 0.f, 0.f);
        } else {
          e.q1 = e4[1]; e.q2 = e4[2];
        }
        return e;
      };
      auto blend = [&](const Ent& e, uint32_t off) {
        const float dx = e.q0.x - pxf, dy = e.q0.y - pyf;
        const float p = power_of(e, dx, dy);
        float wgt;
        bool valid;
        if constexpr (PLAIN) {
          const float alpha = e.q1.y * __builtin_amdgcn_exp2f(p);
          valid = !(p > 0.0f) && !(alpha < 1.0f / 255.0f);
          const float a_eff = valid ? alpha : 0.f;
          wgt = a_eff * T;
          T = T * (1.f - a_eff);
        } else {
          const float alpha = fminf(e.q1.y * __builtin_amdgcn_exp2f(p), 0.99f);
          valid = !done && !(p > 0.0f) && !(alpha < 1.0f / 255.0f);
          const float test_T = T * (1.f - alpha);
          const bool stop = valid && __float_as_uint(test_T) < __float_as_uint(0.0001f);
          done = done || stop;    // this Gaussian is NOT blended; the pixel is finished
          valid = valid != stop;  // (stop implies valid: one mask xor instead of a second compare)
          wgt = valid ? alpha * T : 0.f;
          T = valid ? test_T : T;
          stop_off = stop ? off : stop_off;
        }
        if (ALT) {
          C[3] += e.q2.y * wgt;
        } else {
          C[0] += e.q1.z * wgt; C[1] += e.q1.w * wgt;
          C[2] += e.q2.x * wgt; C[3] += e.q2.y * wgt; C[4] += e.q2.z * wgt;
          if (INVD) invd += e.q2.w * wgt;
          else asm volatile("" :: "v"(e.q2.w));  // (keeps the entry's third read a ds_read_b128: see render_bwd_quad_kernel's grad)
        }
      };
      // software pipeline: sub-list elements two trips ahead, entry reads one trip ahead, two register sets; groups of eight
      // trips are unrolled so that the sub-list reads sit at immediate offsets (four elements per ds_read2_b64)
      uint32_t o0 = myidx[0], o1 = myidx[1];
      Ent ea = fetch_off(o0);
      const int ngrp = nmax >> 3;
      for (int gi = 0; gi < ngrp; gi++) {
        const uint32_t* ip = myidx + 8 * gi;
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
          const Ent eb = fetch_off(o1);
          const uint32_t o2 = ip[t + 2];
          blend(ea, o0);
          ea = fetch_off(o2);
          const uint32_t o3 = ip[t + 3];
          blend(eb, o1);
          o0 = o2;
          o1 = o3;
        }
      }
      {
        const int rem = nmax & 7;
        const uint32_t* ip = myidx + 8 * ngrp;
        int t = 0;
        for (; t + 1 < rem; t += 2) {
          const Ent eb = fetch_off(o1);
          const uint32_t o2 = ip[t + 2];
          blend(ea, o0);
          ea = fetch_off(o2);
          const uint32_t o3 = ip[t + 3];
          blend(eb, o1);
          o0 = o2;
          o1 = o3;
        }
        if (t < rem) blend(ea, o0);
      }
      if (!PLAIN && stop_off != 0xFFFFFFFFu) stop_at = jbase + stop_off / (uint32_t)(4 * ENT);
      jbase += (uint32_t)fill;
    }
  };
  // (a tile that hangs over the image's edge keeps the general loop: its outside pixels are `done` from the start)
  if ((opts & 1) != 0 && __builtin_amdgcn_ballot_w64(!inside) == 0ull) chunks(std::true_type{});
  chunks(std::false_type{});
  if (inside) {
    const size_t HW = (size_t)H * W;
    final_T[pix_id] = T;
    n_contrib[pix_id] = stop_at;
    if (ALT) {
      out_color[pix_id] = C[3] + T * bg[3];
    } else {
#pragma unroll
      for (int ch = 0; ch < NCH; ch++) out_color[ch * HW + pix_id] = C[ch] + T * bg[ch];
      if (INVD && out_invdepth) out_invdepth[pix_id] = invd;
    }
  }
  WTRACE_END(0, tile);
}

// Quad sub-lists: the trip count falls to the longest sub-list (0.6 of the tile's list at four listed tiles per Gaussian)
// against ~15 % more work per trip and ~170 instructions per chunk for the masks. They run whenever lists are per tile:
// measured ahead of the one-list-per-tile kernels at every footprint tried (fwd -20 % / bwd -18 % at 4.0 listed tiles per
// Gaussian, -9 % / -4 % at 10.8, -10 % / -6 % at 12 - 14, level at 31). EOGS_QUAD_SWITCH=<listed tiles per Gaussian>
// overrides the crossover (0 disables the quad kernel).
// `opts` of the forward kernels: bit 0 = the quad kernel may take its plain chunks (PLAIN_LOG2_T above; EOGS_PLAIN_TRIPS=0: never)
static int render_opts() {
  static const int v = [] {
    const char* e = getenv("EOGS_PLAIN_TRIPS");
    return (e && atoi(e) == 0) ? 0 : 1;
  }();
  return v;
}
#define EOGS_QUAD_SWITCH_DEFAULT 1.0e9
static double quad_switch() {
  static const double v = [] {
    const char* e = getenv("EOGS_QUAD_SWITCH");
    return e ? atof(e) : EOGS_QUAD_SWITCH_DEFAULT;
  }();
  return v;
}

// grid of a render launch: with a tile schedule 8 XCD sequences x lg blocks x 16 tiles, else the tiles rounded up to a multiple of 8
// (a forward that lists nothing has built no descriptors: launch_block_lists returned before its kernel)
static inline const uint4* render_desc(const ImgWS& im, int64_t R) { return (RBLK == 64 && nr_entries(R) > 0) ? im.desc : nullptr; }
static inline uint32_t render_grid(int ntiles, const ImgWS& im, int64_t R) {
  if (render_desc(im, R)) return 8u * 16u * im.sched_lg;
  const uint32_t groups = ceil_div_u32((uint64_t)ntiles, RBLK / 64);
  return ((groups + 7u) / 8u) * 8u;  // multiple of 8 for the XCD band mapping
}

int render_fwd_variant(int block, int64_t R, int P) {
  if (nr_alt(R)) return 2;  // altitude-only: the quad kernel's one-channel variant (the token carries per-tile lists)
  if (block > 1) return 1;
  return (quad_switch() > 0.0 && (double)nr_slots(R) <= quad_switch() * (double)P) ? 2 : 0;
}

void launch_render_fwd(const GeomWS& g, const BinWS& b, const ImgWS& im, int P, int H, int W, int64_t R,
                       const float* bg, float* out_color, float* out_invdepth, hipStream_t s) {
  const int gsx = (W + SUBX - 1) / SUBX, gsy = (H + SUBY - 1) / SUBY, ntiles = gsx * gsy;
  const int variant = render_fwd_variant(b.block, R, P);
  auto* kern = variant == 1 ? render_fwd_kernel<BLOCK_BIG> : (variant == 2 ? (nr_alt(R) ? render_fwd_quad_kernel<1, true> : (out_invdepth ? render_fwd_quad_kernel<1, false> : render_fwd_quad_kernel<1, false, false>)) : render_fwd_kernel<1>);
  // (the quad forward leaves its quad masks for the backward in BinWS::qmask, handed over in place of the keys it does not read)
  const uint32_t* keys = variant == 2 ? reinterpret_cast<const uint32_t*>(b.qmask) : b.sorted_keys;
  hipLaunchKernelGGL(kern, dim3(render_grid(ntiles, im, R)), dim3(RBLK), 0, s, im.ranges, keys, b.point_list, W, H, gsx,
                     ntiles, (int)macro_grid_x(W, b.block), render_desc(im, R), g.sched, (int)(16u * im.sched_lg), g.packed, bg, im.final_T, im.n_contrib,
                     out_color, out_invdepth, render_opts());
}

// ------------------------------------------------------------------------------------------------------
// Backward
// ------------------------------------------------------------------------------------------------------
namespace {

// Sum of 11 values over each group of 8 consecutive lanes; the group total is valid in the group's LAST lane (7, 15,
// ..., 63). Three DPP steps, row_shr 1/2/4 (bound_ctrl: sources outside the 16-lane row read 0): after them lane l holds
// the sum of lanes l-7..l, which for l = 8m+7 is exactly group m. Inline asm so each step is ONE v_add_f32_dpp per value
// (hipcc materialises "old = 0" moves otherwise); an asm statement is opaque to the hazard recogniser and a DPP read of
// a VGPR written by the previous VALU instruction needs 2 wait states, hence the leading s_nop 1; inside a block the 11
// chains are independent and every register is re-read 11 instructions after it was written.
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
#define DPP_STEP10(CTRL)                                                                                         \
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
               "v_add_f32_dpp %9, %9, %9 " CTRL                                                                   \
               : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), \
                 "+v"(c[8]), "+v"(c[9]))
#define DPP_STEP7(CTRL)                                                                                          \
  asm volatile("s_nop 1\n\t"                                                                                     \
               "v_add_f32_dpp %0, %0, %0 " CTRL "\n\t"                                                            \
               "v_add_f32_dpp %1, %1, %1 " CTRL "\n\t"                                                            \
               "v_add_f32_dpp %2, %2, %2 " CTRL "\n\t"                                                            \
               "v_add_f32_dpp %3, %3, %3 " CTRL "\n\t"                                                            \
               "v_add_f32_dpp %4, %4, %4 " CTRL "\n\t"                                                            \
               "v_add_f32_dpp %5, %5, %5 " CTRL "\n\t"                                                            \
               "v_add_f32_dpp %6, %6, %6 " CTRL                                                                   \
               : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]))
__device__ inline void group8_sum11(float (&c)[11]) {
  DPP_STEP11("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
  DPP_STEP11("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1");
  DPP_STEP11("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1");
}

// Transposition round: the wave has parked, for up to KSURV = 8 surviving entries k, the per-pixel pairs
//   u[k][p] = alpha T (weight of the colour gradient),  v[k][p] = G dL/dalpha
// in LDS. Now lane (k = lane >> 3, o = lane & 7) owns survivor k and pixel row o of the tile (8 pixels, constant
// dy) and accumulates  sum v, sum v dx, sum v dx^2  and  sum u g_p[ch]  serially in registers with plain FMAs on
// VGPR operands; the y-moments follow from the constant dy. A survivor's eight rows sit in eight consecutive lanes,
// so they are combined with three in-register DPP steps (group8_sum11) and lane o = 7 turns the moments M = sum_p v {1, dx, dy, dx^2, dx dy, dy^2} into the record
// (backward.cu:624-640):
//   dL/dmean2D = o (W/2, H/2) * (-(a M_dx + b M_dy), -(c M_dy + b M_dx)),  dL/dconic = -o/2 (M_dxdx, M_dxdy, M_dydy),
//   dL/dopacity = M_1,  dL/dcolour = sum u g.
// The survivors' geometry and record slots sit in the round buffer `rb` (8 words per survivor, written when the entry
// survived), so a round may span several list chunks.
__device__ inline void transpose_round(int nsurv, int lane, const float* rb, const float* s_u, const float* s_v,
                                       const float* s_pix, float bx0, float by0, float kx, float ky,
                                       float* __restrict__ records, uint8_t* __restrict__ live_flag, uint32_t rec_plane) {
  const int k = lane >> 3, o = lane & 7;
  const bool live = k < nsurv;
  const float4 q0 = *reinterpret_cast<const float4*>(rb + k * 8);      // gx gy A B
  const float4 q1 = *reinterpret_cast<const float4*>(rb + k * 8 + 4);  // C op slot -
  const float gxr = q0.x - bx0;            // centre relative to the tile origin
  const float dy = q0.y - (by0 + (float)o);
  float S0 = 0.f, Sx = 0.f, Sxx = 0.f;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, c4 = 0.f;
  const float* urow = s_u + uv_index(k, 8 * o);  // a pixel row (8 pixels) never straddles the two halves
  const float* vrow = s_v + uv_index(k, 8 * o);
  const float* prow = s_pix + (8 * o) * 8 + 4 * o;  // 8 floats per pixel, +4 floats per pixel row against bank conflicts
  float u8[8], v8[8];
  uv_row8(urow, u8);
  uv_row8(vrow, v8);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const float u = u8[i], v = v8[i];
    const float4 ga = *reinterpret_cast<const float4*>(prow + i * 8);
    const float gb = prow[i * 8 + 4];
    const float dx = gxr - (float)i;
    const float t1 = v * dx;
    S0 += v; Sx += t1; Sxx += t1 * dx;
    c0 += u * ga.x; c1 += u * ga.y; c2 += u * ga.z; c3 += u * ga.w; c4 += u * gb;
  }
  float acc[11] = {S0, Sx, dy * S0, Sxx, dy * Sx, dy * dy * S0, c0, c1, c2, c3, c4};
  group8_sum11(acc);
  if (live && o == 7) {
    const float A = q0.z, B = q0.w, Cq = q1.x, op = q1.y;
    const float m2x = op * kx * (2.f * A * acc[1] - B * acc[2]);   // -a = 2A/log2e, -b = -B/log2e
    const float m2y = op * ky * (2.f * Cq * acc[2] - B * acc[1]);  // -c = 2C/log2e
    const float ho = -0.5f * op;
    const uint32_t slot = __float_as_uint(q1.z);
    float4* r4 = reinterpret_cast<float4*>(records);  // layout: common.h REC, rec_q
    r4[rec_q(slot, 0, rec_plane, REC / 4)] = make_float4(m2x, m2y, ho * acc[3], acc[0]);
    r4[rec_q(slot, 1, rec_plane, REC / 4)] = make_float4(ho * acc[4], ho * acc[5], acc[6], acc[7]);
    reinterpret_cast<float3*>(r4 + rec_q(slot, 2, rec_plane, REC / 4))[0] = make_float3(acc[8], acc[9], acc[10]);
    live_flag[slot] = 1;  // pairs that never get here keep the 0 of the memset and are skipped by gaussian_bwd
  }
}

}  // namespace

// A peek for the back-to-front walks: chunks are taken from the list's end, so an index may also fall in front of the list.
template <int MACRO>
__device__ inline Peek peek_cand_r(uint32_t k, uint32_t begin, uint32_t end, const uint32_t* __restrict__ keys,
                                   const uint2* __restrict__ point_list) {
  return peek_cand<MACRO>(k >= begin ? k : 0xFFFFFFFFu, end, keys, point_list);
}

// HAVE_INV: an upstream gradient of the inverse-depth image exists (the reference always materialises a zero one,
// renderer.py:101 never consumes invdepths; here the common case compiles the term away).
// One list per tile (MACRO = 1) or per 32 x 32-px block (MACRO = BLOCK_BIG: the wave keeps the entries whose sub-mask lists its
// tile). Back to front (file header): the walk starts at the chunk that holds the tile's last contributor — n_contrib counts
// positions of the list the wave walks, render_fwd_kernel — and every eight surviving entries are transposed (transpose_round);
// a round may span chunks.
template <int MACRO, bool HAVE_INV>
__global__ __launch_bounds__(RBLK) void render_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ keys, const uint2* __restrict__ point_list, int W, int H, int gsx, int ntiles, int gmx, const uint4* __restrict__ desc, const uint32_t* __restrict__ sched, int lg16,
    const float4* __restrict__ packed, const uint32_t* __restrict__ n_contrib, const float* __restrict__ final_T,
    const float* __restrict__ bg, const float* __restrict__ dL_dpix, const float* __restrict__ dL_dinv,
    float* __restrict__ records, uint8_t* __restrict__ live_flag, uint32_t rec_plane) {
  __shared__ __attribute__((aligned(16))) float s_slab[RBLK / 64][64 * ENT];
  __shared__ __attribute__((aligned(16))) float s_round[RBLK / 64][KSURV * 8];
  // u and v matrices of a wave sit UV_PITCH floats (a multiple of 64 dwords) apart: one ds_write2st64_b32 stores both
  __shared__ __attribute__((aligned(16))) float s_uv[RBLK / 64][UV_PITCH + UV_SIZE];
  __shared__ __attribute__((aligned(16))) float s_pix[RBLK / 64][64 * 8 + 32];
  __shared__ uint32_t s_slot[RBLK / 64][64];
  const int lane = threadIdx.x & 63;
  uint2 range = make_uint2(0u, 0u);
  const int tile = tile_of_wave(desc, sched, lg16, gsx, range);
  if (tile >= ntiles) return;
  const int w = RBLK == 64 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* slab = s_slab[w];
  float* su = s_uv[w];
  float* sv = s_uv[w] + UV_PITCH;
  float* spix = s_pix[w];
  uint32_t* sslot = s_slot[w];
  float* rb = s_round[w];
  const int tx0 = (tile % gsx) * SUBX, ty0 = (tile / gsx) * SUBY;
  const int px = tx0 + (lane & 7), py = ty0 + (lane >> 3);
  const bool inside = px < W && py < H;
  const uint32_t pix_id = (uint32_t)py * (uint32_t)W + (uint32_t)px;
  const float pxf = (float)px, pyf = (float)py;
  const int ftx = tile % gsx, fty = tile / gsx;
  if (desc == nullptr) range = ranges[(fty / MACRO) * gmx + ftx / MACRO];  // the macro block's list
  const uint32_t sub = (uint32_t)((fty % MACRO) * MACRO + ftx % MACRO);
  const size_t HW = (size_t)H * W;

  float g[NCH] = {0.f, 0.f, 0.f, 0.f, 0.f};
  float ginv = 0.f, Tfin = 1.f, bgdot = 0.f;
  uint32_t ncontrib = 0;
  if (inside) {
    ncontrib = n_contrib[pix_id];
    Tfin = final_T[pix_id];
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
      g[ch] = dL_dpix[ch * HW + pix_id];
      bgdot += bg[ch] * g[ch];  // backward.cu:527-529
    }
    if (HAVE_INV) ginv = dL_dinv[pix_id];
  }
  {  // pixel gradients for the transposition rounds: 8 floats per pixel, each pixel row offset by 4 more floats
    float* d = spix + lane * 8 + 4 * (lane >> 3);
    *reinterpret_cast<float4*>(d) = make_float4(g[0], g[1], g[2], g[3]);
    d[4] = g[4];
  }
  // list entries past the last contributor of every pixel of the tile receive no gradient: never gathered, never written
  const uint32_t tile_last = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(ncontrib));
  const uint32_t n = min(range.y - range.x, tile_last);
  if (n == 0u) return;
  const uint32_t lend = range.x + n;
  const float kx = LN2 * 0.5f * W, ky = LN2 * 0.5f * H;  // d(pixel)/d(ndc) times the ln2 of the log2-domain conic
  const float bx0 = (float)tx0, by0 = (float)ty0;
  // The background is the colour behind the last contributor: `a` starts at bg . g instead of 0, and the reference's separate term
  // dL/dalpha += (-T_final / (1 - alpha)) (bg . g) (backward.cu:617-620) is inside T (gc - a) — T_final / (1 - alpha_j) is T_j times
  // the transmittance of the entries behind j, which is the weight the recursion gives the initial value (file header).
  float T = Tfin, a = bgdot;        // transmittance in front of the entry at hand; g . (colour behind it, background included)
  float* const uvlane = su + uv_index(0, lane);

  int k = 0, kstashed = 0;       // survivors waiting in the current transposition round (rounds span chunks), and how many
                                 // of them already have their geometry in the round buffer
  unsigned long long kj = 0ull;  // slab positions of the survivors not yet stashed, 8 bits each
  int c = (int)((n - 1u) / 64u) * 64;  // chunks of 64 list entries, the last one first
  Cand nxt = gather_cand<MACRO>(peek_cand_r<MACRO>(range.x + (uint32_t)c + lane, range.x, lend, keys, point_list), sub, packed);
  Peek pk = peek_cand_r<MACRO>(range.x + (uint32_t)(c - 64) + lane, range.x, lend, keys, point_list);
  for (; c >= 0; c -= 64) {
    wave_lds_sync();
    unsigned long long hm = __builtin_amdgcn_ballot_w64(nxt.hit);  // which of the chunk's entries list this tile
    const int jn = park(slab, sslot, lane, nxt, 0);
    nxt = gather_cand<MACRO>(pk, sub, packed);                                                        // chunk c-64: in flight during this chunk
    pk = peek_cand_r<MACRO>(range.x + (uint32_t)(c - 128) + lane, range.x, lend, keys, point_list);  // chunk c-128
    wave_lds_sync();
    if (jn == 0) continue;
    // survivors [from, to) of the current round live in this chunk's slab: lane i copies survivor i's geometry and
    // record slot into the round buffer (the slab is overwritten by the next chunk, the round may outlive it)
    auto stash = [&](int from, int to) {
      if (lane >= from && lane < to) {
        const uint32_t jk = (uint32_t)(kj >> (8 * lane)) & 63u;
        const float4 qa = *reinterpret_cast<const float4*>(slab + jk * ENT);
        const float2 qb = *reinterpret_cast<const float2*>(slab + jk * ENT + 4);
        *reinterpret_cast<float4*>(rb + lane * 8) = qa;
        *reinterpret_cast<float4*>(rb + lane * 8 + 4) = make_float4(qb.x, qb.y, __uint_as_float(sslot[jk]), 0.f);
      }
    };
    auto grad = [&](const Ent& e, int j) {
      // list position of slab entry j: the chunk's hit entries were parked in list order, so walking the slab backwards walks
      // the set bits of the hit mask from the top (wave-uniform: scalar instructions)
      uint32_t pos;
      if (MACRO > 1) {
        const int top = 63 - (int)__builtin_clzll(hm);
        hm &= ~(1ull << top);
        pos = (uint32_t)(c + top);
      } else {
        pos = (uint32_t)(c + j);
      }
      const float dx = e.q0.x - pxf, dy = e.q0.y - pyf;
      const float p = power_of(e, dx, dy);
      const float G = __builtin_amdgcn_exp2f(p);
      const float alpha = fminf(e.q1.y * G, 0.99f);
      // (contributor >= last_contributor: skip, backward.cu:561-563; then the forward's own two tests)
      const bool valid = (pos < ncontrib) && !(p > 0.0f) && !(alpha < 1.0f / 255.0f);
      // wave-uniform skip: this entry reaches no pixel of the tile (`valid` is an AND of three compare masks: its ballot is
      // that mask, no v_cndmask + v_cmp as for a general boolean)
      if (__builtin_amdgcn_ballot_w64(valid) == 0ull) return;

      float gc = g[0] * e.q1.z + g[1] * e.q1.w + g[2] * e.q2.x + g[3] * e.q2.y + g[4] * e.q2.z;
      if (HAVE_INV) gc += ginv * e.q2.w;
      else asm volatile("" :: "v"(e.q2.w));  // (keeps the entry's third read a ds_read_b128: see render_bwd_quad_kernel's grad)
      // pixels that skip this Gaussian behave as alpha = 0, G = 0 (selects, not multiplications: exp2 may have
      // overflowed there): weight 0, v = 0, T and a unchanged. No zeroing when alpha was clamped (backward.cu:624).
      const float a_eff = valid ? alpha : 0.f;
      const float G_eff = valid ? G : 0.f;
      const float rinv = __builtin_amdgcn_rcpf(1.f - a_eff);
      T = T * rinv;                 // backward.cu:573
      const float d = gc - a;       // (c_j - accum_rec_j) . g, backward.cu:586-609
      const float dLda = d * T;
      a = __builtin_fmaf(a_eff, d, a);
      float* const uv = uvlane + k * UV_ROW;  // = su + uv_index(k, lane); k is wave-uniform
      uv[0] = a_eff * T;
      uv[UV_PITCH] = G_eff * dLda;  // v = G dL/dalpha
      kj |= (unsigned long long)j << (8 * k);
      if (++k == KSURV) {
        stash(kstashed, KSURV);
        wave_lds_sync();
        transpose_round(KSURV, lane, rb, su, sv, spix, bx0, by0, kx, ky, records, live_flag, rec_plane);
        wave_lds_sync();
        k = 0;
        kstashed = 0;
        kj = 0ull;
      }
    };
    Ent ea = fetch(slab, jn - 1);
    int j = jn - 1;
    for (; j >= 1; j -= 2) {
      const Ent eb = fetch(slab, j - 1);
      grad(ea, j);
      ea = fetch(slab, j >= 2 ? j - 2 : 0);
      grad(eb, j - 1);
    }
    if (j == 0) grad(ea, 0);
    if (k > kstashed) {  // survivors waiting for the next chunk: keep what the round needs of them
      stash(kstashed, k);
      kstashed = k;
    }
  }
  if (k) {  // the last, partial round
    wave_lds_sync();
    transpose_round(k, lane, rb, su, sv, spix, bx0, by0, kx, ky, records, live_flag, rec_plane);
  }
}


// ------------------------------------------------------------------------------------------------------
// Backward with quad sub-lists (per-tile lists, small footprints)
// ------------------------------------------------------------------------------------------------------
// Same decomposition as render_fwd_quad_kernel: every quad (16 lanes, 4x4 pixels) walks its own sub-list of the chunk's
// entries, so one trip of the pixel-parallel loop evaluates four DIFFERENT entries and the trip count falls to the longest
// sub-list. A transposition round covers 8 trips: lane (k = lane >> 3, q = (lane >> 1) & 3, h = lane & 1) owns trip k,
// quad q and two pixel rows of that quad (8 pixels; the u/v matrix layout of render_bwd_kernel carries over because the
// pixel index is the lane in both phases), accumulates the six moments and five colour sums in registers and merges the
// two halves with one DPP step. An entry is visited by up to four quads in different trips, and its partial sums must end
// in ONE record: the (k, q) lanes park their 11 partials in a staging area (plain stores, aliasing the u matrix that has
// just been consumed), and the entry's OWNER lane (lane i = entry i of the chunk, which knows from the sub-list build at
// which trip each quad visits it) pulls them into 11 register accumulators, quads in fixed order. No LDS atomics (measured:
// ds_add_f32 merges made this kernel 2x slower), bitwise reproducible, and the record is written once per
// chunk by the owner exactly as transpose_round writes it.
namespace {

#define QB 68     // dwords per quad sub-list in backward (QLEAD + 64 + 2 spare); an entry is the BYTE OFFSET of its slab
                  // entry (position * 48) as a full dword: the hot loop's ds_read needs no address arithmetic or extraction
#define QLEAD 2   // ... of which the first two lie in front of the list: the backward walk's pipelined over-read
#define PIXB 296  // pixel gradients for the VALU transposition: channels 0..3 as one float4 per pixel (+1 float4 per 8 pixels
                  // against bank conflicts) in floats [0, 288), channel 4 at PIXB + pixel (read eight at a time: two ds_read_b128)
#define STG 12    // floats per staged (trip, quad) partial: 11 used

// Transposition of one round of `nk` trips (trips 8 r .. 8 r + nk - 1 of the chunk). ALT (altitude-only render): the one
// colour sum of channel 3, whose pixel gradients sit in the single plane at s_pix + PIXB; 7 partials per (trip, quad).
// ORG (round 5, EOGS_ORIGIN_MOMENTS): the six moments of v are taken in TILE-LOCAL pixel coordinates {1, X, Y, X^2, XY, Y^2}
// instead of about the Gaussian's centre: a row of four pixels costs three sums with constant weights (v1 + 2 v2 + 3 v3,
// v1 + 4 v2 + 9 v3) instead of a subtraction, a product and two multiply-adds per pixel, the lane needs neither the entry's
// centre nor its sub-list element (two LDS reads per round), and the four quads' partials add up in the owner as they are; the
// owner shifts the sums to the centre once per entry and chunk (sum v (g - X) = g S0 - SX, ...), as the MFMA variant always did.
#ifndef EOGS_ORIGIN_MOMENTS
#define EOGS_ORIGIN_MOMENTS 1
#endif
// NOC4 (raw-parameter renders, EOGS_FLAG_RAW_PARAMS): the fifth feature is the constant 1 (renderer.py:88-95), so nobody consumes its
// gradient: ten sums instead of eleven, no reads of the fifth channel's pixel gradients.
template <bool ALT, bool ORG, bool NOC4>
__device__ inline void transpose_round_quad(int nk, int r, int lane, const uint32_t* sidx, const float* slab, float* s_u,
                                            const float* s_v, const float* s_pix, float bx0, float by0) {
  const int k = lane >> 3, o = lane & 7, q = o >> 1, h = o & 1;
  float gxr = 0.f, dy0 = 0.f, dy1 = 0.f;
  if (!ORG) {
    const uint32_t off = sidx[q * QB + 8 * r + k];  // the entry quad q evaluated in trip k (any staged value is a valid offset)
    const float2 gxy = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(slab) + off);
    gxr = gxy.x - (bx0 + (float)(4 * (q & 1)));  // centre relative to the quad's first column
    dy0 = gxy.y - (by0 + (float)(4 * (q >> 1) + 2 * h));
    dy1 = dy0 - 1.f;
  }
  // !ORG: S0 / Sx / Sxx = sum v {1, dx, dx^2} of a row (dx = centre - pixel); ORG: s0 / s1 / s2 = sum v {1, x, x^2}, x = 0..3
  float S0a = 0.f, Sxa = 0.f, Sxxa = 0.f, S0b = 0.f, Sxb = 0.f, Sxxb = 0.f;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, c4 = 0.f;
  const float* urow = s_u + uv_index(k, 8 * o);
  const float* vrow = s_v + uv_index(k, 8 * o);
  const float* pa = s_pix + (8 * o) * 4 + 4 * o;
  // four pixels (one row of the quad) at a time: one ds_read_b128 each for u, v and the fifth channel's gradients (the same
  // address in all eight lanes of an o: a broadcast); the second row's reads stay behind the first row's arithmetic
  // (all eight pixels' operands in flight at once cost the kernel 52 bytes of scratch per lane)
#pragma unroll
  for (int hrow = 0; hrow < 2; hrow++) {
    const float4 u4 = *reinterpret_cast<const float4*>(urow + 4 * hrow), v4 = *reinterpret_cast<const float4*>(vrow + 4 * hrow);
    float4 gb4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!NOC4) gb4 = *reinterpret_cast<const float4*>(s_pix + PIXB + 8 * o + 4 * hrow);
    const float uu[4] = {u4.x, u4.y, u4.z, u4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, gg[4] = {gb4.x, gb4.y, gb4.z, gb4.w};
    if (ORG) {
      const float s0 = (vv[0] + vv[1]) + (vv[2] + vv[3]);
      const float s1 = __builtin_fmaf(3.f, vv[3], __builtin_fmaf(2.f, vv[2], vv[1]));
      const float s2 = __builtin_fmaf(9.f, vv[3], __builtin_fmaf(4.f, vv[2], vv[1]));
      if (hrow == 0) { S0a = s0; Sxa = s1; Sxxa = s2; }
      else { S0b = s0; Sxb = s1; Sxxb = s2; }
    }
#pragma unroll
    for (int x = 0; x < 4; x++) {
      const int i = 4 * hrow + x;
      const float u = uu[x], v = vv[x];
      if (!ORG) {
        const float dx = gxr - (float)x;
        const float t1 = v * dx;
        if (hrow == 0) { S0a += v; Sxa += t1; Sxxa += t1 * dx; }
        else { S0b += v; Sxb += t1; Sxxb += t1 * dx; }
      }
      if (!ALT) {
        const float4 ga = *reinterpret_cast<const float4*>(pa + i * 4);
        c0 += u * ga.x; c1 += u * ga.y; c2 += u * ga.z; c3 += u * ga.w;
      }
      if (!NOC4) c4 += u * gg[x];
    }
    if (hrow == 0) __builtin_amdgcn_sched_barrier(0);
  }
  float m[6];
  if (ORG) {  // rows Ya, Ya + 1 and columns X0 .. X0 + 3 of the tile: X = X0 + x
    const float X0 = (float)(4 * (q & 1)), Ya = (float)(4 * (q >> 1) + 2 * h), Yb = Ya + 1.f;
    const float s1 = Sxa + Sxb;
    const float xa = __builtin_fmaf(X0, S0a, Sxa), xb = __builtin_fmaf(X0, S0b, Sxb);  // sum v X of each row
    m[0] = S0a + S0b;
    m[1] = xa + xb;
    m[2] = __builtin_fmaf(Ya, S0a, Yb * S0b);
    m[3] = __builtin_fmaf(X0, m[1] + s1, Sxxa + Sxxb);  // sum v X^2 = s2 + 2 X0 s1 + X0^2 s0 = s2 + X0 (s1 + (s1 + X0 s0))
    m[4] = __builtin_fmaf(Ya, xa, Yb * xb);
    m[5] = __builtin_fmaf(Ya * Ya, S0a, (Yb * Yb) * S0b);
  } else {
    m[0] = S0a + S0b; m[1] = Sxa + Sxb; m[2] = dy0 * S0a + dy1 * S0b; m[3] = Sxxa + Sxxb; m[4] = dy0 * Sxa + dy1 * Sxb;
    m[5] = dy0 * dy0 * S0a + dy1 * dy1 * S0b;
  }
  if (ALT) {
    float c[7] = {m[0], m[1], m[2], m[3], m[4], m[5], c4};
    DPP_STEP7("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");  // lane h = 1 += lane h = 0
    if (h == 1 && k < nk) {
      float4* st = reinterpret_cast<float4*>(s_u + (k * 4 + q) * STG);
      st[0] = make_float4(c[0], c[1], c[2], c[3]);
      st[1] = make_float4(c[4], c[5], c[6], 0.f);
    }
    return;
  }
  float c[11] = {m[0], m[1], m[2], m[3], m[4], m[5], c0, c1, c2, c3, c4};
  if (NOC4) DPP_STEP10("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
  else DPP_STEP11("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");  // lane h = 1 += lane h = 0
  // every lane has read its u/v above (LDS instructions of a wave execute in order): the u matrix becomes the staging area
  if (h == 1 && k < nk) {
    float4* st = reinterpret_cast<float4*>(s_u + (k * 4 + q) * STG);
    st[0] = make_float4(c[0], c[1], c[2], c[3]);
    st[1] = make_float4(c[4], c[5], c[6], c[7]);
    if (NOC4) *reinterpret_cast<float2*>(st + 2) = make_float2(c[8], c[9]);  // (ds_write_b64: 6 LDS cycles where the b128 takes 13)
    else st[2] = make_float4(c[8], c[9], c[10], 0.f);
  }
}

}  // namespace

// -DEOGS_BWD_PHASES: where a wave of render_bwd_quad_kernel spends its residency (s_memtime at the phase boundaries, summed
// over all waves into g_bwd_phase[]; read with eogs_debug_bwd_phases). Diagnostics only: every sample waits for the wave's
// outstanding LDS operations, so the instrumented kernel is a few per cent slower and its prefetches are less effective.
#ifdef EOGS_BWD_PHASES
#define PHASE_TILES 65536
__device__ unsigned long long g_bwd_phase[PHASE_TILES][6];  // per tile (wave): no same-address atomics
#define PHASE_DECL unsigned long long ph_t = __builtin_readcyclecounter(), ph_acc[6] = {0, 0, 0, 0, 0, 0}
#define PHASE(i) do { const unsigned long long ph_n = __builtin_readcyclecounter(); ph_acc[i] += ph_n - ph_t; ph_t = ph_n; } while (0)
#define PHASE_FLUSH do { if (lane == 0 && tile < PHASE_TILES) { for (int i = 0; i < 6; i++) g_bwd_phase[tile][i] += ph_acc[i]; g_bwd_phase[tile][5] += 1ull; } } while (0)
#else
#define PHASE_DECL
#define PHASE(i)
#define PHASE_FLUSH
#endif
// ALT (HAVE_INV = false): the backward of an altitude-only render (EOGS_FLAG_ALT_ONLY). dL_dpix is a single plane
// (channel 3); one product instead of a five-term dot per pair, one colour sum instead of five in the
// transposition, 7 instead of 11 values through the DPP merge, the staging area and the owner pull, and a 32-byte record
// {mean2D.x, .y, conic.a, opacity | conic.b, conic.c, colour3, -} (REC_ALT) that gaussian_bwd_kernel<., true> reads.
#ifndef EOGS_BW
#define EOGS_BW 4  // waves per SIMD the quad backward is compiled for (10 KB of LDS per wave: four fit)
#endif
// Back to front (file header): the tile's chunks are taken from the one that holds its last contributor down to the list's
// head, and inside a chunk every quad walks its sub-list from the end. Trip j of a chunk is still slot j & 7 of round j >> 3, so
// the transposition, the staging area and the owner pull do not know the direction; the chunk's partial round comes first.
template <bool HAVE_INV, bool ALT, bool NOC4 = false>
__global__ __launch_bounds__(RBLK) __attribute__((amdgpu_waves_per_eu(EOGS_BW, EOGS_BW))) void render_bwd_quad_kernel(
    const uint2* __restrict__ ranges, const uint8_t* __restrict__ qmask, const uint2* __restrict__ point_list, int W, int H, int gsx, int ntiles, const uint4* __restrict__ desc, const uint32_t* __restrict__ sched, int lg16,
    const float4* __restrict__ packed, const uint32_t* __restrict__ n_contrib, const float* __restrict__ final_T,
    const float* __restrict__ bg, const float* __restrict__ dL_dpix,
    const float* __restrict__ dL_dinv, float* __restrict__ records, uint8_t* __restrict__ live_flag, uint32_t rec_plane,
    const uint32_t* __restrict__ misc, int opts) {
  // slab position 64 holds a DUMMY entry (opacity 0 -> alpha = 0 -> never valid): shorter sub-lists are padded with it
  __shared__ __attribute__((aligned(16))) float s_slab[RBLK / 64][65 * ENT];
  __shared__ __attribute__((aligned(16))) float s_uv[RBLK / 64][UV_PITCH + UV_SIZE];
  // 8 floats per pixel (+ padding) for the transposition
  __shared__ __attribute__((aligned(16))) float s_pix[RBLK / 64][6 * 64];
  __shared__ __attribute__((aligned(16))) uint32_t s_idx[RBLK / 64][4 * QB];
  static_assert(PIXB + 72 <= 6 * 64, "both pixel-gradient planes fit");
  static_assert(32 * STG <= UV_SIZE, "the staging area lives inside the u matrix");
  const int lane = threadIdx.x & 63;
  uint2 range = make_uint2(0u, 0u);
  const int tile = tile_of_wave(desc, sched, lg16, gsx, range);
  if (tile >= ntiles) return;
  WTRACE_BEGIN;
  const int w = RBLK == 64 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* slab = s_slab[w];
  float* su = s_uv[w];
  float* sv = s_uv[w] + UV_PITCH;
  float* spix = s_pix[w];
  uint32_t* sidx = s_idx[w];
  int ox, oy;
  quad_pixel(lane, ox, oy);
  const int tx0 = (tile % gsx) * SUBX, ty0 = (tile / gsx) * SUBY;
  const int px = tx0 + ox, py = ty0 + oy;
  const bool inside = px < W && py < H;
  const uint32_t pix_id = (uint32_t)py * (uint32_t)W + (uint32_t)px;
  const float pxf = (float)px, pyf = (float)py;
  if (desc == nullptr) range = ranges[tile];
  const size_t HW = (size_t)H * W;
  const int myq = lane >> 4;
  // A quad's sub-list starts two dwords into its QB: the pipelined walk reads two elements past the one it evaluates, and walking
  // backwards those lie IN FRONT of the list (kept valid slab offsets: the zeros of the initialisation below)
  const uint32_t* myidx = sidx + myq * QB + QLEAD;

  float g[NCH] = {0.f, 0.f, 0.f, 0.f, 0.f};
  float ginv = 0.f, Tfin = 1.f, bgdot = 0.f;
  uint32_t ncontrib = 0;
  static_assert(!ALT || !HAVE_INV, "the altitude-only variant has no inverse-depth output");
  static_assert(!NOC4 || !ALT, "ten sums: a five-channel render");
  constexpr bool ORG = EOGS_ORIGIN_MOMENTS != 0;  // tile-local moments in the transposition (transpose_round_quad)
  if (inside) {
    ncontrib = n_contrib[pix_id];
    Tfin = final_T[pix_id];
    if (ALT) {  // a single plane: the altitude image's gradient
      g[3] = dL_dpix[pix_id];
      bgdot = bg[3] * g[3];
    } else {
#pragma unroll
      for (int ch = 0; ch < NCH; ch++) {
        g[ch] = dL_dpix[ch * HW + pix_id];
        bgdot += bg[ch] * g[ch];  // backward.cu:527-529
      }
    }
    if (HAVE_INV) ginv = dL_dinv[pix_id];
  }
  if (ALT) {  // the one plane the altitude-only transposition reads
    spix[PIXB + lane] = g[3];
  } else {  // pixel gradients for the transposition rounds (pixel index = lane): see PIXB
    *reinterpret_cast<float4*>(spix + lane * 4 + 4 * (lane >> 3)) = make_float4(g[0], g[1], g[2], g[3]);
    spix[PIXB + lane] = g[4];
  }
  for (int t = lane; t < 4 * QB; t += 64) sidx[t] = 0u;  // over-read entries: valid offsets
  if (lane < ENT) slab[64 * ENT + lane] = 0.f;
  // flag-free records (common.h noflag_scene; opts bit 0: the caller's gaussian_bwd reads them that way): every chunk of the
  // list is walked and every entry's record written, whatever the pixels' last contributors are; no live flags
  const bool noflag = (opts & 1) != 0 && ((opts & 2) != 0 || noflag_scene(misc[MISC_OPW_LO], misc[MISC_OPW_HI], W, H));
  const uint32_t tile_last = noflag ? 0xFFFFFFFFu : (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(ncontrib));
  // entries behind the last contributor of every pixel are dead: never gathered, never written (flags), or zero records (flag-free)
  const uint32_t n = min(range.y - range.x, tile_last);
  if (n == 0u) {
    WTRACE_END(1, tile);
    return;
  }
  const uint32_t lend = range.x + n;
  const float bx0 = (float)tx0, by0 = (float)ty0;
  // The background is the colour behind the last contributor: `a` starts at bg . g instead of 0, and the reference's separate term
  // dL/dalpha += (-T_final / (1 - alpha)) (bg . g) (backward.cu:617-620) is inside T (gc - a) — T_final / (1 - alpha_j) is T_j times
  // the transmittance of the entries behind j, which is the weight the recursion gives the initial value (file header).
  float T = Tfin, a = bgdot;        // transmittance in front of the entry at hand; g . (colour behind it, background included)
  float* const uvlane = su + uv_index(0, lane);
  constexpr int ROWF = UV_ROW;  // floats between the u/v rows of consecutive trips of a round
  float* const vlane = uvlane + UV_PITCH;

  PHASE_DECL;
  // opts bit 3: the quad forward walked these lists and left its quad masks in BinWS::qmask (peek_cand_q); without it the masks
  // are computed here (quad_mask)
  const bool fwd_masks = (opts & 8) != 0 && qmask != nullptr;
  const uint8_t* const qm_bytes = fwd_masks ? qmask : nullptr;
  int c = (int)((n - 1u) / 64u) * 64;  // list position of the chunk at hand: the last one first
  Cand nxt = gather_cand<1>(peek_cand_q(range.x + (uint32_t)c + lane, range.x, lend, qm_bytes, point_list), 0u, packed);
  Peek pk = peek_cand_q(range.x + (uint32_t)(c - 64) + lane, range.x, lend, qm_bytes, point_list);
  // The chunk loop exists twice, one after the other (the forward's construction, render_fwd_quad_kernel). The walk begins in the
  // general loop; PLAIN: once every pixel of the tile contributes past the END of the chunk at hand — then `off < nc_off` holds
  // for every entry of it and of every chunk in front of it — the remaining chunks leave that test out: one v_cmp and one mask
  // operation less per trip, every value the same bits. (A tile that hangs over the image's edge keeps the general loop: its
  // outside pixels must never count as contributing — their T would be divided up without bound.)
  const uint32_t nc_min = __builtin_amdgcn_ballot_w64(!inside) != 0ull
                              ? 0u
                              : ~(uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(~ncontrib));  // earliest last contributor
  auto chunks = [&](auto plainc) {
  constexpr bool PLAIN = decltype(plainc)::value;
  for (; c >= 0; c -= 64) {
    if constexpr (!PLAIN) {
      if ((opts & 4) != 0 && (uint32_t)c + 64u <= nc_min) return;  // the plain loop takes over
    }
    wave_lds_sync();
    int nq[4];
    uint32_t myranks = 0xFFFFFFFFu;  // byte q: the trip in which quad q evaluates this lane's entry (0xFF: never)
    {
      // (the tile origin through an opaque copy: the four quads' box corners are then computed here, per chunk, instead of being
      // hoisted out of the chunk loop into eight VGPRs this kernel does not have — it spilled them)
      float bxq = bx0, byq = by0;
      asm volatile("" : "+v"(bxq), "+v"(byq));
      // (the forward's masks where it left them — of the chunks it walked: the others lie behind every pixel's stop entry, where
      // no evaluation is valid whatever the sub-lists hold)
      const uint32_t qm = fwd_masks ? nxt.qm : (nxt.hit ? quad_mask(nxt, bxq, byq) : 0u);
      // per-tile lists: the in-range entries (list positions < n) are lanes 0..jn-1, parked at their own lane index
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const bool in = nxt.hit && ((qm >> q) & 1u);
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(in);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (in) {
          sidx[q * QB + QLEAD + (int)rank] = (uint32_t)(lane * (4 * ENT));
          myranks = (myranks & ~(0xFFu << (8 * q))) | (rank << (8 * q));
        }
        nq[q] = (int)__popcll(bal);
      }
    }
    const uint32_t cur_slot = nxt.slot;
    const int jn = park(slab, nullptr, lane, nxt, 0);
    nxt = gather_cand<1>(pk, 0u, packed);                                                             // chunk c-64: in flight during this chunk
    pk = peek_cand_q(range.x + (uint32_t)(c - 128) + lane, range.x, lend, qm_bytes, point_list);  // chunk c-128
    wave_lds_sync();
    const int nmax = max(max(nq[0], nq[1]), max(nq[2], nq[3]));
    // pad the shorter sub-lists up to the next multiple of 8 trips with the dummy entry (transposition rounds read the
    // entry of every (trip, quad) of a round, and the hot loop then needs no "is my quad still active" test)
    {
      const int npad = (nmax + 7) & ~7;
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (lane < npad - nq[q]) sidx[q * QB + QLEAD + nq[q] + lane] = (uint32_t)(64 * 4 * ENT);
      wave_lds_sync();
    }
    // contributing list positions of this pixel, relative to the chunk, as a slab byte offset
    const uint32_t nc_off = min(ncontrib - min(ncontrib, (uint32_t)c), 64u) * (uint32_t)(4 * ENT);
    constexpr int NACC = ALT ? 7 : 11;
    float acc[11];
#pragma unroll
    for (int t = 0; t < 11; t++) acc[t] = 0.f;

    // one round: transposition of `nk` trips, then every owner lane pulls the partials of its entry
    auto round = [&](int r, int nk) {
      PHASE(1);
      wave_lds_sync();
      transpose_round_quad<ALT, ORG, NOC4>(nk, r, lane, sidx + QLEAD, slab, su, sv, spix, bx0, by0);
      wave_lds_sync();
      PHASE(2);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t rk = myranks >> (8 * q);  // bits 7..3: the round of quad q's trip (31: never), bits 2..0: its slot
        if (((rk >> 3) & 31u) == (uint32_t)r) {
          const float4* st = reinterpret_cast<const float4*>(su + ((rk & 7u) * 4u + (uint32_t)q) * STG);
          const float4 a0 = st[0], a1 = st[1];
          acc[0] += a0.x; acc[1] += a0.y; acc[2] += a0.z; acc[3] += a0.w;
          acc[4] += a1.x; acc[5] += a1.y; acc[6] += a1.z;
          if (ALT) asm volatile("" :: "v"(a1.w));  // (ds_read_b128, not ds_read_b96: see grad)
          if (!ALT && NOC4) {
            const float2 a2 = *reinterpret_cast<const float2*>(st + 2);
            acc[7] += a1.w;
            acc[8] += a2.x; acc[9] += a2.y;
          } else if (!ALT) {
            const float4 a2 = st[2];
            acc[7] += a1.w;
            acc[8] += a2.x; acc[9] += a2.y; acc[10] += a2.z;
            asm volatile("" :: "v"(a2.w));  // (a ds_read_b128, not the 8-cycle ds_read_b96 the unused fourth float would make of it)
          }
        }
      }
      wave_lds_sync();  // the next trips overwrite the u matrix
      PHASE(3);
    };
    auto fetch_off = [&](uint32_t off) {
      const float4* e4 = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(slab) + off);
      Ent e;
      e.q0 = e4[0];
      if (ALT) {  // C, opacity and the altitude feature: 12 of the other 32 bytes
        const float2 co = *reinterpret_cast<const float2*>(e4 + 1);
        e.q1 = make_float4(co.x, co.y, 0.f, 0.f);
        e.q2 = make_float4(0.f, reinterpret_cast<const float*>(e4 + 2)[1], 0.f, 0.f);
      } else {
        e.q1 = e4[1]; e.q2 = e4[2];
      }
      return e;
    };
    // one (pixel, entry) evaluation; `slot` = trip & 7 selects the row of the u/v matrices
    auto grad = [&](const Ent& e, uint32_t off, int slot) {
      const float dx = e.q0.x - pxf, dy = e.q0.y - pyf;
      const float p = power_of(e, dx, dy);
      const float G = __builtin_amdgcn_exp2f(p);
      const float alpha = fminf(e.q1.y * G, 0.99f);
      const bool valid = (PLAIN || off < nc_off) && !(p > 0.0f) && !(alpha < 1.0f / 255.0f);  // (the dummy: alpha = 0)
      // Keep the entry's third read a ds_read_b128 (4 LDS cycles) although 1/depth goes unused here: the compiler narrows it to
      // a ds_read_b96, which the LDS serves in 8 (MI355X_MICROARCH.md, LDS table), and this kernel keeps the LDS array busy for
      // 80 % of its cycles (profiles/r02_v23_lds: SQ_LDS_IDX_ACTIVE). An empty use HERE, where the other three floats of the
      // quarter are consumed — at the read itself it made the wave wait for the data it had just asked for.
      if (!ALT && !HAVE_INV) asm volatile("" :: "v"(e.q2.w));
      float gc = ALT ? g[3] * e.q2.y : g[0] * e.q1.z + g[1] * e.q1.w + g[2] * e.q2.x + g[3] * e.q2.y + g[4] * e.q2.z;
      if (HAVE_INV) gc += ginv * e.q2.w;
      // pixels that skip this entry behave as alpha = 0, G = 0 (selects, not multiplications: exp2 may have overflowed there):
      // rinv = 1 exactly, so T and a keep their values. No zeroing when alpha was clamped (backward.cu:624).
      const float a_eff = valid ? alpha : 0.f;
      const float G_eff = valid ? G : 0.f;
      const float rinv = __builtin_amdgcn_rcpf(1.f - a_eff);
      T = T * rinv;            // backward.cu:573
      const float d = gc - a;  // (c_j - accum_rec_j) . g, backward.cu:586-609
      const float dLda = d * T;
      a = __builtin_fmaf(a_eff, d, a);
      uvlane[slot * ROWF] = a_eff * T;
      vlane[slot * ROWF] = G_eff * dLda;  // v = G dL/dalpha
    };
    PHASE(0);
    if (nmax > 0 || noflag) {  // (flag-free: an entry that reaches no quad still gets its record of zeros)
      // Trips are never skipped: trip j is slot j & 7 of round j >> 3, walked from nmax - 1 down to 0. Rounds are unrolled (slots,
      // sub-list reads and u/v rows at immediate offsets); the entry of trip j - 1 and the sub-list element of trip j - 2 are in
      // flight during trip j. At the start of trip j: o0 = element j, o1 = element j - 1, ea = the entry at o0.
      const int nfull = nmax >> 3, rem = nmax & 7;
      uint32_t o0 = myidx[nmax - 1], o1 = myidx[nmax - 2];
      Ent ea = fetch_off(o0);
      if (rem) {  // the chunk's partial round: its highest trips, taken first
        const uint32_t* ip = myidx + 8 * nfull;
        int t = rem - 1;
        for (; t >= 1; t -= 2) {
          const Ent eb = fetch_off(o1);
          const uint32_t o2 = ip[t - 2];
          grad(ea, o0, t);
          ea = fetch_off(o2);
          const uint32_t o3 = ip[t - 3];
          grad(eb, o1, t - 1);
          o0 = o2;
          o1 = o3;
        }
        if (t == 0) {
          grad(ea, o0, 0);
          o0 = o1;
          o1 = ip[-2];
          ea = fetch_off(o0);
        }
        round(nfull, rem);
      }
      for (int r = nfull - 1; r >= 0; r--) {
        const uint32_t* ip = myidx + 8 * r;
#pragma unroll
        for (int t = 7; t >= 1; t -= 2) {
          const Ent eb = fetch_off(o1);
          const uint32_t o2 = ip[t - 2];
          grad(ea, o0, t);
          ea = fetch_off(o2);
          const uint32_t o3 = ip[t - 3];
          grad(eb, o1, t - 1);
          o0 = o2;
          o1 = o3;
        }
        round(r, KSURV);
      }
      // The prefetches that travel into the next chunk (its gathered entry, the peek behind it) are made "arrived" HERE, in front
      // of the record stores: loads and stores share one in-order counter, the loop's back edge copies the prefetched registers,
      // and a wait placed there has to wait for the stores issued just before it too — a store round trip at the end of every
      // chunk (found in the ISA: s_waitcnt vmcnt(0) behind the record stores). The loads have been in flight for the whole
      // chunk; the wait here costs nothing.
      asm volatile("" : "+v"(nxt.q0.x), "+v"(nxt.q1.x), "+v"(nxt.q2.x), "+v"(pk.e.x), "+v"(pk.e.y), "+v"(pk.qraw));
      // entry `lane`: accumulated moments -> record (backward.cu:624-640, as in transpose_round)
      bool any = false;
#pragma unroll
      for (int t = 0; t < NACC; t++) any = any || acc[t] != 0.f;
      if (lane < jn && (any || noflag)) {
        const float4 q0 = *reinterpret_cast<const float4*>(slab + lane * ENT);
        const float2 q1 = *reinterpret_cast<const float2*>(slab + lane * ENT + 4);
        const float A = q0.z, B = q0.w, Cq = q1.x, op = q1.y;
        if (ORG) {  // moments about the tile origin -> about the Gaussian centre: sum v (gx - x) = gx S0 - Sx, ...
          const float gxr = q0.x - bx0, gyr = q0.y - by0;
          const float S0 = acc[0], Sx = acc[1], Sy = acc[2], Sxx = acc[3], Sxy = acc[4], Syy = acc[5];
          const float Sdx = gxr * S0 - Sx, Sdy = gyr * S0 - Sy;
          acc[1] = Sdx;
          acc[2] = Sdy;
          acc[3] = gxr * (Sdx - Sx) + Sxx;
          acc[4] = gxr * Sdy - gyr * Sx + Sxy;
          acc[5] = gyr * (Sdy - Sy) + Syy;
        }
        float opq = op;
        asm volatile("" : "+v"(opq));  // (kx, ky formed here, once per chunk, not kept in registers across the trips)
        const float m2x = opq * (LN2 * 0.5f * W) * (2.f * A * acc[1] - B * acc[2]);
        const float m2y = opq * (LN2 * 0.5f * H) * (2.f * Cq * acc[2] - B * acc[1]);
        const float ho = -0.5f * op;
        constexpr int RQ = (ALT ? REC_ALT : REC) / 4;  // layout: common.h REC / REC_ALT, rec_q
        float4* r4 = reinterpret_cast<float4*>(records);
        r4[rec_q(cur_slot, 0, rec_plane, RQ)] = make_float4(m2x, m2y, ho * acc[3], acc[0]);
        if (ALT) {
          reinterpret_cast<float3*>(r4 + rec_q(cur_slot, 1, rec_plane, RQ))[0] = make_float3(ho * acc[4], ho * acc[5], acc[6]);
        } else {
          r4[rec_q(cur_slot, 1, rec_plane, RQ)] = make_float4(ho * acc[4], ho * acc[5], acc[6], acc[7]);
          reinterpret_cast<float3*>(r4 + rec_q(cur_slot, 2, rec_plane, RQ))[0] = make_float3(acc[8], acc[9], acc[10]);
        }
        if (!noflag) live_flag[cur_slot] = 1;
      }
    }
    PHASE(4);
  }
  };
  chunks(std::false_type{});
  chunks(std::true_type{});
  PHASE_FLUSH;
  WTRACE_END(1, tile);
}

static double quad_bwd_switch() {  // EOGS_QUAD_BWD_SWITCH=<listed tiles per Gaussian>, 0 disables the quad backward
  static const double v = [] {
    const char* e = getenv("EOGS_QUAD_BWD_SWITCH");
    return e ? atof(e) : EOGS_QUAD_SWITCH_DEFAULT;
  }();
  return v;
}

int render_bwd_variant(int block, int64_t R, int P) {
  if (nr_alt(R)) return 6;  // altitude-only: the quad backward's one-channel variant
  if (block > 1) return 1;
  return (quad_bwd_switch() > 0.0 && (double)nr_slots(R) <= quad_bwd_switch() * (double)P) ? 2 : 0;
}

// Does the backward that (block, R, P) selects write flag-free records (common.h noflag_scene)? 0 = no, 1 = where the scene
// allows, 3 = whatever the scene (EOGS_NOFLAG=2). The quad backward only. gaussian_bwd_kernel is told the same answer (api.hip).
int render_bwd_noflag_ok(int block, int64_t R, int P) {
  static const int mode = [] {  // EOGS_NOFLAG: 0 = never, 1 = where the scene allows (default), 2 = always (tests: correct for any scene)
    const char* e = getenv("EOGS_NOFLAG");
    return e ? atoi(e) : 1;
  }();
  const int v = render_bwd_variant(block, R, P);
  if (mode <= 0 || !(v == 2 || v == 6)) return 0;
  return mode >= 2 ? 3 : 1;
}
static bool fwd_masks_on() {  // EOGS_FWD_MASKS=0: the quad backward computes its quad masks itself (A/B)
  static const bool v = [] {
    const char* e = getenv("EOGS_FWD_MASKS");
    return !(e && atoi(e) == 0);
  }();
  return v;
}
// `opts` of render_bwd_quad_kernel: bits 0, 1 = render_bwd_noflag_ok(), bit 2 = plain chunks allowed (EOGS_PLAIN_TRIPS=0: never),
// bit 3 = the forward's quad masks are in BinWS::qmask
static int render_bwd_opts(int block, int64_t R, int P) { return render_bwd_noflag_ok(block, R, P) | ((render_opts() & 1) ? 4 : 0); }

void launch_render_bwd(const GeomWS& g, const BinWS& b, const ImgWS& im, int P, int H, int W, int64_t R,
                       const float* dL_dcolor, const float* dL_dinvdepth, const float* bg, bool raw, hipStream_t s) {
  const int gsx = (W + SUBX - 1) / SUBX, gsy = (H + SUBY - 1) / SUBY, ntiles = gsx * gsy;
  const int variant = render_bwd_variant(b.block, R, P);
  if (variant == 2 || variant == 6) {  // the quad backward (variant 6: altitude-only, no inverse-depth gradient: api.hip checks)
    auto* kq = dL_dinvdepth ? render_bwd_quad_kernel<true, false> : render_bwd_quad_kernel<false, false>;
    if (variant == 2 && raw) kq = dL_dinvdepth ? render_bwd_quad_kernel<true, false, true> : render_bwd_quad_kernel<false, false, true>;
    if (variant == 6) kq = render_bwd_quad_kernel<false, true>;
    const bool fwd_quad = render_fwd_variant(b.block, R, P) == 2 && b.qmask != nullptr && fwd_masks_on();
    hipLaunchKernelGGL(kq, dim3(render_grid(ntiles, im, R)), dim3(RBLK), 0, s, im.ranges, b.qmask, b.point_list, W, H, gsx,
                       ntiles, render_desc(im, R), g.sched, (int)(16u * im.sched_lg), g.packed, im.n_contrib, im.final_T, bg,
                       dL_dcolor, dL_dinvdepth, b.records, b.live, b.cap_slots, g.misc, render_bwd_opts(b.block, R, P) | (fwd_quad ? 8 : 0));
    return;
  }
  auto* kern = variant == 1 ? (dL_dinvdepth ? render_bwd_kernel<BLOCK_BIG, true> : render_bwd_kernel<BLOCK_BIG, false>)
                            : (dL_dinvdepth ? render_bwd_kernel<1, true> : render_bwd_kernel<1, false>);
  hipLaunchKernelGGL(kern, dim3(render_grid(ntiles, im, R)), dim3(RBLK), 0, s, im.ranges, b.sorted_keys, b.point_list, W, H, gsx,
                     ntiles, (int)macro_grid_x(W, b.block), render_desc(im, R), g.sched, (int)(16u * im.sched_lg), g.packed, im.n_contrib,
                     im.final_T, bg, dL_dcolor, dL_dinvdepth, b.records, b.live, b.cap_slots);
}

#ifdef EOGS_BWD_PHASES
extern "C" int eogs_debug_bwd_phases(unsigned long long* out8, int reset) {
  static unsigned long long host[PHASE_TILES][6];
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bwd_phase), sizeof(host)) != hipSuccess) return -1;
  for (int i = 0; i < 8; i++) out8[i] = 0;
  for (int t = 0; t < PHASE_TILES; t++) {
    for (int i = 0; i < 5; i++) out8[i] += host[t][i];
    out8[7] += host[t][5];  // waves
  }
  if (reset) {
    void* dptr = nullptr;
    if (hipGetSymbolAddress(&dptr, HIP_SYMBOL(g_bwd_phase)) != hipSuccess) return -1;
    if (hipMemset(dptr, 0, sizeof(host)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

#ifdef EOGS_WAVE_TRACE
// out: [tiles][4] = {start, end (100 MHz), shader-clock ticks, HW_ID | XCC_ID << 32} of the last launch in direction dir
extern "C" int eogs_debug_wave_trace(unsigned long long* out, int dir, int tiles) {
  if (dir < 0 || dir > 1 || tiles < 0 || tiles > WTRACE_TILES) return -1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_trace), sizeof(unsigned long long) * 4 * (size_t)tiles,
                             sizeof(unsigned long long) * 4 * WTRACE_TILES * (size_t)dir) == hipSuccess ? 0 : -1;
}
#endif

// ---- self test of the wave64 primitives (diagnostics) ----
__global__ void selftest_kernel(uint32_t* out) {
  const int lane = threadIdx.x & 63;
  uint32_t bad = 0;
  float c[11];
#pragma unroll
  for (int q = 0; q < 11; q++) c[q] = (float)(((lane * 37 + 11 * q + 5) % 101) - 50);  // freshly written VGPRs (hazard case)
  group8_sum11(c);
#pragma unroll
  for (int q = 0; q < 11; q++) {
    float ref = 0.f;
    for (int i = 0; i < 8; i++) ref += (float)(((((lane & ~7) + i) * 37 + 11 * q + 5) % 101) - 50);
    if ((lane & 7) == 7 && c[q] != ref) bad |= 1u;
  }
  if (wave_max_u32((uint32_t)lane * 3u) != 189u) bad |= 2u;
  if (bad) atomicOr(out, bad);
}
void launch_selftest(uint32_t* out, hipStream_t s) { hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(256), 0, s, out); }
