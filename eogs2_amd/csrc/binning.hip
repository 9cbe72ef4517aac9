// binning.hip — builds the per-tile, depth-ordered Gaussian lists.
//
// Replaces InclusiveSum + duplicateWithKeys + cub::DeviceRadixSort::SortPairs(64-bit keys, 32+log2(T) bits) +
// identifyTileRanges (DGR/cuda_rasterizer/rasterizer_impl.cu:70-138,280-320) with a pipeline that produces
// the SAME list order — by tile, then depth bits ascending, then Gaussian index — without any global sort on depth:
//
//   1. expand in Gaussian-ID order: one 16-byte ENTRY per (32 x 32-pixel block, Gaussian) = {block id | 16-bit sub-mask
//      of the block's 4 x 4 internal tiles that list the Gaussian, depth key, Gaussian id, first record slot}. The
//      per-Gaussian binning records are read in id order (coalesced: no gather), positions come from the scan of the
//      per-workgroup entry counts preprocess wrote;
//   2. ONE stable LSD radix sort of the entries on the block id only (ceil(log2 #blocks) bits: 10 at 1024^2): inside a
//      block the entries stay in Gaussian-id order;
//   3. block ranges + pairs per block from the sorted keys;
//   4. one workgroup per block orders ITS entries by depth key (stable LSD radix, 8-bit digits, only the digits in which the
//      listed Gaussians' keys differ: 3 for EOGS scenes) — a block holds a few thousand entries, so this is local work on
//      L2-resident data — and splits them into the block's 16 per-tile lists (ballot + prefix count per tile): the
//      render kernels get one {Gaussian id, record slot} list per internal 8 x 8 tile, exactly as before.
// Lists are built per INTERNAL tile (SUBX x SUBY pixels = one wave64) and only for the internal tiles of the reference's
// 16-px tile rect in which the Gaussian can reach alpha >= 1/255 (exact hit mask computed in preprocess): a subset of the
// reference's candidates that contains every (pixel, Gaussian) pair the reference blends, so results are unchanged.
// A (tile, Gaussian) pair is unique, so (tile, depth bits, index) is a total order: stable block sort of id-ordered
// entries + stable depth sort inside the block = the reference's stable 64-bit-key sort, bit for bit.
// (Rounds 1-2 sorted the P Gaussians by depth globally, gathered their records into depth order and sorted the R pairs by
// tile in two passes: 20 launches and 0.226 ms at 1 M Gaussians / 1024^2; tools/sort_yardstick.hip has rocPRIM's numbers
// for the same jobs.)
// With block lists (BLOCK_BIG, large footprints) step 4 only orders the entries; the render waves filter by sub-mask.
//
// Also here: the generic u32-key / u32-payload radix sort (knn.hip's Morton order).
// All radix passes: 256-thread workgroups (4 wave64); ranking is wave-private (ballot match on the digit bits, no
// workgroup barrier), items are re-ordered through LDS so each digit run is written contiguously. HBM-bound integer work.
#include "common.h"

namespace {

// Exclusive scan across the 256 threads of a workgroup; `total` = sum over the workgroup. s_w: 4 words of LDS.
__device__ inline uint32_t block_excl_scan(uint32_t v, uint32_t* s_w, uint32_t& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t inc = wave_incl_scan_u32(v);
  if (lane == 63) s_w[w] = inc;
  __syncthreads();
  const uint32_t w0 = s_w[0], w1 = s_w[1], w2 = s_w[2], w3 = s_w[3];
  const uint32_t pre = (w > 0 ? w0 : 0u) + (w > 1 ? w1 : 0u) + (w > 2 ? w2 : 0u);
  total = w0 + w1 + w2 + w3;
  __syncthreads();
  return pre + inc - v;
}

// rank of a lane's digit among the equal digits of its 64-entry group (lane order) and the group's count of that digit.
// Per bit: the lanes whose bit equals mine stay peers, peers &= ~(ballot ^ mine), mine = 0 / ~0 — one v_bfe_i32, one compare
// and one v_bitop3_b32 per half (the select-and-mask form the compiler makes of `bit ? m : ~m` is nine instructions).
template <int NBITS>
__device__ inline uint32_t match_rank(uint32_t d, bool livel, uint32_t& count) {
  const unsigned long long live = __ballot(livel);
  uint32_t plo = (uint32_t)live, phi = (uint32_t)(live >> 32);
#pragma unroll
  for (int bit = 0; bit < NBITS; bit++) {
    const int x = __builtin_amdgcn_sbfe((int)d, bit, 1);  // 0 or -1
    const unsigned long long m = __ballot(x != 0);
    plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t)m, (uint32_t)x, 0x90);  // a & ~(b ^ c)
    phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t)(m >> 32), (uint32_t)x, 0x90);
  }
  count = (uint32_t)__popc(plo) + (uint32_t)__popc(phi);
  return __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));  // peers in the lanes below mine
}

// the entry count of this forward as the device knows it (written by pblock_scan_kernel); 0 when it does not fit `cap`
__device__ inline uint32_t entries_on_device(const uint32_t* __restrict__ misc, uint32_t cap) {
  const uint32_t lo = misc[MISC_MACRO_LO], hi = misc[MISC_MACRO_HI];
  return (hi == 0u && lo <= cap) ? lo : 0u;
}

}  // namespace

// ---- exclusive scans of the per-workgroup pair and entry counts (-> record slots in Gaussian-id order, entry
//      positions in id order), their 64-bit totals, the key range and the number of depth digits: misc[] ----
// Single workgroup of 1024 threads, four workgroups' counts per thread per round (one round up to 1 M Gaussians): the
// kernel sits between preprocess and the host's readback, so what counts is its latency — few loads per thread, close
// together (with 256 threads x 16 counts the loads of a wave were 256 bytes apart: 10 us; now 7).
#define PS_T 1024
#define PS_PER 4
namespace {
__device__ inline unsigned long long wg_excl_scan_u64(unsigned long long v, unsigned long long* s_w, unsigned long long& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned long long inc = wave_incl_scan_u64(v);  // (DPP path, common.h: this one-workgroup kernel is all latency)
  if (lane == 63) s_w[w] = inc;
  __syncthreads();
  unsigned long long pre = 0ull;
  total = 0ull;
#pragma unroll
  for (int k = 0; k < PS_T / 64; k++) {
    if (k < w) pre += s_w[k];
    total += s_w[k];
  }
  __syncthreads();
  return pre + inc - v;
}
}  // namespace

__global__ __launch_bounds__(PS_T) void pblock_scan_kernel(uint32_t* __restrict__ pblock, uint32_t* __restrict__ pblockE,
                                                           const uint32_t* __restrict__ pbkey, uint32_t nblk,
                                                           uint32_t* __restrict__ misc) {
  __shared__ unsigned long long s_w[PS_T / 64];
  __shared__ uint32_t s_k[2][PS_T / 64];
  unsigned long long carry_t = 0ull, carry_e = 0ull, opw = 0ull;
  uint32_t kmax = 0, knmin = 0, err = 0;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (uint32_t b0 = 0; b0 < nblk; b0 += PS_T * PS_PER) {
    const uint32_t i0 = b0 + threadIdx.x * PS_PER;
    uint32_t v[PS_PER], e[PS_PER];
    unsigned long long sum_t = 0ull, sum_e = 0ull;
#pragma unroll
    for (int k = 0; k < PS_PER; k++) {
      v[k] = 0; e[k] = 0;
      if (i0 + k < nblk) {
        v[k] = pblock[i0 + k];
        const uint4 q = reinterpret_cast<const uint4*>(pbkey)[i0 + k];
        kmax = q.x > kmax ? q.x : kmax;
        knmin = q.y > knmin ? q.y : knmin;
        e[k] = q.z;
        opw += q.w & 0x7FFFFFFFu;  // bit 31: error flag of the workgroup
        err |= q.w >> 31;
      }
      sum_t += v[k];
      sum_e += e[k];
    }
    unsigned long long tot_t, tot_e;
    const unsigned long long ex_t = wg_excl_scan_u64(sum_t, s_w, tot_t);
    const unsigned long long ex_e = wg_excl_scan_u64(sum_e, s_w, tot_e);
    // (slots and entry positions are u32: the host rejects totals that do not fit)
    uint32_t run_t = (uint32_t)(carry_t + ex_t), run_e = (uint32_t)(carry_e + ex_e);
#pragma unroll
    for (int k = 0; k < PS_PER; k++) {
      if (i0 + k < nblk) { pblock[i0 + k] = run_t; pblockE[i0 + k] = run_e; }
      run_t += v[k];
      run_e += e[k];
    }
    carry_t += tot_t;
    carry_e += tot_e;
  }
  kmax = wave_max_u32_dpp(kmax);
  knmin = wave_max_u32_dpp(knmin);
  // (a lane's opw < 2^33: the low 24 bits and the rest summed apart, each exact in 32 bits)
  opw = (unsigned long long)wave_sum_u32_dpp((uint32_t)(opw & 0xFFFFFFull)) + ((unsigned long long)wave_sum_u32_dpp((uint32_t)(opw >> 24)) << 24);
  err = wave_or_u32_dpp(err);
  __shared__ unsigned long long s_o[PS_T / 64];
  __shared__ uint32_t s_err[PS_T / 64];
  if (lane == 0) { s_k[0][w] = kmax; s_k[1][w] = knmin; s_o[w] = opw; s_err[w] = err; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t a = s_k[0][0], b = s_k[1][0], er = s_err[0];
    unsigned long long ow = s_o[0];
    for (int i = 1; i < PS_T / 64; i++) {
      a = s_k[0][i] > a ? s_k[0][i] : a;
      b = s_k[1][i] > b ? s_k[1][i] : b;
      ow += s_o[i];
      er |= s_err[i];
    }
    misc[MISC_ERR] = er;  // every word the host or a later kernel reads is written here: the workspace needs no clearing
    misc[MISC_TAG] = MISC_TAG_VALUE;
    misc[MISC_OPW_LO] = (uint32_t)ow;
    misc[MISC_OPW_HI] = (uint32_t)(ow >> 32);
    misc[MISC_MACRO_LO] = (uint32_t)carry_e;
    misc[MISC_MACRO_HI] = (uint32_t)(carry_e >> 32);
    pblock[nblk] = (uint32_t)carry_t;
    pblockE[nblk] = (uint32_t)carry_e;
    misc[MISC_TOTAL_LO] = (uint32_t)carry_t;
    misc[MISC_TOTAL_HI] = (uint32_t)(carry_t >> 32);
    misc[MISC_KEY_MAX] = a;
    misc[MISC_KEY_NMIN] = b;
    // depth digits in which the listed Gaussians' keys can differ (EOGS altitudes span far less than a factor of two
    // around 200 - altitude: the keys share sign, exponent and the top mantissa bits -> 3 digits, not 4)
    const uint32_t diff = a ^ ~b;  // max ^ min
    misc[MISC_DEPTH_PASSES] = a == 0u ? 0u : (diff == 0u ? 0u : (uint32_t)((32 - __builtin_clz(diff) + 7) / 8));
  }
}

void launch_pblock_scan(const GeomWS& g, int P, hipStream_t s) {
  hipLaunchKernelGGL(pblock_scan_kernel, dim3(1), dim3(PS_T), 0, s, g.pblock, g.pblockE, g.pbkey, ceil_div_u32((uint64_t)P, BLK),
                     g.misc);
}

// =====================================================================================================================
// Generic radix passes: u32 keys + u32 payload (knn.hip)
// =====================================================================================================================

// ---- radix pass, kernel 1: per-workgroup digit histogram, written digit-major hist[d][blk] ----
template <int ITEMS>
__global__ __launch_bounds__(BLK) void radix_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, int shift,
                                                         uint32_t mask, uint32_t* __restrict__ hist, uint32_t nblk) {
  __shared__ uint32_t h[256];
  const int t = threadIdx.x;
  h[t] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * (uint32_t)(BLK * ITEMS);
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    const uint32_t k = base + i * BLK + t;
    if (k < n) atomicAdd(&h[(keys[k] >> shift) & mask], 1u);
  }
  __syncthreads();
  if ((uint32_t)t <= mask) hist[(size_t)t * nblk + blockIdx.x] = h[t];
}

// ---- radix pass, kernel 2: one workgroup per digit scans that digit's row over workgroups (exclusive) ----
__global__ __launch_bounds__(BLK) void radix_rowscan_kernel(uint32_t* __restrict__ hist, uint32_t nblk,
                                                            uint32_t* __restrict__ dtotal) {
  __shared__ uint32_t s_w[4];
  uint32_t* row = hist + (size_t)blockIdx.x * nblk;
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 < nblk; b0 += BLK) {
    const uint32_t i = b0 + threadIdx.x;
    const uint32_t v = i < nblk ? row[i] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan(v, s_w, tot);
    if (i < nblk) row[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) dtotal[blockIdx.x] = carry;
}

// ---- radix pass, kernel 3: stable scatter ----
// Wave w of the workgroup owns the contiguous segment [base + w*64*ITEMS, base + (w+1)*64*ITEMS) and takes it 64 items
// at a time (lane order == item order), so ranking needs NO workgroup barrier: per chunk a wave-level match by ballot
// on the digit bits gives the rank among equal digits inside the chunk, a wave-private LDS counter row gives the
// count of that digit in the wave's earlier chunks. One barrier later the workgroup knows, per digit, its start in the
// workgroup's sorted order and each wave's offset inside the digit; items are then placed in LDS in
// sorted order and streamed out so that consecutive lanes write consecutive addresses inside each digit run.
// Stable: (wave segment, chunk, lane) order is index order.
// ET: the item type moved; KEY(e) its sort key. n is either the argument or, with n_misc != NULL, this forward's entry
// count as the device knows it (the launch then covers the capacity and most workgroups leave at once).
struct KeyOfU32Pair {
  __device__ static inline uint32_t key(const uint2& e) { return e.x; }
};
struct KeyOfEntry {
  __device__ static inline uint32_t key(const uint4& e) { return e.x; }
};
template <int ITEMS, typename ET, typename KEY>
__device__ inline void radix_scatter_body(const ET* __restrict__ in, ET* __restrict__ out, uint32_t n, int shift, int nbits,
                                          const uint32_t* __restrict__ hist, uint32_t nblk,
                                          const uint32_t* __restrict__ dtotal) {
  constexpr int NW = BLK / 64, SEG = 64 * ITEMS, TILE_KEYS = BLK * ITEMS;
  __shared__ uint32_t s_gbase[256];     // global output position of this workgroup's first item of each digit
  __shared__ uint32_t s_dstart[256];    // start of each digit in the workgroup's sorted order
  __shared__ uint32_t s_wcnt[NW][256];  // per-wave digit counts -> per-wave offset inside the digit
  __shared__ ET s_item[TILE_KEYS];
  __shared__ uint32_t s_w[4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const uint32_t mask = (1u << nbits) - 1u;
#pragma unroll
  for (int k = 0; k < NW; k++) s_wcnt[k][t] = 0;
  __syncthreads();

  const uint32_t base = blockIdx.x * (uint32_t)TILE_KEYS + (uint32_t)w * SEG;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  ET item[ITEMS];
  uint32_t lrank[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    const uint32_t k = base + i * 64 + lane;
    if (k < n) item[i] = in[k];
  }
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    const bool live = base + i * 64 + lane < n;
    const uint32_t d = live ? (KEY::key(item[i]) >> shift) & mask : mask;
    unsigned long long peers = __ballot(live);
    for (int b = 0; b < nbits; b++) {
      const unsigned long long m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const uint32_t r = (uint32_t)__popcll(peers & lt_mask);
    uint32_t before = 0;
    if (live) before = s_wcnt[w][d];  // this wave's earlier chunks (LDS ops of one wave execute in order)
    __builtin_amdgcn_wave_barrier();
    if (live && r == 0) s_wcnt[w][d] = before + (uint32_t)__popcll(peers);  // one leader per digit
    __builtin_amdgcn_wave_barrier();
    lrank[i] = before + r;
  }
  __syncthreads();
  {  // thread t owns digit t
    uint32_t c[NW], tot = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) {
      c[k] = s_wcnt[k][t];
      s_wcnt[k][t] = tot;  // offset of wave k inside digit t
      tot += c[k];
    }
    uint32_t all;
    const uint32_t ex = block_excl_scan(tot, s_w, all);
    s_dstart[t] = ex;
    const uint32_t gtot = (uint32_t)t <= mask ? dtotal[t] : 0u;
    const uint32_t gex = block_excl_scan(gtot, s_w, all);
    s_gbase[t] = (uint32_t)t <= mask ? gex + hist[(size_t)t * nblk + blockIdx.x] : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    if (base + i * 64 + lane < n) {
      const uint32_t d = (KEY::key(item[i]) >> shift) & mask;
      s_item[s_dstart[d] + s_wcnt[w][d] + lrank[i]] = item[i];
    }
  }
  __syncthreads();
  const uint32_t tile0 = blockIdx.x * (uint32_t)TILE_KEYS;
  const uint32_t nvalid = n - tile0 < (uint32_t)TILE_KEYS ? n - tile0 : (uint32_t)TILE_KEYS;
  for (uint32_t idx = t; idx < nvalid; idx += BLK) {
    const ET e = s_item[idx];
    const uint32_t d = (KEY::key(e) >> shift) & mask;
    out[s_gbase[d] + (idx - s_dstart[d])] = e;
  }
}

// (the u32 + u32 sort keeps keys and payloads in separate arrays: it packs them for the LDS re-order only)
template <int ITEMS>
__global__ __launch_bounds__(BLK) void radix_scatter_u32_kernel(const uint32_t* __restrict__ keys_in,
                                                                const uint32_t* __restrict__ vals_in,
                                                                uint32_t* __restrict__ keys_out,
                                                                uint32_t* __restrict__ vals_out, uint32_t n, int shift,
                                                                int nbits, const uint32_t* __restrict__ hist, uint32_t nblk,
                                                                const uint32_t* __restrict__ dtotal) {
  constexpr int NW = BLK / 64, SEG = 64 * ITEMS, TILE_KEYS = BLK * ITEMS;
  __shared__ uint32_t s_gbase[256];
  __shared__ uint32_t s_dstart[256];
  __shared__ uint32_t s_wcnt[NW][256];
  __shared__ uint32_t s_key[TILE_KEYS];
  __shared__ uint32_t s_val[TILE_KEYS];
  __shared__ uint32_t s_w[4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const uint32_t mask = (1u << nbits) - 1u;
#pragma unroll
  for (int k = 0; k < NW; k++) s_wcnt[k][t] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * (uint32_t)TILE_KEYS + (uint32_t)w * SEG;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint32_t key[ITEMS], val[ITEMS], lrank[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    const uint32_t k = base + i * 64 + lane;
    key[i] = k < n ? keys_in[k] : 0xFFFFFFFFu;
    val[i] = k < n ? vals_in[k] : 0u;
  }
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    const bool live = base + i * 64 + lane < n;
    const uint32_t d = (key[i] >> shift) & mask;
    unsigned long long peers = __ballot(live);
    for (int b = 0; b < nbits; b++) {
      const unsigned long long m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const uint32_t r = (uint32_t)__popcll(peers & lt_mask);
    uint32_t before = 0;
    if (live) before = s_wcnt[w][d];
    __builtin_amdgcn_wave_barrier();
    if (live && r == 0) s_wcnt[w][d] = before + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    lrank[i] = before + r;
  }
  __syncthreads();
  {
    uint32_t c[NW], tot = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) {
      c[k] = s_wcnt[k][t];
      s_wcnt[k][t] = tot;
      tot += c[k];
    }
    uint32_t all;
    const uint32_t ex = block_excl_scan(tot, s_w, all);
    s_dstart[t] = ex;
    const uint32_t gtot = (uint32_t)t <= mask ? dtotal[t] : 0u;
    const uint32_t gex = block_excl_scan(gtot, s_w, all);
    s_gbase[t] = (uint32_t)t <= mask ? gex + hist[(size_t)t * nblk + blockIdx.x] : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    if (base + i * 64 + lane < n) {
      const uint32_t d = (key[i] >> shift) & mask;
      const uint32_t pos = s_dstart[d] + s_wcnt[w][d] + lrank[i];
      s_key[pos] = key[i];
      s_val[pos] = val[i];
    }
  }
  __syncthreads();
  const uint32_t tile0 = blockIdx.x * (uint32_t)TILE_KEYS;
  const uint32_t nvalid = n - tile0 < (uint32_t)TILE_KEYS ? n - tile0 : (uint32_t)TILE_KEYS;
  for (uint32_t idx = t; idx < nvalid; idx += BLK) {
    const uint32_t k = s_key[idx];
    const uint32_t d = (k >> shift) & mask;
    const uint32_t pos = s_gbase[d] + (idx - s_dstart[d]);
    keys_out[pos] = k;
    vals_out[pos] = s_val[idx];
  }
}

void launch_sort_u32(uint32_t* keyA, uint32_t* valA, uint32_t* keyB, uint32_t* valB, uint32_t n, int passes, uint32_t* hist,
                     uint32_t nblk, uint32_t* dtotal, hipStream_t s) {
  for (int pass = 0; pass < passes; pass++) {
    const bool a2b = (pass & 1) == 0;
    const uint32_t* kin = a2b ? keyA : keyB;
    const uint32_t* vin = a2b ? valA : valB;
    uint32_t* kout = a2b ? keyB : keyA;
    uint32_t* vout = a2b ? valB : valA;
    hipLaunchKernelGGL((radix_hist_kernel<SORTP_ITEMS>), dim3(nblk), dim3(BLK), 0, s, kin, n, 8 * pass, 255u, hist, nblk);
    hipLaunchKernelGGL(radix_rowscan_kernel, dim3(256), dim3(BLK), 0, s, hist, nblk, dtotal);
    hipLaunchKernelGGL((radix_scatter_u32_kernel<SORTP_ITEMS>), dim3(nblk), dim3(BLK), 0, s, kin, vin, kout, vout, n, 8 * pass,
                       8, hist, nblk, dtotal);
  }
}

// =====================================================================================================================
// List entries: expand (id order), block sort, block ranges
// =====================================================================================================================

#define EXPAND_LANE_MAX 256u  // a single lane walks at most this many list entries
#define EXPAND_STAGE 2048     // entries per LDS window (32 KB + 4 KB of owner lanes)
struct ExpandItem {
  uint32_t id, c, pos0, rbase, sx0, sy0, sx1, sy1, kind, depth;
  unsigned long long m;
};
// One lane per Gaussian, in id order: workgroup b = preprocess workgroup b, so the per-workgroup prefixes pblock / pblockE
// apply as they are and every input is read coalesced. An entry = (macro block, Gaussian): key = block id | sub-mask << 16
// (which internal tiles of the block list the Gaussian), then the Gaussian's depth key, its id and the record slot of the
// entry's first listed internal tile (the q-th set bit of the sub-mask owns slot + q; slots run through a Gaussian's
// entries in emission order and are Gaussian-id ordered across Gaussians, see GeomWS::pblock). The blocks come from
// walk_macro_row (common.h), the same code preprocess counted with.
// Gaussians with at most EXPAND_LANE_MAX entries are walked by their own lane into an LDS window and streamed out with
// consecutive lanes writing consecutive addresses; larger ones are emitted by the whole wave, one macro row per lane.
template <int MACRO>
__global__ __launch_bounds__(BLK, 4) void expand_entries_kernel(const uint4* __restrict__ binfo, const float4* __restrict__ bext,
                                                                const uint32_t* __restrict__ pblock,
                                                                const uint32_t* __restrict__ pblockE,
                                                                const uint32_t* __restrict__ misc, uint32_t cap, uint32_t P,
                                                                uint32_t gmx, uint4* __restrict__ ent) {
  if (entries_on_device(misc, cap) == 0u) return;  // nothing listed, or more entries than this buffer holds
  __shared__ uint32_t s_w[4];
  const int lane = threadIdx.x & 63;
  const uint32_t k = blockIdx.x * BLK + threadIdx.x;
  ExpandItem it;
  it.id = k; it.c = 0; it.m = 0ull; it.sx0 = it.sy0 = it.sx1 = it.sy1 = 0; it.rbase = 0; it.kind = BK_RECT; it.depth = 0;
  float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = e0;  // SpanParams of a BK_SPANS Gaussian
  uint4 ia = make_uint4(0u, 0u, 0u, 0u), ib = ia;
  if (k < P) {
    ia = binfo[(size_t)k];              // (two planes of P records: common.h GeomWS::binfo)
    ib = binfo[(size_t)P + (size_t)k];
    it.c = ib.x ? (ib.w >> 2) : 0u;
  }
  uint32_t tot;
  it.pos0 = pblockE[blockIdx.x] + block_excl_scan(it.c, s_w, tot);
  if (it.c) {
    it.m = ((unsigned long long)ia.w << 32) | ia.z;
    it.sx0 = ia.x & 0xFFFFu; it.sy0 = ia.y & 0xFFFFu;  // internal-tile rect, already clipped (preprocess_fwd_kernel)
    it.sx1 = ia.x >> 16; it.sy1 = ia.y >> 16;
    it.rbase = pblock[blockIdx.x] + ib.y;
    it.depth = ib.z;
    it.kind = ib.w & 3u;
    if (it.kind == BK_SPANS) {
      e0 = bext[2 * (size_t)k];
      e1 = bext[2 * (size_t)k + 1];
    }
  }
  SpanParams sp;
  sp.gx = e0.x; sp.gy = e0.y; sp.ex = e0.z; sp.ey = e0.w; sp.boa = e1.x; sp.boc = e1.y; sp.ta = e1.z; sp.da = e1.w;

  // ---- lane-walked Gaussians: their entries are staged in LDS in output order (EXPAND_STAGE per round, one round at the
  //      usual 1.7 entries per Gaussian) and written out as whole 16-byte entries with consecutive lanes on consecutive
  //      addresses (lanes storing their own entries straight to memory measured slower: 2.3x the bytes reach HBM as
  //      partial lines) ----
  __shared__ uint4 s_ent[EXPAND_STAGE];
  // Lane-walked: a Gaussian with at most EXPAND_LANE_MAX entries — unless it is a row-span footprint of three macro rows or more
  // in a wave where walking such footprints TOGETHER is cheaper: in its own lane a footprint costs rows x (four row spans + the
  // block loop) instructions while the wave's other lanes wait for the tallest one; walked by the whole wave below, one macro row
  // per lane, each costs about one row's worth, one after the other. The wave compares the two estimates (instructions:
  // max over lanes of rows x (100 + 50 blocks per row) against the sum over those lanes of 350 + 50 blocks per row). With the
  // size-heterogeneous Gaussians of a trained scene two waves in three hold one or two such lanes (1 M surface-shaped Gaussians:
  // expand 130 us against 25 at the uniform synthetic scene, profiles/r06_surface_front_end.txt); where a wave holds many
  // (large images, every footprint wide) they keep running in parallel in their lanes.
  const int mrows = it.c != 0u ? ((int)it.sy1 - 1) / MACRO - (int)it.sy0 / MACRO + 1 : 0;
  const int mcols = it.c != 0u ? ((int)it.sx1 - 1) / MACRO - (int)it.sx0 / MACRO + 1 : 0;
  const bool tall = it.c != 0u && it.kind != BK_MASK && mrows >= 3 && it.c <= EXPAND_LANE_MAX;
  const uint32_t walk_cost = wave_max_u32_dpp(tall ? (uint32_t)(mrows * (100 + 50 * mcols)) : 0u);
  const uint32_t coop_cost = wave_sum_u32_dpp(tall ? (uint32_t)(350 + 50 * mcols) : 0u);
  const bool together = coop_cost < walk_cost;
  const bool mine = it.c != 0u && it.c <= EXPAND_LANE_MAX && !(tall && together);
  uint32_t ltot;
  const uint32_t l0 = block_excl_scan(mine ? it.c : 0u, s_w, ltot);  // local position among the lane-walked entries
  // the lane-walked entries of this workgroup are NOT contiguous in the output when a large Gaussian sits between them:
  // every staged entry carries its own output position in the owner's `pos0 - l0` (s_gp)
  __shared__ uint32_t s_gp[BLK];
  __shared__ uint16_t s_own[EXPAND_STAGE];
  s_gp[threadIdx.x] = it.pos0 - l0;
  for (uint32_t base = 0; base < ltot; base += EXPAND_STAGE) {
    __syncthreads();  // s_gp visible (first round) / previous window drained
    if (mine && l0 < base + EXPAND_STAGE && l0 + it.c > base) {
      uint32_t l = l0, slot = it.rbase;
      for (int MY = (int)it.sy0 / MACRO; MY <= ((int)it.sy1 - 1) / MACRO; MY++)  // MACRO: template parameter
        walk_macro_row<MACRO>(it.kind, it.m, sp, (int)it.sx0, (int)it.sy0, (int)it.sx1, (int)it.sy1, MY, [&](int MX, uint32_t sub) {
          const uint32_t w = l - base;  // wraps below the window: fails the unsigned test
          if (w < (uint32_t)EXPAND_STAGE) {
            s_ent[w] = make_uint4(((uint32_t)MY * gmx + (uint32_t)MX) | (sub << MACRO_KEY_BITS), it.depth, it.id, slot);
            s_own[w] = (uint16_t)threadIdx.x;
          }
          l++;
          slot += (uint32_t)__popc(sub);
        });
    }
    __syncthreads();
    const uint32_t nwin = ltot - base < (uint32_t)EXPAND_STAGE ? ltot - base : (uint32_t)EXPAND_STAGE;
    for (uint32_t i = threadIdx.x; i < nwin; i += BLK) ent[s_gp[s_own[i]] + base + i] = s_ent[i];
  }

  // ---- large Gaussians: the wave emits them cooperatively, one after the other, one macro row per lane ----
  unsigned long long big = __ballot(it.c != 0u && !mine);
  while (big) {
    const int src = __builtin_ctzll(big);
    big &= big - 1ull;
    ExpandItem g;
    g.id = __shfl(it.id, src, 64); g.c = __shfl(it.c, src, 64); g.pos0 = __shfl(it.pos0, src, 64);
    g.rbase = __shfl(it.rbase, src, 64); g.sx0 = __shfl(it.sx0, src, 64); g.sy0 = __shfl(it.sy0, src, 64);
    g.sx1 = __shfl(it.sx1, src, 64); g.sy1 = __shfl(it.sy1, src, 64); g.kind = __shfl(it.kind, src, 64);
    g.depth = __shfl(it.depth, src, 64);
    const uint32_t mlo = __shfl((uint32_t)it.m, src, 64), mhi = __shfl((uint32_t)(it.m >> 32), src, 64);
    g.m = ((unsigned long long)mhi << 32) | mlo;
    SpanParams gs;
    gs.gx = __shfl(sp.gx, src, 64); gs.gy = __shfl(sp.gy, src, 64); gs.ex = __shfl(sp.ex, src, 64);
    gs.ey = __shfl(sp.ey, src, 64); gs.boa = __shfl(sp.boa, src, 64); gs.boc = __shfl(sp.boc, src, 64);
    gs.ta = __shfl(sp.ta, src, 64); gs.da = __shfl(sp.da, src, 64);
    const int MY0 = (int)g.sy0 / MACRO, MY1 = ((int)g.sy1 - 1) / MACRO;
    uint32_t done = 0, slot0 = g.rbase;  // entries / record slots of this Gaussian emitted so far
    for (int r0 = MY0; r0 <= MY1; r0 += 64) {
      const int MY = r0 + lane;
      uint32_t ne = 0, nf = 0;  // this lane's macro row: entries and listed internal tiles
      if (MY <= MY1) {
        if (g.kind == BK_MASK)
          walk_macro_row<MACRO>(g.kind, g.m, gs, (int)g.sx0, (int)g.sy0, (int)g.sx1, (int)g.sy1, MY, [&](int, uint32_t sub) {
            ne++;
            nf += (uint32_t)__popc(sub);
          });
        else
          count_macro_row<MACRO>(g.kind, gs, (int)g.sx0, (int)g.sy0, (int)g.sx1, (int)g.sy1, MY, ne, nf);  // (the same count, common.h)
      }
      const uint32_t ie = wave_incl_scan_u32(ne), jf = wave_incl_scan_u32(nf);
      uint32_t l = done + ie - ne, slot = slot0 + jf - nf;
      if (MY <= MY1)
        walk_macro_row<MACRO>(g.kind, g.m, gs, (int)g.sx0, (int)g.sy0, (int)g.sx1, (int)g.sy1, MY, [&](int MX, uint32_t sub) {
          if (l < g.c)  // always (same walk as the count)
            ent[g.pos0 + l] = make_uint4(((uint32_t)MY * gmx + (uint32_t)MX) | (sub << MACRO_KEY_BITS), g.depth, g.id, slot);
          l++;
          slot += (uint32_t)__popc(sub);
        });
      done += __shfl(ie, 63, 64);
      slot0 += __shfl(jf, 63, 64);
    }
  }
}

// ---- block sort: ONE stable counting pass on the whole block id when it has at most ES_MAXBITS bits (4096 blocks:
//      every image up to 2048 x 2048 pixels), otherwise two passes of half the bits each. 8192 entries per workgroup
//      (1024 threads x 8): the histogram tables stay small although the launch covers the buffer's capacity, and a
//      workgroup's entries of one block form runs of several 16-byte entries in the output. ----
#define ES_T 1024
#ifndef ES_ITEMS
#define ES_ITEMS 8
#endif
#define ES_TILE (ES_T * ES_ITEMS)
#define ES_NW (ES_T / 64)
#define ES_MAXBITS 12
static_assert(ES_TILE == SORTE_TILE, "sort_layout sizes the histogram tables for this tile");

// kernel 1: per-workgroup histogram of the digit, one ROW per workgroup (hist[blk][digit]: written and later read back as
// one contiguous piece); with `histp` also the listed (internal tile, Gaussian) pairs per digit (single-pass sort: digit =
// block, so the column sums are the blocks' entry and pair counts and nobody has to find the block boundaries in the
// sorted array afterwards). Workgroups past the live entries leave at once; nobody reads their rows.
__global__ __launch_bounds__(ES_T) void entry_hist_kernel(const uint4* __restrict__ ent, const uint32_t* __restrict__ misc,
                                                          uint32_t cap, int shift, uint32_t mask, uint32_t* __restrict__ hist,
                                                          uint32_t* __restrict__ histp) {
  const uint32_t n = entries_on_device(misc, cap);
  const int t = threadIdx.x;
  const uint32_t base = blockIdx.x * (uint32_t)ES_TILE, nb = mask + 1u;
  if (base >= n) return;  // (a launch covers the buffer's capacity)
  __shared__ uint32_t h[1 << ES_MAXBITS], hp[1 << ES_MAXBITS];
  for (uint32_t d = t; d < nb; d += ES_T) { h[d] = 0; hp[d] = 0; }
  __syncthreads();
  // (all of a thread's keys are requested before the first is used: with load and LDS atomic in one bounds check the compiler
  // waits for each load before it issues the next — ES_ITEMS serialised round trips in a kernel that is little else)
  uint32_t keyv[ES_ITEMS];
#pragma unroll
  for (int i = 0; i < ES_ITEMS; i++) {
    const uint32_t k = base + i * ES_T + t;
    keyv[i] = k < n ? ent[k].x : 0u;
  }
#pragma unroll
  for (int i = 0; i < ES_ITEMS; i++) {
    const uint32_t k = base + i * ES_T + t;
    if (k < n) {
      const uint32_t key = keyv[i], d = (key >> shift) & mask;
      atomicAdd(&h[d], 1u);
      if (histp) atomicAdd(&hp[d], (uint32_t)__popc(key >> MACRO_KEY_BITS));
    }
  }
  __syncthreads();
  for (uint32_t d = t; d < nb; d += ES_T) {
    hist[(size_t)blockIdx.x * nb + d] = h[d];
    if (histp) histp[(size_t)blockIdx.x * nb + d] = hp[d];
  }
}

// kernel 2: per digit, exclusive scan of its column over the live workgroups + the column totals. One workgroup takes 64
// digits (lane = digit: every access is a coalesced 256-byte piece of a row); its 16 waves each take a slab of rows.
__global__ __launch_bounds__(ES_T) void entry_colscan_kernel(uint32_t* __restrict__ hist, const uint32_t* __restrict__ histp,
                                                             const uint32_t* __restrict__ misc, uint32_t cap, uint32_t nb,
                                                             uint32_t* __restrict__ dtotal, uint32_t* __restrict__ ptotal) {
  const uint32_t n = entries_on_device(misc, cap);
  const uint32_t nlive = (n + ES_TILE - 1u) / ES_TILE;  // rows that were written
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t d = blockIdx.x * 64u + (uint32_t)lane;
  __shared__ uint32_t s_part[ES_NW][64], s_pp[ES_NW][64];
  const uint32_t slab = (nlive + ES_NW - 1u) / ES_NW;
  const uint32_t r0 = (uint32_t)w * slab, r1 = r0 + slab < nlive ? r0 + slab : nlive;
  // (rows eight at a time, all eight loads requested before the first is added: a loop of load-and-add waits for every row)
  constexpr uint32_t CS_CH = 8;
  uint32_t sum = 0, psum = 0;
  if (d < nb)
    for (uint32_t r = r0; r < r1; r += CS_CH) {
      uint32_t c[CS_CH], pc[CS_CH];
#pragma unroll
      for (uint32_t j = 0; j < CS_CH; j++) {
        c[j] = r + j < r1 ? hist[(size_t)(r + j) * nb + d] : 0u;
        pc[j] = (histp && r + j < r1) ? histp[(size_t)(r + j) * nb + d] : 0u;
      }
#pragma unroll
      for (uint32_t j = 0; j < CS_CH; j++) { sum += c[j]; psum += pc[j]; }
    }
  s_part[w][lane] = sum;
  s_pp[w][lane] = psum;
  __syncthreads();
  uint32_t run = 0, tot = 0, ptot = 0;
#pragma unroll
  for (int k = 0; k < ES_NW; k++) {
    const uint32_t c = s_part[k][lane];
    if (k < w) run += c;
    tot += c;
    ptot += s_pp[k][lane];
  }
  if (d < nb) {
    for (uint32_t r = r0; r < r1; r += CS_CH) {
      uint32_t c[CS_CH];
#pragma unroll
      for (uint32_t j = 0; j < CS_CH; j++) c[j] = r + j < r1 ? hist[(size_t)(r + j) * nb + d] : 0u;
#pragma unroll
      for (uint32_t j = 0; j < CS_CH; j++) {
        if (r + j < r1) hist[(size_t)(r + j) * nb + d] = run;
        run += c[j];
      }
    }
    if (w == 0) {
      dtotal[d] = tot;
      if (histp) ptotal[d] = ptot;
    }
  }
}

// =====================================================================================================================
// Tile schedule of the render launches
// =====================================================================================================================
// The render kernels run one wave per 8 x 8 tile and the dispatcher deals workgroup b to XCD b % 8. Rounds 1-3 gave every XCD
// a contiguous band of tile rows: equal tile COUNTS, and with them whatever imbalance the scene has — at the headline scene
// (nothing in the outer 5 % of the image) the two border XCDs hold 61 % of the others' pairs and idle for the last third of
// both render launches, and the launches end with full tiles still starting (wave traces: profiles/r04_wave_trace.txt).
// This body — ONE workgroup of 1024 threads, run as the EXTRA workgroup of the entry sort's scatter launch: it needs what the
// column scan before that launch left in bpairs, nothing of the scatter, and nobody needs it before block_lists_kernel, so it
// costs no launch of its own (a dependent one-workgroup launch is 6-9 us here) and no time on the stream — orders the
// 32 x 32-px blocks for the render launches:
//   * blocks with fewer than an eighth of the mean pair count are LIGHT (the empty rim of a scene); every XCD's sequence is its
//     share of the other blocks, then of the light ones: a launch ends on cheap tiles, not on full ones (without this class
//     every variant below loses 3-4 %);
//   * the other blocks, in row-major order, are cut into eight CONTIGUOUS runs — an XCD's blocks stay neighbours and its L2
//     keeps serving the records neighbouring tiles share — of equal WORK, where a block's work is its listed pairs up to a cap:
//     a tile stops blending once all its pixels are saturated, after about SCHED_K / (mean pair opacity) list entries, whatever
//     its list still holds. With opacities of 0.1 and below nothing saturates and the runs hold equal pairs; with trained
//     opacities every interior block is at the cap and the runs hold equal counts. (Equal pairs alone left two XCDs 15 % behind
//     at trained opacities; dealing the blocks round-robin in units of 1-8 balanced every regime but cost the saturating ones
//     2-3 % against contiguous runs — their footprints span 3 x 3 tiles and want their neighbours on the same XCD;
//     profiles/r04_sched_units.txt, profiles/r04_experiments/ab_sched_modes*.txt);
//   * the light blocks are dealt one by one;
//   * a sequence holds at most lg blocks (the render grid is fixed before this runs): the blocks beyond are handed, in order,
//     to the free places at the end of the other XCDs' sequences.
// Output: sched[x] = blocks in XCD x's sequence, where[b] = XCD << 24 | position of block b. block_lists_kernel then writes the
// descriptors of its block's 16 tiles (tile, list range) at that place of ImgWS::desc, so a render workgroup finds its tile AND
// its list with one 16-byte load. Results never depend on the schedule (a tile's wave does the same arithmetic wherever it runs).
// Five workgroup scans, no atomics (8 k same-address LDS atomics made the first version 10 us), no 64-bit division (the
// scatter launch that carries it is compiled for 64 VGPRs).
#define SCHED_T 1024
#define SCHED_C0 10u     // fixed cost of a tile's wave, in list entries
#ifndef SCHED_K
#define SCHED_K 60.0f    // list entries x mean pair opacity after which a tile counts as saturated. Bracketed by measurement: at
#endif                   // opacity 0.1 (555 entries per tile) runs of equal PAIRS win by 2.5 % -> K >= 55; at opacity 0.3 (680
                         // entries) runs of equal COUNT win by 4 % -> K <= 60 (profiles/r04_experiments/ab_sched_*.txt)
namespace {
__device__ inline uint4 u4_add(const uint4& a, const uint4& b) { return make_uint4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
// exclusive prefix over the workgroup of four 32-bit sums per thread; totals returned through `total`. All threads call.
template <int ST>
__device__ inline uint4 sched_scan(uint4 v, uint4* s_w, uint4& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint4 inc = make_uint4(wave_incl_scan_u32(v.x), wave_incl_scan_u32(v.y), wave_incl_scan_u32(v.z), wave_incl_scan_u32(v.w));
  __syncthreads();
  if (lane == 63) s_w[w] = inc;
  __syncthreads();
  uint4 base = make_uint4(0u, 0u, 0u, 0u), tot = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int k = 0; k < ST / 64; k++) {
    const uint4 c = s_w[k];
    if (k < w) base = u4_add(base, c);
    tot = u4_add(tot, c);
  }
  total = tot;
  return make_uint4(base.x + inc.x - v.x, base.y + inc.y - v.y, base.z + inc.z - v.z, base.w + inc.w - v.w);
}
// floor(8 a / b) for a < b, as the number of k in 1..7 with 8 a >= k b (b == 0: 0)
__device__ inline uint32_t eighth_of(uint32_t a, uint32_t b) {
  const unsigned long long a8 = 8ull * a;
  uint32_t x = 0;
#pragma unroll
  for (uint32_t k = 1; k < 8; k++) x += (b != 0u && a8 >= (unsigned long long)k * b) ? 1u : 0u;
  return x;
}

// ST threads (the workgroup that runs it: 1024, or 512 inside the 12-bit scatter launch), SCHED_MAX_BLOCKS / ST blocks per thread
template <int ST>
__device__ __forceinline__ void tile_sched_body(const uint32_t* __restrict__ bpairs, const uint32_t* __restrict__ bcount,
                                                const uint32_t* __restrict__ misc,
                                                uint32_t nblocks, uint32_t lg, uint32_t flags, uint32_t* __restrict__ sched,
                                                uint32_t* __restrict__ where) {
  constexpr int SCHED_ITEMS = (int)SCHED_MAX_BLOCKS / ST;
  __shared__ uint4 s_w[ST / 64];
  __shared__ uint32_t s_first[8], s_n0[8], s_nE[8], s_cnt[8], s_spill[8], s_free[9];
  const int t = threadIdx.x;
  // thread t holds blocks t * SCHED_ITEMS ... in block (row-major) order; the sums below fit 32 bits (pairs < 2^31, api.hip)
  uint32_t wk[SCHED_ITEMS];
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  {
    uint32_t ne[SCHED_ITEMS];
#pragma unroll
    for (int i = 0; i < SCHED_ITEMS; i++) {
      const uint32_t b = (uint32_t)t * SCHED_ITEMS + i;
      wk[i] = b < nblocks ? bpairs[b] : 0u;
      ne[i] = b < nblocks ? bcount[b] : 0u;
      v.x += wk[i];
      v.y += ne[i];
    }
    // Where every block's entries and pairs start (exclusive prefixes of the per-block counts, in the second and third third of
    // `where`): block_lists_kernel's 1024 ... 4096 workgroups each summed the counts of all the blocks before their own —
    // a global round trip and two workgroup reductions at the head of a chain of dependent steps (tools/bl_phases.py: 8 % of a
    // workgroup's time at 1024 blocks, 18 % at 4096) — for what this scan has as a by-product.
    uint4 tot0;
    uint4 pre = sched_scan<ST>(v, s_w, tot0);
#pragma unroll
    for (int i = 0; i < SCHED_ITEMS; i++) {
      const uint32_t b = (uint32_t)t * SCHED_ITEMS + i;
      if (b < nblocks) *reinterpret_cast<uint2*>(where + SCHED_MAX_BLOCKS + 2u * b) = make_uint2(pre.y, pre.x);  // {entries, pairs} before b
      pre.x += wk[i];
      pre.y += ne[i];
    }
    v.y = 0u;
  }
  uint4 tot;
  (void)sched_scan<ST>(v, s_w, tot);
  const unsigned long long total_pairs = tot.x;
  // pairs of a block beyond which its tiles have saturated: 16 tiles x SCHED_K / (mean pair opacity), the mean pair opacity
  // being sum(round(64 opacity)) / (64 pairs) over the listed pairs (misc[MISC_OPW], written by the count scan)
  const float opw = (float)misc[MISC_OPW_LO] + 4294967296.0f * (float)misc[MISC_OPW_HI];
  const float capf = (opw > 0.f && !(flags & 0x20u)) ? 16.0f * 64.0f * SCHED_K * (float)tot.x / opw : 4.0e9f;
  const uint32_t cap = capf < 4.0e9f ? (uint32_t)capf : 0xFFFFFFFFu;
  // raw[] keeps the pairs; wk[] becomes the block's WORK: its pairs up to the cap
  uint32_t raw[SCHED_ITEMS];
  v = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int i = 0; i < SCHED_ITEMS; i++) {
    raw[i] = wk[i];
    wk[i] = wk[i] < cap ? wk[i] : cap;
    v.x += wk[i];
  }
  (void)sched_scan<ST>(v, s_w, tot);
  const unsigned long long total_work = tot.x;
  (void)total_pairs;
  // light: less than an eighth of the mean work (compared as products). By WORK, not pairs: where tiles saturate early a rim
  // block that blends all of its few pairs costs as much as an interior one and must not wait for the end of the launch
  bool light[SCHED_ITEMS];
  v = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int i = 0; i < SCHED_ITEMS; i++) {
    const uint32_t b = (uint32_t)t * SCHED_ITEMS + i;
    light[i] = 8ull * wk[i] * nblocks < total_work;
    if (b < nblocks && !light[i]) { v.x += raw[i]; v.y++; }
  }
  (void)sched_scan<ST>(v, s_w, tot);
  // EDGE blocks (saturating regimes only: the mean block is beyond the cap): a block that lists clearly fewer pairs than its
  // peers — under three quarters of their mean — but more than the cap is a block the scene covers only partly: the pixels the
  // scene does not reach never saturate, its tiles walk their whole lists, and they are the launch's stragglers (wave traces
  // at trained opacities: tiles of 90-130 us among a mean of 37, profiles/r04_wave_trace.txt). They start FIRST, dealt one by one.
  const unsigned long long heavy_pairs = tot.x, heavy_n = tot.y;
  const bool saturating = !(flags & 0x40u) && (unsigned long long)cap * heavy_n < heavy_pairs;
  bool edge[SCHED_ITEMS];
  v = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int i = 0; i < SCHED_ITEMS; i++) {
    const uint32_t b = (uint32_t)t * SCHED_ITEMS + i;
    edge[i] = saturating && !light[i] && raw[i] > cap && 4ull * raw[i] * heavy_n < 3ull * heavy_pairs;
    wk[i] += 16u * SCHED_C0;
    if (b < nblocks) {
      if (light[i]) v.w++;
      else if (edge[i]) v.z++;
      else { v.x += wk[i]; v.y++; }
    }
  }
  uint4 run = sched_scan<ST>(v, s_w, tot);  // before this thread's blocks: {work, count} of the run blocks, edge blocks, light blocks
  const uint32_t nE = tot.z, n1 = tot.w;
  uint32_t xcd[SCHED_ITEMS], crank[SCHED_ITEMS];
  uint4 c0x = make_uint4(0u, 0u, 0u, 0u);  // run blocks per XCD, 16 bits each: x = {xcd 0, 1}, y = {2, 3}, z = {4, 5}, w = {6, 7}
#pragma unroll
  for (int i = 0; i < SCHED_ITEMS; i++) {
    const uint32_t b = (uint32_t)t * SCHED_ITEMS + i;
    xcd[i] = 0; crank[i] = 0;
    if (b >= nblocks) continue;
    if (light[i]) {
      xcd[i] = run.w % 8u;
      crank[i] = run.w++ / 8u;
    } else if (edge[i]) {
      xcd[i] = run.z % 8u;
      crank[i] = run.z++ / 8u;
    } else {
      xcd[i] = eighth_of(run.x + wk[i] / 2u, tot.x);  // by the block's centre of work
      crank[i] = run.y;
      run.x += wk[i]; run.y++;
      const uint32_t one = 1u << (16u * (xcd[i] & 1u));
      c0x.x += (xcd[i] >> 1) == 0u ? one : 0u; c0x.y += (xcd[i] >> 1) == 1u ? one : 0u;
      c0x.z += (xcd[i] >> 1) == 2u ? one : 0u; c0x.w += (xcd[i] >> 1) == 3u ? one : 0u;
    }
  }
  uint4 c0tot;
  (void)sched_scan<ST>(c0x, s_w, c0tot);  // (at most 4096 blocks: a 16-bit field cannot overflow into its neighbour)
  if (t == 0) {
    uint32_t first0 = 0, a = 0, f = 0;
    for (uint32_t x = 0; x < 8; x++) {
      const uint32_t pair = x >> 1 == 0 ? c0tot.x : (x >> 1 == 1 ? c0tot.y : (x >> 1 == 2 ? c0tot.z : c0tot.w));
      const uint32_t n0x = (pair >> (16u * (x & 1u))) & 0xFFFFu;
      const uint32_t nEx = nE / 8u + (x < nE % 8u ? 1u : 0u);
      s_first[x] = first0;  // (the XCD rises with the rank among the run blocks: a run is contiguous in rank)
      s_nE[x] = nEx;
      s_n0[x] = n0x;
      first0 += n0x;
      const uint32_t c = nEx + n0x + n1 / 8u + (x < n1 % 8u ? 1u : 0u);
      s_cnt[x] = c < lg ? c : lg;  // the XCD's own blocks that stay
      s_spill[x] = a;              // spilled blocks of the XCDs before x
      s_free[x] = f;               // free places of the XCDs before x
      a += c > lg ? c - lg : 0u;
      f += c < lg ? lg - c : 0u;
    }
    s_free[8] = f;
    for (uint32_t x = 0; x < 8; x++) {  // final length of XCD x's sequence: its own blocks + the spilled ones that land in its free places
      const uint32_t fr = s_free[x + 1] - s_free[x];
      const uint32_t take = a > s_free[x] ? (a - s_free[x] < fr ? a - s_free[x] : fr) : 0u;
      sched[x] = s_cnt[x] + take;
      sched[8 + x] = 0u;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < SCHED_ITEMS; i++) {
    const uint32_t b = (uint32_t)t * SCHED_ITEMS + i;
    if (b >= nblocks) continue;
    const uint32_t x = xcd[i];
    // an XCD's sequence: its edge blocks, its run, its light blocks
    uint32_t pos = light[i] ? s_nE[x] + s_n0[x] + crank[i] : (edge[i] ? crank[i] : s_nE[x] + crank[i] - s_first[x]);
    uint32_t dst = x;
    if (pos >= lg) {  // spilled: the r-th spilled block overall takes the r-th free place (8 lg >= nblocks: there is one)
      const uint32_t r = s_spill[x] + (pos - lg);
      uint32_t y = 0;
      for (int k = 1; k < 8; k++) y += (r >= s_free[k]) ? 1u : 0u;
      dst = y;
      pos = s_cnt[y] + (r - s_free[y]);
    }
    where[b] = (dst << 24) | pos;
  }
}
}  // namespace

// kernel 3: stable scatter. Ranking as in radix_scatter_body (wave-private ballot match, NBITS ballots per 64 entries,
// 16-bit wave counters: a wave holds 512 entries). The workgroup's entries are then brought into sorted order through an
// LDS window, ES_WIN entries per round, and written so that consecutive lanes write consecutive addresses inside each
// digit's run (entries stored straight from their ranking lanes reached HBM as partial lines: 2x the bytes written,
// 26 us against the 6 us the histogram takes to read the same data).
#define ES_WIN 2048
// T_ threads per workgroup (ES_TILE / T_ entries per thread): 1024 up to 2048 digits, 512 for 4096 (the wave counters are
// NW x digits x 2 bytes of LDS).
// Up to 1024 digits two workgroups fit a CU's LDS (2 x 74 KB): compiled for 64 VGPRs (4 spilled at 10 bits) so that they
// also fit its registers — the launch has more workgroups than CUs from 2.1 M entries on (1 M Gaussians at trained
// opacities: 325; scatter 34 -> 30 us, 2 M: -9 us), and is unchanged below (208 workgroups at opacity 0.01).
#ifndef ES_WAVES
#define ES_WAVES 8
#endif
template <int NBITS, int T_>
__global__ __launch_bounds__(T_, (NBITS <= 10 && T_ == 1024 ? ES_WAVES : 1)) void entry_scatter_kernel(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                             const uint32_t* __restrict__ misc, uint32_t cap, int shift,
                                                             const uint32_t* __restrict__ hist,
                                                             const uint32_t* __restrict__ dtotal,
                                                             const uint32_t* __restrict__ bpairs, uint32_t sched_blocks,
                                                             uint32_t sched_lg, uint32_t* __restrict__ sched,
                                                             uint32_t* __restrict__ where) {
  constexpr int ES_ITEMS_ = ES_TILE / T_, ES_NW_ = T_ / 64;
  constexpr uint32_t nb = 1u << NBITS, mask = nb - 1u;
  if constexpr (T_ == 1024 || T_ == 512) {
    if (sched_blocks && blockIdx.x == gridDim.x - 1) {  // the launch's extra workgroup: the tile schedule
      tile_sched_body<T_>(bpairs, dtotal, misc, sched_blocks, sched_lg & 0xFFFFFFu, sched_lg >> 24, sched, where);  // (one pass: dtotal IS the per-block entry count)
      return;
    }
  }
  const uint32_t n = entries_on_device(misc, cap);
  const uint32_t tile0 = blockIdx.x * (uint32_t)ES_TILE;
  if (tile0 >= n) return;
  __shared__ uint16_t s_wcnt[ES_NW_][nb < 2u ? 2u : nb];  // per-wave digit counts -> per-wave offset inside the digit
  __shared__ uint32_t s_gbase[nb];   // global output position of this workgroup's first entry of each digit
  __shared__ uint32_t s_dstart[nb];  // start of each digit in the workgroup's sorted order
  __shared__ uint32_t s_w[ES_NW_], s_w2[ES_NW_];
  __shared__ uint4 s_win[ES_WIN];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (uint32_t k = t; k < (uint32_t)ES_NW_ * (nb < 2u ? 2u : nb) / 2u; k += T_) reinterpret_cast<uint32_t*>(&s_wcnt[0][0])[k] = 0u;
  __syncthreads();
  const uint32_t base = tile0 + (uint32_t)w * (64 * ES_ITEMS_);
  uint4 item[ES_ITEMS_];
  uint32_t lrank[ES_ITEMS_];
#pragma unroll
  for (int i = 0; i < ES_ITEMS_; i++) {
    const uint32_t k = base + i * 64 + lane;
    item[i] = make_uint4(0u, 0u, 0u, 0u);
    if (k < n) item[i] = in[k];
  }
#pragma unroll
  for (int i = 0; i < ES_ITEMS_; i++) {
    lrank[i] = 0;
    if (base + i * 64 >= n) continue;  // (wave-uniform)
    const bool live = base + i * 64 + lane < n;
    const uint32_t d = (item[i].x >> shift) & mask;
    uint32_t count;
    const uint32_t r = match_rank<NBITS>(d, live, count);
    uint32_t before = 0;
    if (live) before = s_wcnt[w][d];  // this wave's earlier groups (LDS ops of one wave execute in order)
    __builtin_amdgcn_wave_barrier();
    if (live && r == 0) s_wcnt[w][d] = (uint16_t)(before + count);  // one leader per digit
    __builtin_amdgcn_wave_barrier();
    lrank[i] = before + r;
  }
  __syncthreads();
  // per digit: offsets of the waves inside the workgroup's run; exclusive scans over the digits of the global totals
  // (-> global base) and of the workgroup's totals (-> place in the workgroup's sorted order).
  // Digits are dealt to threads as d = t, t + T_, ...
  uint32_t carry = 0, carry2 = 0;
  for (uint32_t d0 = 0; d0 < nb; d0 += T_) {
    const uint32_t d = d0 + t;
    const uint32_t v = d < nb ? dtotal[d] : 0u;
    uint32_t run = 0;
    if (d < nb) {
#pragma unroll
      for (int k = 0; k < ES_NW_; k++) {
        const uint32_t c = s_wcnt[k][d];
        s_wcnt[k][d] = (uint16_t)run;
        run += c;
      }
    }
    const uint32_t inc = wave_incl_scan_u32(v), inc2 = wave_incl_scan_u32(run);
    if (lane == 63) { s_w[w] = inc; s_w2[w] = inc2; }
    __syncthreads();
    uint32_t pre = 0, tot = 0, pre2 = 0, tot2 = 0;
#pragma unroll
    for (int k = 0; k < ES_NW_; k++) {
      if (k < w) { pre += s_w[k]; pre2 += s_w2[k]; }
      tot += s_w[k];
      tot2 += s_w2[k];
    }
    if (d < nb) {
      s_gbase[d] = carry + pre + inc - v + hist[(size_t)blockIdx.x * nb + d];
      s_dstart[d] = carry2 + pre2 + inc2 - run;
    }
    carry += tot;
    carry2 += tot2;
    __syncthreads();
  }
  uint32_t lpos[ES_ITEMS_];  // place in the workgroup's sorted order
#pragma unroll
  for (int i = 0; i < ES_ITEMS_; i++) {
    const uint32_t d = (item[i].x >> shift) & mask;
    lpos[i] = base + i * 64 + lane < n ? s_dstart[d] + s_wcnt[w][d] + lrank[i] : 0xFFFFFFFFu;
  }
  const uint32_t nvalid = n - tile0 < (uint32_t)ES_TILE ? n - tile0 : (uint32_t)ES_TILE;
  for (uint32_t w0 = 0; w0 < nvalid; w0 += ES_WIN) {
#pragma unroll
    for (int i = 0; i < ES_ITEMS_; i++)
      if (lpos[i] - w0 < (uint32_t)ES_WIN) s_win[lpos[i] - w0] = item[i];  // (wraps below the window: fails the test)
    __syncthreads();
    const uint32_t nwin = nvalid - w0 < (uint32_t)ES_WIN ? nvalid - w0 : (uint32_t)ES_WIN;
    for (uint32_t idx = t; idx < nwin; idx += T_) {
      const uint4 e = s_win[idx];
      const uint32_t d = (e.x >> shift) & mask;
      out[s_gbase[d] + (w0 + idx - s_dstart[d])] = e;
    }
    __syncthreads();
  }
}

// ---- two-pass sort only: block ranges (identifyTileRanges, rasterizer_impl.cu:116-138, at block granularity) and the
//      number of listed (internal tile, Gaussian) pairs per block, from the block-sorted entries. One wave takes 64
//      consecutive entries: equal block ids are adjacent, so a wave adds its pair count with one atomic per block it meets
//      (integer adds: order-independent). (A second word per 256 blocks, added to by every wave, was tried first: all waves
//      of the launch then queue on four addresses, 314 us.) Written as per-block COUNTS, like the single-pass sort's totals. ----
__global__ __launch_bounds__(BLK) void block_counts_kernel(const uint4* __restrict__ ent, const uint32_t* __restrict__ misc,
                                                           uint32_t cap, uint32_t* __restrict__ dtotal,
                                                           uint32_t* __restrict__ ptotal) {
  const uint32_t n = entries_on_device(misc, cap);
  const uint32_t i = blockIdx.x * BLK + threadIdx.x;
  if ((blockIdx.x * BLK) >= n) return;
  const bool livel = i < n;
  const uint32_t key = livel ? ent[i].x : 0u;
  const uint32_t blk = key & ((1u << MACRO_KEY_BITS) - 1u);
  const uint32_t pc = livel ? (uint32_t)__popc(key >> MACRO_KEY_BITS) : 0u;  // 1..16 listed internal tiles
  unsigned long long rem = __ballot(livel);
  while (rem) {
    const int first = __builtin_ctzll(rem);
    const uint32_t b0 = __shfl(blk, first, 64);
    const unsigned long long m = __ballot(livel && blk == b0) & rem;
    uint32_t sum = 0;
#pragma unroll
    for (int bit = 0; bit < 5; bit++) sum += (uint32_t)__popcll(__ballot((pc >> bit) & 1u) & m) << bit;
    if ((threadIdx.x & 63) == first) {
      atomicAdd(&dtotal[b0], (uint32_t)__popcll(m));
      atomicAdd(&ptotal[b0], sum);
    }
    rem &= ~m;
  }
}
__global__ __launch_bounds__(BLK) void clear_counts_kernel(uint32_t* __restrict__ a, uint32_t* __restrict__ b, uint32_t n) {
  const uint32_t i = blockIdx.x * BLK + threadIdx.x;
  if (i < n) { a[i] = 0u; b[i] = 0u; }
}

// the tile schedule rides in the scatter launch when that has 1024-thread workgroups (up to 2048 blocks); with 512-thread
// ones (4096 blocks: 2048^2 images) it is a one-workgroup launch of its own
struct SchedArgs {
  const uint32_t* bpairs;
  uint32_t blocks, lg;  // blocks == 0: no schedule
  uint32_t* sched;
  uint32_t* where;
};
__global__ __launch_bounds__(SCHED_T) void tile_sched_kernel(const uint32_t* __restrict__ bpairs, const uint32_t* __restrict__ bcount,
                                                             const uint32_t* __restrict__ misc,
                                                             uint32_t nblocks, uint32_t lg, uint32_t* __restrict__ sched,
                                                             uint32_t* __restrict__ where) {
  tile_sched_body<SCHED_T>(bpairs, bcount, misc, nblocks, lg & 0xFFFFFFu, lg >> 24, sched, where);
}
template <int NBITS>
static void launch_entry_scatter_n(uint32_t nblk, hipStream_t s, const uint4* in, uint4* out, const uint32_t* misc, uint32_t cap,
                                   int shift, const uint32_t* hist, const uint32_t* dtotal, const SchedArgs& sa) {
  constexpr int T_ = NBITS <= 11 ? 1024 : 512;
  const bool ride = sa.blocks != 0u && (T_ == 1024 || T_ == 512);  // (every variant today: the stand-alone launch is the fallback)
  if (sa.blocks != 0u && !ride)
    hipLaunchKernelGGL(tile_sched_kernel, dim3(1), dim3(SCHED_T), 0, s, sa.bpairs, dtotal, misc, sa.blocks, sa.lg, sa.sched, sa.where);
  hipLaunchKernelGGL((entry_scatter_kernel<NBITS, T_>), dim3(nblk + (ride ? 1u : 0u)), dim3(T_), 0, s, in, out, misc, cap, shift,
                     hist, dtotal, sa.bpairs, ride ? sa.blocks : 0u, sa.lg, sa.sched, sa.where);
}
static void launch_entry_scatter(int bits, uint32_t nblk, hipStream_t s, const uint4* in, uint4* out, const uint32_t* misc,
                                 uint32_t cap, int shift, const uint32_t* hist, const uint32_t* dtotal, const SchedArgs& sa) {
  switch (bits) {  // the ballot loop of the ranking is unrolled for the digit width
#define ES_CASE(N) case N: launch_entry_scatter_n<N>(nblk, s, in, out, misc, cap, shift, hist, dtotal, sa); break;
    ES_CASE(1) ES_CASE(2) ES_CASE(3) ES_CASE(4) ES_CASE(5) ES_CASE(6) ES_CASE(7) ES_CASE(8) ES_CASE(9) ES_CASE(10) ES_CASE(11)
    default: launch_entry_scatter_n<12>(nblk, s, in, out, misc, cap, shift, hist, dtotal, sa); break;
#undef ES_CASE
  }
}

void launch_entry_sort(const GeomWS& g, const SortWS& w, int P, int H, int W, hipStream_t s) {
  const uint32_t gmx = macro_grid_x(W, BLOCK_BIG), nblocks = gmx * macro_grid_y(H, BLOCK_BIG);
  hipLaunchKernelGGL(expand_entries_kernel<BLOCK_BIG>, dim3(g.nblkE), dim3(BLK), 0, s, g.binfo, g.bext, g.pblock, g.pblockE,
                     g.misc, w.cap, (uint32_t)P, gmx, w.entA);
  int passes, bits;
  block_sort_geometry(H, W, passes, bits);
  const uint4* in = w.entA;
  uint4* out = w.entB;
  for (int pass = 0; pass < passes; pass++) {
    const int shift = pass * bits;
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t* histp = passes == 1 ? w.histp : nullptr;
    hipLaunchKernelGGL(entry_hist_kernel, dim3(w.nblk), dim3(ES_T), 0, s, in, g.misc, w.cap, shift, mask, w.hist, histp);
    hipLaunchKernelGGL(entry_colscan_kernel, dim3((mask + 64u) / 64u), dim3(ES_T), 0, s, w.hist, histp, g.misc, w.cap, mask + 1u,
                       passes == 1 ? g.bcount : w.dtotal, g.bpairs);
    // (one pass <=> at most SCHED_MAX_BLOCKS blocks) the render launches' tile schedule goes with this launch
    // (lg travels with the experiment switches in its top byte)
    const SchedArgs sa{g.bpairs, (passes == 1 && sched_enabled()) ? nblocks : 0u, sched_capacity(nblocks) | (sched_flags() << 24), g.sched, g.where};
    launch_entry_scatter(bits, w.nblk, s, in, out, g.misc, w.cap, shift, w.hist, passes == 1 ? g.bcount : w.dtotal, sa);
    const uint4* t = in; in = out; out = const_cast<uint4*>(t);
  }
  if (passes > 1) {
    hipLaunchKernelGGL(clear_counts_kernel, dim3(ceil_div_u32(nblocks, BLK)), dim3(BLK), 0, s, g.bcount, g.bpairs, nblocks);
    hipLaunchKernelGGL(block_counts_kernel, dim3(ceil_div_u32(w.cap, BLK)), dim3(BLK), 0, s, in, g.misc, w.cap, g.bcount, g.bpairs);
  }
}

// =====================================================================================================================
// Per block: depth order + split into the internal tiles' lists
// =====================================================================================================================
// threads per workgroup: 1024 (16 wave64), or — chosen per forward by launch_block_lists — 512 / 256 where blocks hold few entries

namespace {
template <int BL_NW>
__device__ inline uint32_t bl_sum(uint32_t v, uint32_t* s_red) {  // sum over the workgroup (all threads call)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  uint32_t tot = 0;
#pragma unroll
  for (int k = 0; k < BL_NW; k++) tot += s_red[k];
  return tot;
}
}  // namespace

// One workgroup per 32 x 32-pixel block. Its n entries sit in ent[s0, s0+n) in Gaussian-id order (s0, n and the pairs
// listed before the block: sums of the per-block counts the sort produced). The kernel orders them by depth key with
// `npass` stable LSD radix passes of 8 bits (wave-private ballot ranking as in radix_scatter_body) over {depth key, index}
// pairs, then
//   MODE 1: walks the ordered entries and appends, per internal tile of the block, {Gaussian id, record slot} for every
//       entry whose sub-mask lists the tile (per tile: ballot + prefix count, running offsets across waves and chunks in
//       LDS) at that tile's place in point_list — tile lists of a block are adjacent, blocks follow each other in block
//       order — and writes the 16 tile ranges and clears the block's share of the backward's live flags;
//   MODE BLOCK_BIG: writes the ordered entries' {key, {Gaussian id, first slot}} for the block-list render kernels.
// Up to BL_CH = 1024 x BL_ITEMS entries (the usual case: a block holds a few thousand) the pairs live in registers and move through LDS
// between passes, the sub-mask travelling in the index word's upper half. Longer lists stream chunk by chunk through the
// scratch ping-pong buffer ki (the entry buffer the block sort left free: 16 bytes per entry = two 8-byte pairs) with
// running digit bases in LDS, the next digit's histogram taken while scattering; a block's data stays in its CU's L2.
// Any n is handled (a block with a million entries only takes long).
// Compiled for 64 VGPRs (eight waves per SIMD): two 1024-thread workgroups per CU, the launch is one workgroup per block.
// BL_T = 1024 threads where a block holds 1400 entries or more on average (1 M Gaussians at 1024^2: 1600 ... 2600), 512 / 256
// below that (large images, small scenes: 4096 blocks of ~600 entries at 2048^2 — sixteen waves then wait at barriers for two
// busy ones and the CU holds two such workgroups: 134 -> 72 us there; launch_block_lists, profiles/r04_experiments/ab_bl_threads.txt).
// BL_ITEMS = 4 (LDS path up to 4096 entries, no spills) unless the average block holds 2800 ... 6000 entries: then 8
// (8192 entries, a few spilled registers: 2 M Gaussians at trained opacities 218 -> 158 us; at 1700 entries per block the
// 4-item build is faster, 59 against 77 us, and beyond 6000 the 8-item build loses to the 4-item streaming path).
#define BL_NARROW_256 750.0   // entries per block (average) up to which a block's workgroup has 256 threads ...
#define BL_NARROW_512 1400.0  // ... 512 threads; above: 1024 (measured crossovers: profiles/r04_experiments/ab_bl_threads.txt)
// -DEOGS_BL_PHASES: where a block's workgroup spends its time (wave 0's clock at the phase boundaries, summed over the workgroups
// into g_bl_phase[]; read with eogs_debug_bl_phases, tools/bl_phases.py). Diagnostics only.
#ifdef EOGS_BL_PHASES
__device__ unsigned long long g_bl_phase[8];
#define BLP_DECL unsigned long long blp_t = __builtin_readcyclecounter()
#define BLP(i) do { const unsigned long long blp_n = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g_bl_phase[i], blp_n - blp_t); blp_t = blp_n; } while (0)
#else
#define BLP_DECL
#define BLP(i)
#endif
template <int MODE, int BL_ITEMS, int BL_T>
__global__ __launch_bounds__(BL_T, 8) void block_lists_kernel(const uint4* __restrict__ ent, uint2* ki,
                                                              const uint32_t* __restrict__ bcount,
                                                              const uint32_t* __restrict__ bpairs,
                                                              const uint32_t* __restrict__ misc, uint32_t gmx, uint32_t gsx,
                                                              uint32_t gsy, uint2* __restrict__ point_list,
                                                              uint32_t* __restrict__ sorted_keys, uint2* __restrict__ ranges,
                                                              uint8_t* __restrict__ live, uint32_t cap_slots,
                                                              uint32_t cap_entries, const uint32_t* __restrict__ where,
                                                              uint4* __restrict__ desc, uint32_t lg16) {
  constexpr int BL_NW = BL_T / 64;
  constexpr int BL_CH = BL_T * BL_ITEMS;  // entries per chunk of a pass = longest list the LDS path takes
  __shared__ uint16_t s_wcnt[BL_NW][256];
  __shared__ uint32_t s_base[256], s_cb[256], s_h[256], s_hn[256];
  __shared__ uint32_t s_red[BL_NW];
  __shared__ uint32_t s_tc[16], s_tb[16], s_cw[BL_NW][16];
  __shared__ uint2 s_stage[BL_CH];
  // {Gaussian id, first record slot} of the block's entries by their position in the entry buffer, kept from the one coalesced
  // load of the entries: the split into tile lists then finds them in LDS instead of gathering 16 bytes per ordered entry from
  // global memory — a dependent round trip of ~5 us per workgroup under the launch's own load (tools/bl_phases.py: the split was
  // 40 % of a workgroup's 32 us). Per-tile lists with four items per thread only: with eight the two arrays would leave one
  // workgroup per CU; and 1024-thread workgroups only: the smaller ones (few entries per block) measured no gain and would
  // lose a resident workgroup to the extra LDS.
  constexpr bool IDS = MODE == 1 && BL_ITEMS == 4 && BL_T == 1024;
  __shared__ uint2 s_ids[IDS ? BL_CH : 1];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const uint32_t b = blockIdx.x;
  const int npass = (int)misc[MISC_DEPTH_PASSES];
  BLP_DECL;
  // The render workgroup that takes internal tile `sub` of this block finds tile and list in ONE 16-byte descriptor at the
  // block's place in the tile schedule (tile_sched_body): XCD x's sequence starts at desc[x * lg16], 16 descriptors per block.
  auto put_desc = [&](uint32_t sub, uint32_t tx, uint32_t ty, uint32_t begin, uint32_t end) {
    if (desc == nullptr) return;
    const uint32_t wh = where[b];
    desc[(size_t)(wh >> 24) * lg16 + (size_t)(wh & 0xFFFFFFu) * 16u + sub] =
        make_uint4((tx < gsx && ty < gsy) ? (tx | (ty << 16)) : 0xFFFFFFFFu, begin, end, 0u);
  };
  {
    // The workspaces may have been sized before the host knew this forward's counts (EOGS_FLAG_DEFER_COUNTS): a forward
    // that does not fit them (or lists nothing: the entry sort then left the per-block counts alone) gets empty lists —
    // every later kernel reads lists through `ranges` only — and the host, which learns the counts a moment later, repeats
    // the call with workspaces that fit.
    const uint32_t ne = misc[MISC_MACRO_LO];
    const bool fits = misc[MISC_TOTAL_HI] == 0u && misc[MISC_MACRO_HI] == 0u && misc[MISC_TOTAL_LO] <= cap_slots && ne <= cap_entries;
    if (!fits || ne == 0u) {  // (uniform over the grid)
      if (MODE != 1 && t == 0) ranges[b] = make_uint2(0u, 0u);
      if (t < 16) {
        const uint32_t tx = (b % gmx) * BLOCK_BIG + (uint32_t)(t % BLOCK_BIG), ty = (b / gmx) * BLOCK_BIG + (uint32_t)(t / BLOCK_BIG);
        if (MODE == 1 && tx < gsx && ty < gsy) ranges[ty * gsx + tx] = make_uint2(0u, 0u);
        put_desc((uint32_t)t, tx, ty, 0u, 0u);
      }
      return;
    }
  }

  // where this block's entries and pairs start: read from the prefixes the tile schedule's workgroup left behind `where`
  // (tile_sched_body; desc != nullptr <=> that workgroup ran), else sums of the counts of the blocks before it
  uint32_t s0, pairs_before = 0;
  const uint32_t n = bcount[b];
  if (desc != nullptr) {
    const uint2 pf = *reinterpret_cast<const uint2*>(where + SCHED_MAX_BLOCKS + 2u * b);
    s0 = pf.x;
    pairs_before = pf.y;
  } else {
    uint32_t ve = 0, vp = 0;
    for (uint32_t i = t; i < b; i += BL_T) {
      ve += bcount[i];
      if (MODE == 1) vp += bpairs[i];
    }
    s0 = bl_sum<BL_NW>(ve, s_red);
    if (MODE == 1) pairs_before = bl_sum<BL_NW>(vp, s_red);
  }
  if (MODE != 1 && t == 0) ranges[b] = make_uint2(s0, s0 + n);
  if (MODE != 1 && t < 16)  // block lists: every tile of the block reads the block's list
    put_desc((uint32_t)t, (b % gmx) * BLOCK_BIG + (uint32_t)(t % BLOCK_BIG), (b / gmx) * BLOCK_BIG + (uint32_t)(t / BLOCK_BIG), s0, s0 + n);
  if (t < 256) s_h[t] = 0;
  if (t < 16) s_tc[t] = 0;
  __syncthreads();
  BLP(0);  // prologue: counts fit, where the block's entries and pairs start

  const bool fast = n <= (uint32_t)BL_CH;
  uint2* src = ki + 2 * (size_t)s0;
  uint2* dst = src + n;
  // A chunk's 64-entry groups are dealt to the waves as equal contiguous runs (`per` groups each: a short list keeps every
  // wave busy instead of filling the first waves' groups).
  const uint32_t per0 = (((n < (uint32_t)BL_CH ? n : (uint32_t)BL_CH) + 63u) / 64u + BL_NW - 1u) / BL_NW;
  const uint32_t wb0 = (uint32_t)w * per0 * 64u;

  if (fast) {
    // ---- LDS path: {depth key, sub-mask << 16 | index} in registers ----
    uint32_t dk[BL_ITEMS], px[BL_ITEMS];
#pragma unroll
    for (int i = 0; i < BL_ITEMS; i++) {
      dk[i] = 0xFFFFFFFFu; px[i] = 0;
      const uint32_t k = wb0 + i * 64 + lane;
      if ((uint32_t)i < per0 && k < n) {
        if (IDS) {
          const uint4 e = ent[s0 + k];  // {key, depth key, Gaussian id, first slot}
          dk[i] = e.y; px[i] = (e.x & 0xFFFF0000u) | k;
          s_ids[k] = make_uint2(e.z, e.w);
        } else {
          const uint2 kd = *reinterpret_cast<const uint2*>(ent + s0 + k);  // {key, depth key}
          dk[i] = kd.y; px[i] = (kd.x & 0xFFFF0000u) | k;
        }
      }
    }
    BLP(1);  // entries loaded (issued)
    for (int p = 0; p < npass; p++) {
      const int sh = 8 * p;
      for (int k = t; k < BL_NW * 128; k += BL_T) reinterpret_cast<uint32_t*>(&s_wcnt[0][0])[k] = 0u;
      __syncthreads();  // (the previous pass has read its pairs back from s_stage)
      uint32_t lr[BL_ITEMS];
#pragma unroll
      for (int i = 0; i < BL_ITEMS; i++) {
        lr[i] = 0;
        if ((uint32_t)i >= per0 || wb0 + i * 64 >= n) continue;  // (wave-uniform)
        const bool livel = wb0 + i * 64 + lane < n;
        const uint32_t d = (dk[i] >> sh) & 255u;
        uint32_t count;
        const uint32_t r = match_rank<8>(d, livel, count);
        uint32_t before = 0;
        if (livel) before = s_wcnt[w][d];
        __builtin_amdgcn_wave_barrier();
        if (livel && r == 0) s_wcnt[w][d] = (uint16_t)(before + count);
        __builtin_amdgcn_wave_barrier();
        lr[i] = before + r;
      }
      __syncthreads();
      {  // digit t (threads 256.. carry zeros): offsets of the waves inside the digit, exclusive scan of the digit totals
        uint32_t tot = 0;
        if (t < 256) {
#pragma unroll
          for (int k = 0; k < BL_NW; k++) {
            const uint32_t c = s_wcnt[k][t];
            s_wcnt[k][t] = (uint16_t)tot;
            tot += c;
          }
        }
        const uint32_t inc = wave_incl_scan_u32(tot);
        if (lane == 63) s_red[w] = inc;
        __syncthreads();
        uint32_t pre = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (k < w) pre += s_red[k];
        if (t < 256) s_cb[t] = pre + inc - tot;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < BL_ITEMS; i++) {
        if ((uint32_t)i < per0 && wb0 + i * 64 + lane < n) {
          const uint32_t d = (dk[i] >> sh) & 255u;
          s_stage[s_cb[d] + s_wcnt[w][d] + lr[i]] = make_uint2(dk[i], px[i]);
        }
      }
      __syncthreads();
      if (p + 1 < npass) {
#pragma unroll
        for (int i = 0; i < BL_ITEMS; i++) {
          if ((uint32_t)i < per0 && wb0 + i * 64 + lane < n) {
            const uint2 e = s_stage[wb0 + i * 64 + lane];
            dk[i] = e.x; px[i] = e.y;
          }
        }
      }
    }
    if (npass == 0) {  // all keys equal: id order
#pragma unroll
      for (int i = 0; i < BL_ITEMS; i++)
        if ((uint32_t)i < per0 && wb0 + i * 64 + lane < n) s_stage[wb0 + i * 64 + lane] = make_uint2(dk[i], px[i]);
    }
  } else {
    // ---- streaming path: {depth key, index} pairs in the global ping-pong buffer ----
    {
      uint32_t cnt[16];
#pragma unroll
      for (int j = 0; j < 16; j++) cnt[j] = 0;
      for (uint32_t i0 = 0; i0 < n; i0 += BL_T) {
        const uint32_t i = i0 + t;
        uint32_t sub = 0;
        if (i < n) {
          const uint2 kd = *reinterpret_cast<const uint2*>(ent + s0 + i);
          src[i] = make_uint2(kd.y, i);
          if (npass > 0) atomicAdd(&s_h[kd.y & 255u], 1u);
          sub = kd.x >> MACRO_KEY_BITS;
        }
        if (MODE == 1) {
#pragma unroll
          for (int j = 0; j < 16; j++) cnt[j] += (uint32_t)__popcll(__ballot((sub >> j) & 1u));
        }
      }
      if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 16; j++)
          if (lane == j && cnt[j]) atomicAdd(&s_tc[j], cnt[j]);
      }
    }
    for (int p = 0; p < npass; p++) {
      const int sh = 8 * p;
      __syncthreads();  // s_h complete; the pairs written so far are visible to the whole workgroup
      {
        // exclusive scan of the 256 digit counts (threads 256.. carry zeros)
        const uint32_t v = t < 256 ? s_h[t] : 0u;
        const uint32_t inc = wave_incl_scan_u32(v);
        if (lane == 63) s_red[w] = inc;
        __syncthreads();
        uint32_t pre = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (k < w) pre += s_red[k];
        if (t < 256) { s_base[t] = pre + inc - v; s_hn[t] = 0; }
      }
      for (uint32_t c0 = 0; c0 < n; c0 += BL_CH) {
        for (int k = t; k < BL_NW * 128; k += BL_T) reinterpret_cast<uint32_t*>(&s_wcnt[0][0])[k] = 0u;
        __syncthreads();  // (also orders s_base / s_hn of the scan above and the previous chunk's last reads)
        uint32_t sdk[BL_ITEMS], six[BL_ITEMS], lr[BL_ITEMS];
        const uint32_t cend = n - c0 < (uint32_t)BL_CH ? n : c0 + (uint32_t)BL_CH;
        const uint32_t per = ((cend - c0 + 63u) / 64u + BL_NW - 1u) / BL_NW;  // 1..BL_ITEMS
        const uint32_t wb = c0 + (uint32_t)w * per * 64u;
#pragma unroll
        for (int i = 0; i < BL_ITEMS; i++) {
          const uint32_t k = wb + i * 64 + lane;
          uint2 e = make_uint2(0xFFFFFFFFu, 0u);
          if ((uint32_t)i < per && k < cend) e = src[k];
          sdk[i] = e.x; six[i] = e.y;
        }
#pragma unroll
        for (int i = 0; i < BL_ITEMS; i++) {
          lr[i] = 0;
          if ((uint32_t)i >= per || wb + i * 64 >= cend) continue;  // (wave-uniform)
          const bool livel = wb + i * 64 + lane < cend;
          const uint32_t d = (sdk[i] >> sh) & 255u;
          uint32_t count;
          const uint32_t r = match_rank<8>(d, livel, count);
          uint32_t before = 0;
          if (livel) before = s_wcnt[w][d];
          __builtin_amdgcn_wave_barrier();
          if (livel && r == 0) s_wcnt[w][d] = (uint16_t)(before + count);
          __builtin_amdgcn_wave_barrier();
          lr[i] = before + r;
        }
        __syncthreads();
        if (t < 256) {  // digit t: offsets of the waves inside the digit, the chunk's place in the digit's run
          uint32_t tot = 0;
#pragma unroll
          for (int k = 0; k < BL_NW; k++) {
            const uint32_t c = s_wcnt[k][t];
            s_wcnt[k][t] = (uint16_t)tot;
            tot += c;
          }
          const uint32_t cb = s_base[t];
          s_cb[t] = cb;
          s_base[t] = cb + tot;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < BL_ITEMS; i++) {
          if ((uint32_t)i < per && wb + i * 64 + lane < cend) {
            const uint32_t d = (sdk[i] >> sh) & 255u;
            dst[s_cb[d] + s_wcnt[w][d] + lr[i]] = make_uint2(sdk[i], six[i]);
            if (p + 1 < npass) atomicAdd(&s_hn[(sdk[i] >> (sh + 8)) & 255u], 1u);
          }
        }
        __syncthreads();
      }
      if (t < 256) s_h[t] = s_hn[t];
      uint2* tmp = src; src = dst; dst = tmp;
    }
  }
  __syncthreads();  // the ordered pairs (s_stage / src) and s_tc are visible
  BLP(2);  // radix passes

  if (MODE != 1) {
    for (uint32_t i = t; i < n; i += BL_T) {
      const uint32_t idx = fast ? (s_stage[i].y & 0xFFFFu) : src[i].y;
      const uint4 e = ent[s0 + idx];
      sorted_keys[s0 + i] = e.x;
      point_list[s0 + i] = make_uint2(e.z, e.w);
    }
    return;
  }

  // ---- the internal tiles' lists ----
  // this wave's entries of a chunk per tile, in list order -> s_cw[w][tile]
  auto count_chunk = [&](uint32_t c0) {
    const uint32_t cend = n - c0 < (uint32_t)BL_CH ? n : c0 + (uint32_t)BL_CH;
    const uint32_t per = ((cend - c0 + 63u) / 64u + BL_NW - 1u) / BL_NW;
    const uint32_t wb = c0 + (uint32_t)w * per * 64u;
    uint32_t cnt[16];
#pragma unroll
    for (int j = 0; j < 16; j++) cnt[j] = 0;
#pragma unroll 1
    for (uint32_t i = 0; i < per; i++) {
      if (wb + i * 64 >= cend) break;  // (wave-uniform)
      const uint32_t k = wb + i * 64 + lane;
      uint32_t sub = 0;
      if (k < cend) sub = fast ? (s_stage[k].y >> 16) : (ent[s0 + src[k].y].x >> MACRO_KEY_BITS);
#pragma unroll
      for (int j = 0; j < 16; j++) cnt[j] += (uint32_t)__popcll(__ballot((sub >> j) & 1u));
    }
#pragma unroll
    for (int j = 0; j < 16; j++)
      if (lane == j) s_cw[w][j] = cnt[j];
  };
  if (fast) {  // one chunk: its per-wave counts also give the block's entries per tile (the streaming path counted them up front)
    count_chunk(0);
    __syncthreads();
    if (t < 16) {
      uint32_t tot = 0;
#pragma unroll
      for (int k = 0; k < BL_NW; k++) tot += s_cw[k][t];
      s_tc[t] = tot;
    }
    __syncthreads();
  }
  if (t < 16) {
    uint32_t pre = pairs_before;
    for (int j = 0; j < t; j++) pre += s_tc[j];
    s_tb[t] = pre;
    const uint32_t tx = (b % gmx) * BLOCK_BIG + (uint32_t)(t % BLOCK_BIG), ty = (b / gmx) * BLOCK_BIG + (uint32_t)(t / BLOCK_BIG);
    if (tx < gsx && ty < gsy) ranges[ty * gsx + tx] = make_uint2(pre, pre + s_tc[t]);
    put_desc((uint32_t)t, tx, ty, pre, pre + s_tc[t]);
  }
  BLP(3);  // per-tile counts, ranges, descriptors
  {
    // live flags of this block's pairs: which records get written depends only on forward state (lists and n_contrib),
    // so one clear per forward serves every backward over this workspace
    uint32_t total = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) total += s_tc[j];
    const uint32_t a = pairs_before, e = pairs_before + total;
    const uint32_t a4 = (a + 3u) & ~3u, e4 = e & ~3u;
    if (a4 <= e4) {
      for (uint32_t i = a + t; i < a4; i += BL_T) live[i] = 0;
      for (uint32_t i = a4 + 4u * t; i < e4; i += 4u * BL_T) *reinterpret_cast<uint32_t*>(live + i) = 0u;
      for (uint32_t i = e4 + t; i < e; i += BL_T) live[i] = 0;
    } else {
      for (uint32_t i = a + t; i < e; i += BL_T) live[i] = 0;
    }
  }
  __syncthreads();
  BLP(4);  // live flags cleared
  for (uint32_t c0 = 0; c0 < n; c0 += BL_CH) {
    const uint32_t cend = n - c0 < (uint32_t)BL_CH ? n : c0 + (uint32_t)BL_CH;
    const uint32_t per = ((cend - c0 + 63u) / 64u + BL_NW - 1u) / BL_NW;
    const uint32_t wb = c0 + (uint32_t)w * per * 64u;
    if (!fast) {
      count_chunk(c0);
      __syncthreads();
    }
    if (t < 16) {  // tile t: where each wave's entries of this chunk go, then advance the tile's running position
      uint32_t run = s_tb[t];
#pragma unroll
      for (int k = 0; k < BL_NW; k++) {
        const uint32_t c = s_cw[k][t];
        s_cw[k][t] = run;
        run += c;
      }
      s_tb[t] = run;
    }
    __syncthreads();
    uint32_t off[16];
#pragma unroll
    for (int j = 0; j < 16; j++) off[j] = s_cw[w][j];
#pragma unroll 1
    for (uint32_t i = 0; i < per; i++) {
      if (wb + i * 64 >= cend) break;  // (wave-uniform)
      const uint32_t k = wb + i * 64 + lane;
      uint32_t sub = 0, id = 0, sl = 0;
      if (k < cend) {
        if (IDS && fast) {
          const uint32_t sv = s_stage[k].y;  // sub-mask << 16 | position in the entry buffer
          const uint2 is = s_ids[sv & 0xFFFFu];
          sub = sv >> 16; id = is.x; sl = is.y;
        } else {
          const uint4 e = ent[s0 + (fast ? (s_stage[k].y & 0xFFFFu) : src[k].y)];
          sub = e.x >> MACRO_KEY_BITS; id = e.z; sl = e.w;
        }
      }
      // The values are made "arrived" HERE, once, for every lane: each of the sixteen conditional stores below otherwise got its
      // own s_waitcnt vmcnt(0) from the compiler (the first use of id / sl sits in a block that does not dominate the next one),
      // and vmcnt(0) also waits for every STORE before it — sixteen serialised store round trips per group of 64 entries, 9 of a
      // workgroup's 27 us (tools/bl_phases.py with a timer around this loop).
      asm volatile("" : "+v"(sub), "+v"(id), "+v"(sl));
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const bool has = (sub >> j) & 1u;
        const unsigned long long m = __ballot(has);
        if (has)
          point_list[off[j] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
              make_uint2(id, sl + (uint32_t)__popc(sub & ((1u << j) - 1u)));
        off[j] += (uint32_t)__popcll(m);
      }
    }
    __syncthreads();  // s_cw is rewritten by the next chunk
  }
  BLP(5);  // split into the tiles' lists
#ifdef EOGS_BL_PHASES
  if (threadIdx.x == 0) atomicAdd(&g_bl_phase[7], 1ull);
#endif
}

void launch_block_lists(const GeomWS& g, const SortWS& w, const BinWS& b, const ImgWS& im, int P, int H, int W, int64_t R,
                        hipStream_t s) {
  (void)P;
  const int M = b.block;
  const uint32_t gmx = macro_grid_x(W, BLOCK_BIG), nblocks = gmx * macro_grid_y(H, BLOCK_BIG);
  const uint32_t gsx = macro_grid_x(W, 1), gsy = macro_grid_y(H, 1);
  if (nr_entries(R) == 0) {
    (void)hipMemsetAsync(im.ranges, 0, (size_t)(M > 1 ? nblocks : gsx * gsy) * sizeof(uint2), s);
    return;
  }
  const bool inA = entry_sort_result_in_A(H, W);
  const uint4* ent = inA ? w.entA : w.entB;
  uint2* ki = reinterpret_cast<uint2*>(inA ? w.entB : w.entA);
  const bool wide = nr_wide(R) != 0;  // 2800 ... 6000 entries per block on average (api.hip forward_counts)
  const uint32_t cap_slots = nr_slots(R), cap_entries = nr_entries(R) < w.cap ? nr_entries(R) : w.cap;
  // Workgroup size by the entries a block holds on average (R carries the forward's entry count, or a capacity above it: then
  // the larger workgroup is taken, which is only slower). A block's workgroup orders its entries in LDS up to 4 per thread; with a
  // few hundred entries per block — large images: 4096 blocks at 2048^2 — sixteen waves mostly wait at barriers for one or two
  // busy ones, and only two such workgroups fit a CU. Lists longer than 4 x the workgroup stream through global memory (any
  // length is handled), so the thresholds leave room for the spread between blocks.
  static const int forced = [] { const char* e = getenv("EOGS_BL_T"); return e ? atoi(e) : 0; }();  // tuning aid: 256 / 512 / 1024
  const double per_block = (double)nr_entries(R) / (double)nblocks;
  int T = wide ? 1024 : per_block <= BL_NARROW_256 ? 256 : per_block <= BL_NARROW_512 ? 512 : 1024;
  if (!wide && (forced == 256 || forced == 512 || forced == 1024)) T = forced;
  // (Round 6 tried eight entries per thread in workgroups of 256 / 512 threads — a quarter / half of the waves per barrier, four /
  // two blocks per CU at once, ids in LDS: 61 / 50 us against 48 at the headline, 139 / 66 against 64 at trained opacities:
  // profiles/r06_ab_block_lists_small_workgroups.txt. Not kept.)
  if (M > 1) (void)hipMemsetAsync(b.live, 0, (size_t)nr_slots(R), s);
#define BL_LAUNCH(MODE_, ITEMS_, T_)                                                                                            \
  hipLaunchKernelGGL((block_lists_kernel<MODE_, ITEMS_, T_>), dim3(nblocks), dim3(T_), 0, s, ent, ki, g.bcount, g.bpairs, g.misc, \
                     gmx, gsx, gsy, b.point_list, b.sorted_keys, im.ranges, b.live, cap_slots, cap_entries, g.where, im.desc,    \
                     16u * im.sched_lg)
  if (M > 1) {
    if (wide) BL_LAUNCH(BLOCK_BIG, 8, 1024);
    else if (T == 256) BL_LAUNCH(BLOCK_BIG, 4, 256);
    else if (T == 512) BL_LAUNCH(BLOCK_BIG, 4, 512);
    else BL_LAUNCH(BLOCK_BIG, 4, 1024);
  } else {
    if (wide) BL_LAUNCH(1, 8, 1024);
    else if (T == 256) BL_LAUNCH(1, 4, 256);
    else if (T == 512) BL_LAUNCH(1, 4, 512);
    else BL_LAUNCH(1, 4, 1024);
  }
#undef BL_LAUNCH
}

#ifdef EOGS_BL_PHASES
extern "C" int eogs_debug_bl_phases(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_bl_phase), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    void* dptr = nullptr;
    if (hipGetSymbolAddress(&dptr, HIP_SYMBOL(g_bl_phase)) != hipSuccess) return -1;
    if (hipMemset(dptr, 0, 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
