// binning.hip — builds the per-tile, depth-ordered Gaussian lists.
//
// Replaces InclusiveSum + duplicateWithKeys + cub::DeviceRadixSort::SortPairs(64-bit keys, 32+log2(T) bits) +
// identifyTileRanges (DGR/cuda_rasterizer/rasterizer_impl.cu:70-138,280-320) with a pipeline that produces
// the SAME list order — by tile, then depth bits ascending, then Gaussian index — while moving far fewer bytes:
//
//   1. depth sort of the P Gaussians: stable LSD radix on the 32 depth bits, payload = Gaussian id (8 B/item,
//      4 passes over P items instead of 6 passes over R pairs of 12 B);
//   2. expand in depth order: list position = exclusive scan of the per-Gaussian tile counts over the depth-sorted
//      Gaussians (their 32-byte binning records are gathered once into depth order); a workgroup stages its pairs
//      in LDS and writes (tile id, {Gaussian id, record slot}) with contiguous lanes.
//      Lists are built per INTERNAL tile (SUBX x SUBY pixels = one wave64) and only for the internal tiles of the
//      reference's 16-px tile rect in which the Gaussian can reach alpha >= 1/255 (exact hit mask computed
//      in preprocess): a subset of the reference's candidates that contains every (pixel, Gaussian) pair
//      the reference blends, so rendering results are unchanged;
//   3. stable LSD radix on the tile id only (ceil(log2 T) bits, 1-2 passes, 12 B/pair): stability keeps the
//      depth order inside every tile;
//   4. tile ranges from the sorted tile ids.
// A (tile, Gaussian) pair is unique, so (tile, depth bits, index) is a total order and the result is
// bit-identical to the reference's stable 64-bit-key sort.
//
// The depth sort runs only as many 8-bit passes as the key range needs (EOGS depths share their top byte).
// All kernels: 256-thread workgroups (4 wave64); radix ranking is wave-private (ballot match on the digit bits, no
// workgroup barrier), keys are re-ordered through LDS so each digit run is written contiguously. HBM-bound integer work.
#include "common.h"

namespace {

__device__ inline uint32_t wave_incl_scan_u32(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t n = __shfl_up(v, o, 64);
    if (lane >= o) v += n;
  }
  return v;
}

// LDS produced and consumed by the same wave (in-order LDS pipeline): only the compiler must not reorder.
__device__ inline void wave_lds_sync_b() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

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

}  // namespace

// Single workgroup: exclusive scan of the per-workgroup pair counts (-> record slots in Gaussian-id order), their
// 64-bit total (= num_rendered) and the key range, written to misc[] for the host readback.

__device__ inline void pblock_scan_body(uint32_t* __restrict__ pblock, const uint32_t* __restrict__ pbkey, uint32_t nblk,
                                        uint32_t* __restrict__ misc) {
  __shared__ uint32_t s_w[4];
  __shared__ uint32_t s_k[2][BLK / 64];
  unsigned long long carry = 0ull, entries = 0ull, opw = 0ull;
  uint32_t kmax = 0, knmin = 0, err = 0;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (uint32_t b0 = 0; b0 < nblk; b0 += BLK * 16) {
    const uint32_t i0 = b0 + threadIdx.x * 16;
    uint32_t v[16], sum = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      v[k] = 0;
      if (i0 + k < nblk) {
        v[k] = pblock[i0 + k];
        const uint32_t a = pbkey[4 * (i0 + k)], b = pbkey[4 * (i0 + k) + 1];
        kmax = a > kmax ? a : kmax;
        knmin = b > knmin ? b : knmin;
        entries += pbkey[4 * (i0 + k) + 2];
        const uint32_t ow = pbkey[4 * (i0 + k) + 3];  // bit 31: error flag of the workgroup
        opw += ow & 0x7FFFFFFFu;
        err |= ow >> 31;
      }
      sum += v[k];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t nb = __shfl_up(inc, o, 64);
      if (lane >= o) inc += nb;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    const uint32_t w0 = s_w[0], w1 = s_w[1], w2 = s_w[2], w3 = s_w[3];
    const uint32_t pre = (w > 0 ? w0 : 0u) + (w > 1 ? w1 : 0u) + (w > 2 ? w2 : 0u);
    __syncthreads();
    uint32_t run = (uint32_t)carry + pre + inc - sum;  // slots are u32: the host rejects totals >= 2^31
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (i0 + k < nblk) pblock[i0 + k] = run;
      run += v[k];
    }
    carry += (unsigned long long)w0 + w1 + w2 + w3;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const uint32_t a = __shfl_xor(kmax, o, 64), b = __shfl_xor(knmin, o, 64);
    kmax = a > kmax ? a : kmax;
    knmin = b > knmin ? b : knmin;
  }
  // total list entries: every thread holds a partial sum
  __shared__ unsigned long long s_e[BLK / 64], s_o[BLK / 64];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    entries += __shfl_xor(entries, o, 64);
    opw += __shfl_xor(opw, o, 64);
    err |= __shfl_xor(err, o, 64);
  }
  __shared__ uint32_t s_err[BLK / 64];
  if (lane == 0) { s_k[0][w] = kmax; s_k[1][w] = knmin; s_e[w] = entries; s_o[w] = opw; s_err[w] = err; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t a = s_k[0][0], b = s_k[1][0];
    unsigned long long e = s_e[0], ow = s_o[0];
    uint32_t er = s_err[0];
    for (int i = 1; i < BLK / 64; i++) {
      a = s_k[0][i] > a ? s_k[0][i] : a;
      b = s_k[1][i] > b ? s_k[1][i] : b;
      e += s_e[i];
      ow += s_o[i];
      er |= s_err[i];
    }
    misc[MISC_ERR] = er;  // every readback word is written here: the workspace needs no clearing
    misc[MISC_OPW_LO] = (uint32_t)ow;
    misc[MISC_OPW_HI] = (uint32_t)(ow >> 32);
    misc[MISC_MACRO_LO] = (uint32_t)e;
    misc[MISC_MACRO_HI] = (uint32_t)(e >> 32);
    pblock[nblk] = (uint32_t)carry;
    misc[MISC_TOTAL_LO] = (uint32_t)carry;
    misc[MISC_TOTAL_HI] = (uint32_t)(carry >> 32);
    misc[MISC_KEY_MAX] = a;
    misc[MISC_KEY_NMIN] = b;
  }
}


// ---- radix pass, kernel 1: per-workgroup digit histogram, written digit-major hist[d][blk] ----
// The first pass of the depth sort runs one extra workgroup (blockIdx.x == nblk, `pblock` non-NULL) that does the
// single-workgroup scan of the preprocess pair counts beside the histogram workgroups: one launch less on the forward's
// critical path (a dependent single-workgroup kernel costs ~7 us however little it does).
template <int ITEMS>
__global__ __launch_bounds__(BLK) void radix_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, int shift,
                                                         uint32_t mask, uint32_t* __restrict__ hist, uint32_t nblk,
                                                         uint32_t* __restrict__ pblock, const uint32_t* __restrict__ pbkey,
                                                         uint32_t npb, uint32_t* __restrict__ misc) {
  if (blockIdx.x == nblk) {  // (only launched when pblock != NULL)
    pblock_scan_body(pblock, pbkey, npb, misc);
    return;
  }
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
// Wave w of the workgroup owns the contiguous segment [base + w*64*ITEMS, base + (w+1)*64*ITEMS) and takes it 64 keys
// at a time (lane order == key order), so ranking needs NO workgroup barrier: per chunk a wave-level match by ballot
// on the digit bits gives the rank among equal digits inside the chunk, a wave-private LDS counter row gives the
// count of that digit in the wave's earlier chunks. One barrier later the workgroup knows, per digit, its start in the
// workgroup's sorted order and each wave's offset inside the digit; keys and payloads are then placed in LDS in
// sorted order and streamed out so that consecutive lanes write consecutive addresses inside each digit run.
// Stable: (wave segment, chunk, lane) order is index order.
template <int ITEMS, typename VT>
__global__ __launch_bounds__(BLK) void radix_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                            const VT* __restrict__ vals_in,
                                                            uint32_t* __restrict__ keys_out,
                                                            VT* __restrict__ vals_out, uint32_t n, int shift,
                                                            int nbits, const uint32_t* __restrict__ hist,
                                                            uint32_t nblk, const uint32_t* __restrict__ dtotal) {
  constexpr int NW = BLK / 64, SEG = 64 * ITEMS, TILE_KEYS = BLK * ITEMS;
  __shared__ uint32_t s_gbase[256];     // global output position of this workgroup's first key of each digit
  __shared__ uint32_t s_dstart[256];    // start of each digit in the workgroup's sorted order
  __shared__ uint32_t s_wcnt[NW][256];  // per-wave digit counts -> per-wave offset inside the digit
  __shared__ uint32_t s_key[TILE_KEYS];
  __shared__ VT s_val[TILE_KEYS];
  __shared__ uint32_t s_w[4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const uint32_t mask = (1u << nbits) - 1u;
#pragma unroll
  for (int k = 0; k < NW; k++) s_wcnt[k][t] = 0;
  __syncthreads();

  const uint32_t base = blockIdx.x * (uint32_t)TILE_KEYS + (uint32_t)w * SEG;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint32_t key[ITEMS], lrank[ITEMS];
  VT val[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; i++) {
    const uint32_t k = base + i * 64 + lane;
    key[i] = k < n ? keys_in[k] : 0xFFFFFFFFu;
    if (k < n) val[i] = vals_in[k];
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

template <int ITEMS>
static void radix_hist(const uint32_t* kin, uint32_t n, int shift, int nbits, uint32_t* hist, uint32_t nblk, hipStream_t s) {
  hipLaunchKernelGGL((radix_hist_kernel<ITEMS>), dim3(nblk), dim3(BLK), 0, s, kin, n, shift, (1u << nbits) - 1u, hist, nblk,
                     (uint32_t*)nullptr, (const uint32_t*)nullptr, 0u, (uint32_t*)nullptr);
}

template <int ITEMS, typename VT>
static void radix_pass(const uint32_t* kin, const VT* vin, uint32_t* kout, VT* vout, uint32_t n, int shift,
                       int nbits, uint32_t* hist, uint32_t nblk, uint32_t* dtotal, hipStream_t s, bool hist_done = false) {
  const uint32_t mask = (1u << nbits) - 1u;
  if (!hist_done) radix_hist<ITEMS>(kin, n, shift, nbits, hist, nblk, s);
  hipLaunchKernelGGL(radix_rowscan_kernel, dim3(mask + 1), dim3(BLK), 0, s, hist, nblk, dtotal);
  hipLaunchKernelGGL((radix_scatter_kernel<ITEMS, VT>), dim3(nblk), dim3(BLK), 0, s, kin, vin, kout, vout, n, shift, nbits,
                     hist, nblk, dtotal);
}

// Depth sort: stable 8-bit passes over the depth bits, ping-ponging A -> B -> A ...; after an even number of passes
// the ids in depth order are in svalA.
void launch_depth_sort(const GeomWS& g, int P, int first, int last, hipStream_t s, bool first_hist_done) {
  for (int pass = first; pass < last; pass++) {
    const bool a2b = (pass & 1) == 0;
    radix_pass<SORTP_ITEMS, uint32_t>(a2b ? g.skeyA : g.skeyB, a2b ? g.svalA : g.svalB, a2b ? g.skeyB : g.skeyA,
                                      a2b ? g.svalB : g.svalA, (uint32_t)P, 8 * pass, 8, g.hist, g.nblkP, g.dtotal, s,
                                      first_hist_done && pass == first);
  }
}

// Histogram of the depth sort's pass 0 + (one extra workgroup) the scan of the preprocess pair counts, whose totals the
// host reads back: forward_prepare records its readback event right after this launch.
void launch_depth_sort_head(const GeomWS& g, int P, hipStream_t s) {
  hipLaunchKernelGGL((radix_hist_kernel<SORTP_ITEMS>), dim3(g.nblkP + 1), dim3(BLK), 0, s, g.skeyA, (uint32_t)P, 0, 255u, g.hist,
                     g.nblkP, g.pblock, g.pbkey, ceil_div_u32((uint64_t)P, BLK), g.misc);
}

void launch_sort_u32(uint32_t* keyA, uint32_t* valA, uint32_t* keyB, uint32_t* valB, uint32_t n, int passes, uint32_t* hist,
                     uint32_t nblk, uint32_t* dtotal, hipStream_t s) {
  for (int pass = 0; pass < passes; pass++) {
    const bool a2b = (pass & 1) == 0;
    radix_pass<SORTP_ITEMS, uint32_t>(a2b ? keyA : keyB, a2b ? valA : valB, a2b ? keyB : keyA, a2b ? valB : valA, n,
                                      8 * pass, 8, hist, nblk, dtotal, s);
  }
}

// ---- exclusive scan of a small array in place (single workgroup, 16 elements per thread per round);
//      data[n] receives the total ----
__global__ __launch_bounds__(BLK) void small_scan_kernel(uint32_t* __restrict__ data, uint32_t n) {
  __shared__ uint32_t s_w[4];
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 < n; b0 += BLK * 16) {
    const uint32_t i0 = b0 + threadIdx.x * 16;
    uint32_t v[16], sum = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      v[k] = i0 + k < n ? data[i0 + k] : 0u;
      sum += v[k];
    }
    uint32_t tot;
    uint32_t run = carry + block_excl_scan(sum, s_w, tot);
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (i0 + k < n) data[i0 + k] = run;
      run += v[k];
    }
    carry += tot;
  }
  if (threadIdx.x == 0) data[n] = carry;
}
void launch_small_scan(uint32_t* data, uint32_t n, hipStream_t s) {
  hipLaunchKernelGGL(small_scan_kernel, dim3(1), dim3(BLK), 0, s, data, n);
}

// ---- expand step A: pair count of each chunk of 256 depth-sorted Gaussians ----
__global__ __launch_bounds__(BLK) void expand_count_kernel(const uint32_t* __restrict__ sorted_ids,
                                                           const uint4* __restrict__ binfo, uint32_t P, int big,
                                                           uint4* __restrict__ sinfo, uint32_t* __restrict__ blocksum) {
  __shared__ uint32_t s_w[4];
  const uint32_t k = blockIdx.x * BLK + threadIdx.x;
  uint32_t v = 0;
  if (k < P) {
    // the only per-Gaussian gather of the binning stage: one 32-byte record, re-written in depth order
    const uint32_t id = sorted_ids[k];
    const uint4 a = binfo[2 * (size_t)id], b = binfo[2 * (size_t)id + 1];
    sinfo[2 * (size_t)k] = a;
    sinfo[2 * (size_t)k + 1] = b;
    v = b.x ? (big ? (b.w >> 2) : b.x) : 0u;  // list entries of this Gaussian at the chosen block size
  }
  uint32_t tot;
  (void)block_excl_scan(v, s_w, tot);
  if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}

#define EXPAND_LANE_MAX 256u  // a single lane walks at most this many list entries
// ---- expand step C, block size 1: emission of (internal tile id, record slot) in depth order ----
// One lane per depth-sorted Gaussian. Its listed tiles are, by kind (GeomWS::binfo): the set bits of the hit mask, the
// per-row column spans (row_span, re-evaluated on the bits preprocess counted with), or the whole rect.
// Output position = depth-order offset (exclusive scan of the counts); payload = {Gaussian id, record slot in
// Gaussian-id order}, carried through the tile sort so the render kernels read both with one coalesced load.
// Gaussians with at most EXPAND_LANE_MAX tiles are walked by their own lane into an LDS window (tile id + owner lane),
// FINE_STAGE pairs per round, and streamed out with consecutive lanes writing consecutive addresses; larger ones
// are emitted by the whole wave, one after the other, straight to their (reserved) global positions.
#ifndef FINE_STAGE
#define FINE_STAGE 3072     // pairs per LDS window (12 KB of tile ids + 6 KB of owner lanes: 7 workgroups per CU; 6144 held the
#endif                      // kernel at 3 per CU: binning -4 us at 4 and at 8.7 listed tiles per Gaussian; 1024 costs rounds at 8.7)
struct FineItem {
  uint32_t id, c, pos0, rbase, sx0, sy0, sw, sh;
  unsigned long long m;
};
__device__ inline uint32_t fine_tile_of(const FineItem& it, uint32_t bit_or_q, uint32_t gsx) {
  const uint32_t row = bit_or_q / it.sw, col = bit_or_q - row * it.sw;
  return (it.sy0 + row) * gsx + it.sx0 + col;
}
__global__ __launch_bounds__(BLK) void expand_fine_kernel(const uint4* __restrict__ sinfo, const float4* __restrict__ bext,
                                                     const uint32_t* __restrict__ pblock,
                                                     const uint32_t* __restrict__ blocksum, uint32_t P, uint32_t gsx,
                                                     uint32_t gsy, uint32_t* __restrict__ tkey,
                                                     uint2* __restrict__ tval, uint2* __restrict__ ranges) {
  __shared__ uint32_t s_w[4];
  // the tile ranges are rewritten after the sort (tile_ranges_kernel): clear them here
  for (uint32_t i = blockIdx.x * BLK + threadIdx.x; i < gsx * gsy; i += gridDim.x * BLK) ranges[i] = make_uint2(0u, 0u);
  __shared__ uint32_t s_tk[FINE_STAGE];   // staged tile ids
  __shared__ uint16_t s_own[FINE_STAGE];  // ... and the lane that owns each staged pair
  __shared__ uint32_t s_id[BLK], s_l0[BLK], s_gp[BLK], s_rb[BLK];
  const int lane = threadIdx.x & 63;
  const uint32_t k = blockIdx.x * BLK + threadIdx.x;
  FineItem it;
  it.id = 0; it.c = 0; it.m = 0ull; it.sx0 = it.sy0 = 0; it.sw = 1; it.sh = 0; it.rbase = 0;
  float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = e0;  // SpanParams of a BK_SPANS Gaussian
  uint32_t kind = BK_RECT;
  uint4 ia = make_uint4(0u, 0u, 0u, 0u), ib = ia;
  if (k < P) {
    ia = sinfo[2 * (size_t)k];
    ib = sinfo[2 * (size_t)k + 1];
    it.c = ib.x;
    it.id = ib.z;
  }
  uint32_t tot;
  it.pos0 = blocksum[blockIdx.x] + block_excl_scan(it.c, s_w, tot);
  if (it.c) {
    it.m = ((unsigned long long)ia.w << 32) | ia.z;
    it.sx0 = ia.x & 0xFFFFu; it.sy0 = ia.y & 0xFFFFu;
    it.sw = (ia.x >> 16) - it.sx0;  // internal-tile rect, already clipped (preprocess_fwd_kernel)
    it.sh = (ia.y >> 16) - it.sy0;
    it.rbase = pblock[it.id / BLK] + ib.y;  // pblock is 4 bytes per 256 Gaussians: cache resident
    kind = ib.w & 3u;  // (the upper bits hold the entry count at block size BLOCK_BIG)
    if (kind == BK_SPANS) {
      e0 = bext[2 * (size_t)it.id];
      e1 = bext[2 * (size_t)it.id + 1];
    }
  }
  SpanParams sp;
  sp.gx = e0.x; sp.gy = e0.y; sp.ex = e0.z; sp.ey = e0.w; sp.boa = e1.x; sp.boc = e1.y; sp.ta = e1.z; sp.da = e1.w;

  // ---- lane-walked Gaussians: compact local positions among themselves, staged in rounds of FINE_STAGE ----
  const bool mine = it.c != 0u && it.c <= EXPAND_LANE_MAX;
  uint32_t ltot;
  const uint32_t l0 = block_excl_scan(mine ? it.c : 0u, s_w, ltot);
  s_id[threadIdx.x] = it.id; s_l0[threadIdx.x] = l0; s_gp[threadIdx.x] = it.pos0; s_rb[threadIdx.x] = it.rbase;
  // per-lane cursor, carried across rounds so that every pair is generated exactly once
  uint32_t l = l0;                     // local position of the lane's next pair
  const uint32_t l_end = mine ? l0 + it.c : l0;
  unsigned long long m_rem = it.m;     // BK_MASK: bits still to emit
  uint32_t row = 0, q_cur = 0;         // BK_SPANS: current row / BK_RECT: next index
  int c_cur = 0, c_end = 0;            // BK_SPANS: remaining columns of the current row
  for (uint32_t base = 0; base < ltot; base += FINE_STAGE) {
    __syncthreads();  // s_id.. visible (first round) / previous window drained
    const uint32_t wend = base + FINE_STAGE < l_end ? base + FINE_STAGE : l_end;  // this lane stops here this round
    if (l < wend) {  // (l >= base always: windows are consecutive and the lane stopped at the previous window's end)
      if (kind == BK_MASK) {
        for (; l < wend; l++, m_rem &= m_rem - 1ull) {
          s_tk[l - base] = fine_tile_of(it, (uint32_t)__builtin_ctzll(m_rem), gsx);
          s_own[l - base] = (uint16_t)threadIdx.x;
        }
      } else if (kind == BK_SPANS) {
        while (l < wend) {
          if (c_cur >= c_end) {  // next non-empty row
            row_span(sp, (int)(it.sy0 + row), (int)it.sx0, (int)(it.sx0 + it.sw), c_cur, c_end);
            row++;
            if (row > it.sh) break;  // never: the spans add up to it.c
            continue;
          }
          s_tk[l - base] = (it.sy0 + row - 1) * gsx + (uint32_t)c_cur;
          s_own[l - base] = (uint16_t)threadIdx.x;
          c_cur++;
          l++;
        }
      } else {
        for (; l < wend; l++, q_cur++) {
          s_tk[l - base] = fine_tile_of(it, q_cur, gsx);
          s_own[l - base] = (uint16_t)threadIdx.x;
        }
      }
    }
    __syncthreads();
    const uint32_t nwin = ltot - base < (uint32_t)FINE_STAGE ? ltot - base : (uint32_t)FINE_STAGE;
    for (uint32_t i = threadIdx.x; i < nwin; i += BLK) {
      const uint32_t o = s_own[i], q = base + i - s_l0[o];  // q-th listed tile of its Gaussian
      tkey[s_gp[o] + q] = s_tk[i];
      tval[s_gp[o] + q] = make_uint2(s_id[o], s_rb[o] + q);
    }
  }

  // ---- large Gaussians: the wave emits them cooperatively, one after the other ----
  __syncthreads();  // the last window is drained: s_tk becomes four wave-private staging areas
  uint32_t* wstage = s_tk + (threadIdx.x >> 6) * (FINE_STAGE / 4);
  unsigned long long big = __ballot(it.c > EXPAND_LANE_MAX);
  while (big) {
    const int src = __builtin_ctzll(big);
    big &= big - 1ull;
    FineItem g;
    g.id = __shfl(it.id, src, 64); g.c = __shfl(it.c, src, 64); g.pos0 = __shfl(it.pos0, src, 64);
    g.rbase = __shfl(it.rbase, src, 64); g.sx0 = __shfl(it.sx0, src, 64); g.sy0 = __shfl(it.sy0, src, 64);
    g.sw = __shfl(it.sw, src, 64); g.sh = __shfl(it.sh, src, 64);
    if (__shfl(kind, src, 64) != BK_SPANS) {  // whole rect (a mask never has more than 64 tiles)
      for (uint32_t q = lane; q < g.c; q += 64) {
        tkey[g.pos0 + q] = fine_tile_of(g, q, gsx);
        tval[g.pos0 + q] = make_uint2(g.id, g.rbase + q);
      }
      continue;
    }
    // BK_SPANS: lane = row of the rect (64 rows per step); a wave scan of the span lengths gives every row its place
    SpanParams gs;
    gs.gx = __shfl(sp.gx, src, 64); gs.gy = __shfl(sp.gy, src, 64); gs.ex = __shfl(sp.ex, src, 64);
    gs.ey = __shfl(sp.ey, src, 64); gs.boa = __shfl(sp.boa, src, 64); gs.boc = __shfl(sp.boc, src, 64);
    gs.ta = __shfl(sp.ta, src, 64); gs.da = __shfl(sp.da, src, 64);
    uint32_t done = 0;  // pairs of this Gaussian emitted so far
    for (uint32_t r0 = 0; r0 < g.sh; r0 += 64) {
      const uint32_t row = r0 + (uint32_t)lane;
      int c0 = 0, c1 = 0;
      if (row < g.sh) row_span(gs, (int)(g.sy0 + row), (int)g.sx0, (int)(g.sx0 + g.sw), c0, c1);
      const uint32_t len = (uint32_t)(c1 - c0);
      const uint32_t inc = wave_incl_scan_u32(len);
      const uint32_t chunk = __shfl(inc, 63, 64), off = inc - len;
      const uint32_t t0 = (g.sy0 + row) * gsx + (uint32_t)c0;
      if (chunk <= (uint32_t)(FINE_STAGE / 4)) {
        // through wave-private LDS so that consecutive lanes write consecutive addresses
        for (uint32_t j = 0; j < len; j++) wstage[off + j] = t0 + j;
        wave_lds_sync_b();
        for (uint32_t i = lane; i < chunk && done + i < g.c; i += 64) {
          tkey[g.pos0 + done + i] = wstage[i];
          tval[g.pos0 + done + i] = make_uint2(g.id, g.rbase + done + i);
        }
        wave_lds_sync_b();
      } else {
        for (uint32_t j = 0; j < len && done + off + j < g.c; j++) {
          tkey[g.pos0 + done + off + j] = t0 + j;
          tval[g.pos0 + done + off + j] = make_uint2(g.id, g.rbase + done + off + j);
        }
      }
      done += chunk;
    }
    // never taken (the spans are a pure function of the stored bits); keeps every slot a valid tile id regardless
    for (uint32_t i = done + lane; i < g.c; i += 64) {
      tkey[g.pos0 + i] = g.sy0 * gsx + g.sx0;
      tval[g.pos0 + i] = make_uint2(g.id, g.rbase + i);
    }
  }
}

// ---- expand step C: emission of the list entries in depth order ----
// One lane per depth-sorted Gaussian. An entry = (macro block, Gaussian): key = block id | sub-mask << 16 (which internal
// tiles of the block list the Gaussian), payload = {Gaussian id, record slot of the entry's first listed internal tile}
// (the q-th set bit of the sub-mask owns slot + q; slots run through a Gaussian's entries in emission order and are
// Gaussian-id ordered across Gaussians, see GeomWS::pblock). The blocks come from walk_macro_row (common.h), the same
// code preprocess counted with. Output position = depth-order offset (exclusive scan of the entry counts).
// Gaussians with at most EXPAND_LANE_MAX entries are walked by their own lane into an LDS window and streamed out with
// consecutive lanes writing consecutive addresses; larger ones are emitted by the whole wave, one macro row per lane.
#define EXPAND_STAGE 4096     // entries per LDS window (16 KB keys + 16 KB slots + 8 KB owner lanes; 2048 / 1024 measured: no better)
struct ExpandItem {
  uint32_t id, c, pos0, rbase, sx0, sy0, sx1, sy1, kind;
  unsigned long long m;
};
template <int MACRO>
__global__ __launch_bounds__(BLK) void expand_kernel(const uint4* __restrict__ sinfo, const float4* __restrict__ bext,
                                                     const uint32_t* __restrict__ pblock,
                                                     const uint32_t* __restrict__ blocksum, uint32_t P, uint32_t gmx,
                                                     uint32_t nblocks, uint32_t* __restrict__ tkey,
                                                     uint2* __restrict__ tval, uint2* __restrict__ ranges) {
  __shared__ uint32_t s_w[4];
  // the block ranges are rewritten after the sort (tile_ranges_kernel): clear them here
  for (uint32_t i = blockIdx.x * BLK + threadIdx.x; i < nblocks; i += gridDim.x * BLK) ranges[i] = make_uint2(0u, 0u);
  __shared__ uint32_t s_tk[EXPAND_STAGE];   // staged keys
  __shared__ uint32_t s_sl[EXPAND_STAGE];   // ... record slots
  __shared__ uint16_t s_own[EXPAND_STAGE];  // ... and the lane that owns each staged entry
  __shared__ uint32_t s_id[BLK], s_l0[BLK], s_gp[BLK];
  const int lane = threadIdx.x & 63;
  const uint32_t k = blockIdx.x * BLK + threadIdx.x;
  ExpandItem it;
  it.id = 0; it.c = 0; it.m = 0ull; it.sx0 = it.sy0 = it.sx1 = it.sy1 = 0; it.rbase = 0; it.kind = BK_RECT;
  float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = e0;  // SpanParams of a BK_SPANS Gaussian
  uint4 ia = make_uint4(0u, 0u, 0u, 0u), ib = ia;
  if (k < P) {
    ia = sinfo[2 * (size_t)k];
    ib = sinfo[2 * (size_t)k + 1];
    it.c = ib.x ? (MACRO > 1 ? (ib.w >> 2) : ib.x) : 0u;
    it.id = ib.z;
  }
  uint32_t tot;
  it.pos0 = blocksum[blockIdx.x] + block_excl_scan(it.c, s_w, tot);
  if (it.c) {
    it.m = ((unsigned long long)ia.w << 32) | ia.z;
    it.sx0 = ia.x & 0xFFFFu; it.sy0 = ia.y & 0xFFFFu;  // internal-tile rect, already clipped (preprocess_fwd_kernel)
    it.sx1 = ia.x >> 16; it.sy1 = ia.y >> 16;
    it.rbase = pblock[it.id / BLK] + ib.y;  // pblock is 4 bytes per 256 Gaussians: cache resident
    it.kind = ib.w & 3u;
    if (it.kind == BK_SPANS) {
      e0 = bext[2 * (size_t)it.id];
      e1 = bext[2 * (size_t)it.id + 1];
    }
  }
  SpanParams sp;
  sp.gx = e0.x; sp.gy = e0.y; sp.ex = e0.z; sp.ey = e0.w; sp.boa = e1.x; sp.boc = e1.y; sp.ta = e1.z; sp.da = e1.w;

  // ---- lane-walked Gaussians: compact local positions among themselves, staged in rounds of EXPAND_STAGE ----
  const bool mine = it.c != 0u && it.c <= EXPAND_LANE_MAX;
  uint32_t ltot;
  const uint32_t l0 = block_excl_scan(mine ? it.c : 0u, s_w, ltot);
  s_id[threadIdx.x] = it.id; s_l0[threadIdx.x] = l0; s_gp[threadIdx.x] = it.pos0;
  for (uint32_t base = 0; base < ltot; base += EXPAND_STAGE) {
    __syncthreads();  // s_id.. visible (first round) / previous window drained
    if (mine && l0 < base + EXPAND_STAGE && l0 + it.c > base) {
      uint32_t l = l0, slot = it.rbase;
      for (int MY = (int)it.sy0 / MACRO; MY <= ((int)it.sy1 - 1) / MACRO; MY++)  // MACRO: template parameter
        walk_macro_row<MACRO>(it.kind, it.m, sp, (int)it.sx0, (int)it.sy0, (int)it.sx1, (int)it.sy1, MY, [&](int MX, uint32_t sub) {
          const uint32_t w = l - base;  // wraps below the window: fails the unsigned test
          if (w < (uint32_t)EXPAND_STAGE) {
            s_tk[w] = ((uint32_t)MY * gmx + (uint32_t)MX) | (MACRO > 1 ? sub << MACRO_KEY_BITS : 0u);
            s_sl[w] = slot;
            s_own[w] = (uint16_t)threadIdx.x;
          }
          l++;
          slot += (uint32_t)__popc(sub);
        });
    }
    __syncthreads();
    const uint32_t nwin = ltot - base < (uint32_t)EXPAND_STAGE ? ltot - base : (uint32_t)EXPAND_STAGE;
    for (uint32_t i = threadIdx.x; i < nwin; i += BLK) {
      const uint32_t o = s_own[i], q = base + i - s_l0[o];  // q-th entry of its Gaussian
      tkey[s_gp[o] + q] = s_tk[i];
      tval[s_gp[o] + q] = make_uint2(s_id[o], s_sl[i]);
    }
  }

  // ---- large Gaussians: the wave emits them cooperatively, one after the other, one macro row per lane ----
  unsigned long long big = __ballot(it.c > EXPAND_LANE_MAX);
  while (big) {
    const int src = __builtin_ctzll(big);
    big &= big - 1ull;
    ExpandItem g;
    g.id = __shfl(it.id, src, 64); g.c = __shfl(it.c, src, 64); g.pos0 = __shfl(it.pos0, src, 64);
    g.rbase = __shfl(it.rbase, src, 64); g.sx0 = __shfl(it.sx0, src, 64); g.sy0 = __shfl(it.sy0, src, 64);
    g.sx1 = __shfl(it.sx1, src, 64); g.sy1 = __shfl(it.sy1, src, 64); g.kind = __shfl(it.kind, src, 64);
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
      if (MY <= MY1)
        walk_macro_row<MACRO>(g.kind, g.m, gs, (int)g.sx0, (int)g.sy0, (int)g.sx1, (int)g.sy1, MY, [&](int, uint32_t sub) {
          ne++;
          nf += (uint32_t)__popc(sub);
        });
      const uint32_t ie = wave_incl_scan_u32(ne), jf = wave_incl_scan_u32(nf);
      uint32_t l = done + ie - ne, slot = slot0 + jf - nf;
      if (MY <= MY1)
        walk_macro_row<MACRO>(g.kind, g.m, gs, (int)g.sx0, (int)g.sy0, (int)g.sx1, (int)g.sy1, MY, [&](int MX, uint32_t sub) {
          if (l < g.c) {  // always (same walk as the count)
            tkey[g.pos0 + l] = ((uint32_t)MY * gmx + (uint32_t)MX) | (MACRO > 1 ? sub << MACRO_KEY_BITS : 0u);
            tval[g.pos0 + l] = make_uint2(g.id, slot);
          }
          l++;
          slot += (uint32_t)__popc(sub);
        });
      done += __shfl(ie, 63, 64);
      slot0 += __shfl(jf, 63, 64);
    }
  }
}

// ---- block ranges from the sorted keys (identifyTileRanges, rasterizer_impl.cu:116-138) ----
// With per-tile lists (one record slot per entry) it also clears the backward's per-record live flags.
// Four consecutive keys per thread (one 16-byte load, one 4-byte store of cleared flags).
__global__ __launch_bounds__(BLK) void tile_ranges_kernel(const uint32_t* __restrict__ skeys, uint32_t R, uint32_t kmask,
                                                          uint2* __restrict__ ranges, uint8_t* __restrict__ live) {
  const uint32_t i0 = (blockIdx.x * BLK + threadIdx.x) * 4u;
  if (i0 >= R) return;
  uint32_t k[4];
  if (i0 + 4u <= R) {
    const uint4 v = *reinterpret_cast<const uint4*>(skeys + i0);
    k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w;
    if (live) *reinterpret_cast<uint32_t*>(live + i0) = 0u;
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      k[j] = i0 + j < R ? skeys[i0 + j] : 0u;
      if (live && i0 + j < R) live[i0 + j] = 0;
    }
  }
  uint32_t prev = i0 ? skeys[i0 - 1] & kmask : 0u;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t i = i0 + j;
    if (i >= R) break;
    const uint32_t cur = k[j] & kmask;
    if (i == 0) ranges[cur].x = 0;
    else if (cur != prev) {
      ranges[prev].y = i;
      ranges[cur].x = i;
    }
    if (i == R - 1) ranges[cur].y = R;
    prev = cur;
  }
}

void launch_binning_head(const GeomWS& g, int P, int block, const uint32_t* sorted_ids, hipStream_t s) {
  hipLaunchKernelGGL(expand_count_kernel, dim3(g.nblkE), dim3(BLK), 0, s, sorted_ids, g.binfo, (uint32_t)P, (int)(block > 1),
                     g.sinfo, g.blocksum);
  // (Folding this scan into the emission kernels was measured twice: through per-group atomic sums the 4096 atomicAdds on 64
  // addresses cost expand_count 14 us; with every emission workgroup summing the counts before it by itself — no atomics,
  // <= 16 coalesced loads per thread — binning did not change at 4 listed tiles per Gaussian and lost 4 us in block mode.)
  launch_small_scan(g.blocksum, g.nblkE, s);
}

void launch_binning(const GeomWS& g, const BinWS& b, const ImgWS& im, int P, int H, int W, int64_t R, hipStream_t s) {
  const int M = b.block;
  const uint32_t gmx = macro_grid_x(W, M), nblocks = gmx * macro_grid_y(H, M);
  const uint32_t Re = nr_entries(R), Rs = nr_slots(R);
  if (Re == 0) {
    (void)hipMemsetAsync(im.ranges, 0, (size_t)nblocks * sizeof(uint2), s);
    return;
  }
  // The backward's per-record live flags: which records get written depends only on forward state (lists and
  // n_contrib), so one clear per forward serves every backward over this workspace.
  if (M > 1) (void)hipMemsetAsync(b.live, 0, (size_t)Rs, s);  // (per-tile lists: cleared by tile_ranges_kernel)
  if (M > 1)
    hipLaunchKernelGGL(expand_kernel<BLOCK_BIG>, dim3(g.nblkE), dim3(BLK), 0, s, g.sinfo, g.bext, g.pblock, g.blocksum,
                       (uint32_t)P, gmx, nblocks, b.tkeyA, b.tvalA, im.ranges);
  else  // per-tile lists: the specialised walker (set bits / spans directly, cursors carried across LDS windows)
    hipLaunchKernelGGL(expand_fine_kernel, dim3(g.nblkE), dim3(BLK), 0, s, g.sinfo, g.bext, g.pblock, g.blocksum,
                       (uint32_t)P, gmx, macro_grid_y(H, 1), b.tkeyA, b.tvalA, im.ranges);
  uint32_t *ka = b.tkeyA, *kb = b.tkeyB;
  uint2 *va = b.tvalA, *vb = b.tvalB;
  int shift = 0;
  for (int pass = 0; pass < b.passes; pass++) {
    const int nbits = (b.tile_bits - shift) < b.bits_per_pass ? (b.tile_bits - shift) : b.bits_per_pass;
    if (b.sort_items == SORTR_ITEMS_BIG)
      radix_pass<SORTR_ITEMS_BIG, uint2>(ka, va, kb, vb, Re, shift, nbits, b.hist, b.nblkR, b.dtotal, s);
    else
      radix_pass<SORTR_ITEMS, uint2>(ka, va, kb, vb, Re, shift, nbits, b.hist, b.nblkR, b.dtotal, s);
    shift += nbits;
    uint32_t* tk = ka; ka = kb; kb = tk;
    uint2* tv = va; va = vb; vb = tv;
  }
  hipLaunchKernelGGL(tile_ranges_kernel, dim3(ceil_div_u32((uint64_t)Re, BLK * 4)), dim3(BLK), 0, s, b.sorted_keys, Re,
                     M > 1 ? (1u << MACRO_KEY_BITS) - 1u : 0xFFFFFFFFu, im.ranges, M > 1 ? (uint8_t*)nullptr : b.live);
}
