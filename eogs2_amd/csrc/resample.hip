// resample.hip — fused virtual-camera resample: uv = cam2virt (u,v,alt), bilinear grid sample (align_corners=True,
// zeros padding) of the virtual render, out-of-view fill; forward and backward.
// Reference semantics: src/gaussiansplatting/gaussian_renderer/renderer_cc_shadow.py:32-50 on top of
// torch.nn.functional.grid_sample (ATen grid_sampler_2d: unnormalize ((x+1)/2)(size-1), floor corner, weights from
// the opposite corner, taps outside the image contribute zero and receive no gradient).
// One lane per output pixel; neighbouring lanes sample neighbouring taps, so the gathers coalesce in L2. HBM-bound.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"

namespace {


struct Taps {
  int x0, y0;           // north-west corner
  float wx1, wy1;       // fractional parts: weight of the east / south neighbours
  bool in_x0, in_x1, in_y0, in_y1;
};

__device__ inline Taps make_taps(float u, float v, int Wv, int Hv) {
  Taps t;
  const float ix = (u + 1.f) * 0.5f * (float)(Wv - 1), iy = (v + 1.f) * 0.5f * (float)(Hv - 1);
  const float fx = floorf(ix), fy = floorf(iy);
  t.wx1 = ix - fx; t.wy1 = iy - fy;
  // clamp before the int conversion: far-away coordinates only need to end up out of bounds
  t.x0 = (int)fminf(fmaxf(fx, -2.f), (float)Wv + 1.f);
  t.y0 = (int)fminf(fmaxf(fy, -2.f), (float)Hv + 1.f);
  t.in_x0 = t.x0 >= 0 && t.x0 < Wv; t.in_x1 = t.x0 + 1 >= 0 && t.x0 + 1 < Wv;
  t.in_y0 = t.y0 >= 0 && t.y0 < Hv; t.in_y1 = t.y0 + 1 >= 0 && t.y0 + 1 < Hv;
  return t;
}

__global__ __launch_bounds__(BLK) void resample_fwd_kernel(int C, int Hv, int Wv, int HW, int n_out,
                                                           const float* __restrict__ vr, const float* __restrict__ uva,
                                                           const float* __restrict__ M, int fill_channel, float fill_value,
                                                           float* __restrict__ sample, float* __restrict__ uv) {
  const int p = blockIdx.x * BLK + threadIdx.x;
  if (p >= HW) return;
  const float a = uva[3 * (size_t)p], b = uva[3 * (size_t)p + 1], c = uva[3 * (size_t)p + 2];
  const float u = M[0] * a + M[1] * b + M[2] * c, v = M[3] * a + M[4] * b + M[5] * c;
  reinterpret_cast<float2*>(uv)[p] = make_float2(u, v);
  const Taps t = make_taps(u, v, Wv, Hv);
  const float wnw = (1.f - t.wx1) * (1.f - t.wy1), wne = t.wx1 * (1.f - t.wy1), wsw = (1.f - t.wx1) * t.wy1, wse = t.wx1 * t.wy1;
  const size_t plane = (size_t)Hv * Wv;
  const size_t o00 = (size_t)t.y0 * Wv + t.x0;
  const bool outside = fabsf(u) > 1.f || fabsf(v) > 1.f;
  // the taps of up to four channels are requested together (the reference resamples four; a runtime-bounded loop made every
  // channel wait for the previous one's four gathers)
  for (int c0 = 0; c0 < n_out; c0 += 4) {
    float v[4][4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool on = c0 + k < n_out;
      const float* src = vr + (size_t)(on ? c0 + k : 0) * plane;
      v[k][0] = (on && t.in_y0 && t.in_x0) ? src[o00] : 0.f;
      v[k][1] = (on && t.in_y0 && t.in_x1) ? src[o00 + 1] : 0.f;
      v[k][2] = (on && t.in_y1 && t.in_x0) ? src[o00 + Wv] : 0.f;
      v[k][3] = (on && t.in_y1 && t.in_x1) ? src[o00 + Wv + 1] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int ch = c0 + k;
      if (ch >= n_out) break;
      float s = 0.f;
      if (t.in_y0 && t.in_x0) s += v[k][0] * wnw;
      if (t.in_y0 && t.in_x1) s += v[k][1] * wne;
      if (t.in_y1 && t.in_x0) s += v[k][2] * wsw;
      if (t.in_y1 && t.in_x1) s += v[k][3] * wse;
      if (ch == fill_channel && outside) s = fill_value;
      sample[(size_t)ch * HW + p] = s;
    }
  }
}

// ---- backward ----
// dL/duva needs only per-pixel data. dL/dvirtual_render is a scatter of four taps per output pixel; global fp32 atomics
// sustain only ~26 G/s here (0.7 ms for a 1024^2 grid), so the scatter is turned into a gather by VIRTUAL tile:
//   pixel kernel : per 16x16 output tile, dL/duva and the bounding box of the cells its pixels touch;
//   tile kernel  : one workgroup per virtual tile collects the taps of every output tile whose box meets it (recomputing the
//                  few coordinates involved), then writes its tile with plain stores. Two forms: resample_bwd_tile_kernel
//                  (round 2: ds_add_f32 into an LDS image of the tile) and resample_bwd_gather_kernel (round 4, the default:
//                  pixels bucketed by tap cell with integer atomics, cells gather their buckets; profiles/r04_resample_bwd.txt).
// No global atomics, no memset; every virtual pixel is written exactly once.
constexpr int OT = 16;  // output tile edge (256 pixels = one workgroup)

__device__ inline void lds_add(float* p, float v) {  // ds_add_f32, result unused
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
constexpr int VTX = 64, VTY = 32;  // virtual tile (32 KB of LDS for four channels)

template <bool ATOMIC_SCATTER>
__global__ __launch_bounds__(BLK) void resample_bwd_pixel_kernel(int Hv, int Wv, int H, int W, int n_out,
                                                                 const float* __restrict__ vr, const float* __restrict__ uva,
                                                                 const float* __restrict__ M, int fill_channel,
                                                                 const float* __restrict__ gs, const float* __restrict__ guv,
                                                                 float* __restrict__ gvr, float* __restrict__ guva,
                                                                 int4* __restrict__ bbox) {
  __shared__ int s_box[4][BLK / 64];
  const int tx = threadIdx.x & (OT - 1), ty = threadIdx.x >> 4;
  const int x = blockIdx.x * OT + tx, y = blockIdx.y * OT + ty;
  const bool live = x < W && y < H;
  const int HW = H * W;
  const size_t p = (size_t)y * W + x;
  int bx0 = 0x7FFFFFFF, by0 = 0x7FFFFFFF, bx1 = -0x7FFFFFFF, by1 = -0x7FFFFFFF;
  if (live) {
    const float a = uva[3 * p], b = uva[3 * p + 1], c = uva[3 * p + 2];
    const float u = M[0] * a + M[1] * b + M[2] * c, v = M[3] * a + M[4] * b + M[5] * c;
    const Taps t = make_taps(u, v, Wv, Hv);
    const float wx0 = 1.f - t.wx1, wy0 = 1.f - t.wy1;
    const size_t plane = (size_t)Hv * Wv;
    const size_t o00 = (size_t)t.y0 * Wv + t.x0;
    const bool outside = fabsf(u) > 1.f || fabsf(v) > 1.f;
    float gix = 0.f, giy = 0.f;
    // (gradients and taps of up to four channels requested together, as in the forward; sums in channel order as before)
    for (int c0 = 0; c0 < n_out; c0 += 4) {
      float gq[4], v[4][4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool on = c0 + k < n_out;
        const int ch = on ? c0 + k : 0;
        gq[k] = on ? gs[(size_t)ch * HW + p] : 0.f;
        const float* src = vr + (size_t)ch * plane;
        v[k][0] = (on && t.in_y0 && t.in_x0) ? src[o00] : 0.f;
        v[k][1] = (on && t.in_y0 && t.in_x1) ? src[o00 + 1] : 0.f;
        v[k][2] = (on && t.in_y1 && t.in_x0) ? src[o00 + Wv] : 0.f;
        v[k][3] = (on && t.in_y1 && t.in_x1) ? src[o00 + Wv + 1] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int ch = c0 + k;
        if (ch >= n_out) break;
        float g = gq[k];
        if (ch == fill_channel && outside) g = 0.f;  // the value was overwritten by a constant
        float* dst = gvr + (size_t)ch * plane;
        if (t.in_y0 && t.in_x0) {
          const float val = v[k][0];
          if (ATOMIC_SCATTER) unsafeAtomicAdd(dst + o00, g * wx0 * wy0);
          gix -= val * wy0 * g; giy -= val * wx0 * g;
        }
        if (t.in_y0 && t.in_x1) {
          const float val = v[k][1];
          if (ATOMIC_SCATTER) unsafeAtomicAdd(dst + o00 + 1, g * t.wx1 * wy0);
          gix += val * wy0 * g; giy -= val * t.wx1 * g;
        }
        if (t.in_y1 && t.in_x0) {
          const float val = v[k][2];
          if (ATOMIC_SCATTER) unsafeAtomicAdd(dst + o00 + Wv, g * wx0 * t.wy1);
          gix -= val * t.wy1 * g; giy += val * wx0 * g;
        }
        if (t.in_y1 && t.in_x1) {
          const float val = v[k][3];
          if (ATOMIC_SCATTER) unsafeAtomicAdd(dst + o00 + Wv + 1, g * t.wx1 * t.wy1);
          gix += val * t.wy1 * g; giy += val * t.wx1 * g;
        }
      }
    }
    // d(ix)/du = (Wv-1)/2 (align_corners=True)
    float gu = gix * (0.5f * (float)(Wv - 1)), gv = giy * (0.5f * (float)(Hv - 1));
    if (guv) {
      const float2 e = reinterpret_cast<const float2*>(guv)[p];
      gu += e.x; gv += e.y;
    }
    guva[3 * p] = M[0] * gu + M[3] * gv;
    guva[3 * p + 1] = M[1] * gu + M[4] * gv;
    guva[3 * p + 2] = M[2] * gu + M[5] * gv;
    if ((t.in_x0 || t.in_x1) && (t.in_y0 || t.in_y1)) {  // cells [x0, x0+1] x [y0, y0+1], clamped values
      bx0 = t.x0; bx1 = t.x0 + 1; by0 = t.y0; by1 = t.y0 + 1;
    }
  }
  if (!ATOMIC_SCATTER) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      bx0 = min(bx0, __shfl_xor(bx0, o, 64)); by0 = min(by0, __shfl_xor(by0, o, 64));
      bx1 = max(bx1, __shfl_xor(bx1, o, 64)); by1 = max(by1, __shfl_xor(by1, o, 64));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_box[0][w] = bx0; s_box[1][w] = by0; s_box[2][w] = bx1; s_box[3][w] = by1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int k = 1; k < BLK / 64; k++) {
        bx0 = min(bx0, s_box[0][k]); by0 = min(by0, s_box[1][k]); bx1 = max(bx1, s_box[2][k]); by1 = max(by1, s_box[3][k]);
      }
      bbox[blockIdx.y * gridDim.x + blockIdx.x] = make_int4(bx0, by0, bx1, by1);  // empty: x0 > x1
    }
  }
}

template <int NACC>
__global__ __launch_bounds__(BLK) void resample_bwd_tile_kernel(int C, int Hv, int Wv, int H, int W, int n_out,
                                                                const float* __restrict__ uva, const float* __restrict__ M,
                                                                int fill_channel, const float* __restrict__ gs,
                                                                const int4* __restrict__ bbox, int ntx, int nty,
                                                                float* __restrict__ gvr) {
  __shared__ float s_acc[NACC][VTY][VTX];
  __shared__ uint32_t s_list[1024];
  __shared__ uint32_t s_n;
  const int vx0 = blockIdx.x * VTX, vy0 = blockIdx.y * VTY;
  for (int e = threadIdx.x; e < NACC * VTY * VTX; e += BLK) (&s_acc[0][0][0])[e] = 0.f;
  const int HW = H * W;
  const int nt = ntx * nty;
  for (int scanned = 0; scanned < nt; scanned += 1024) {  // candidate output tiles: 1024 boxes per round
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int lim = min(scanned + 1024, nt);
    for (int t = scanned + threadIdx.x; t < lim; t += BLK) {
      const int4 bb = bbox[t];
      if (bb.x <= vx0 + VTX - 1 && bb.z >= vx0 && bb.y <= vy0 + VTY - 1 && bb.w >= vy0)
        s_list[atomicAdd(&s_n, 1u)] = (uint32_t)t;
    }
    __syncthreads();
    const uint32_t n = s_n;
    for (uint32_t k = 0; k < n; k++) {
      const int t = (int)s_list[k];
      const int x = (t % ntx) * OT + (threadIdx.x & (OT - 1)), y = (t / ntx) * OT + (threadIdx.x >> 4);
      if (x >= W || y >= H) continue;
      const size_t p = (size_t)y * W + x;
      const float a = uva[3 * p], b = uva[3 * p + 1], c = uva[3 * p + 2];
      const float u = M[0] * a + M[1] * b + M[2] * c, v = M[3] * a + M[4] * b + M[5] * c;
      const Taps tp = make_taps(u, v, Wv, Hv);
      const int lx = tp.x0 - vx0, ly = tp.y0 - vy0;
      if (lx < -1 || lx >= VTX || ly < -1 || ly >= VTY) continue;
      const float wx0 = 1.f - tp.wx1, wy0 = 1.f - tp.wy1;
      const bool outside = fabsf(u) > 1.f || fabsf(v) > 1.f;
      const bool cx0 = lx >= 0 && tp.in_x0, cx1 = lx + 1 < VTX && tp.in_x1;
      const bool cy0 = ly >= 0 && tp.in_y0, cy1 = ly + 1 < VTY && tp.in_y1;
#pragma unroll
      for (int ch = 0; ch < NACC; ch++) {
        if (ch >= n_out) break;
        float g = gs[(size_t)ch * HW + p];
        if (ch == fill_channel && outside) g = 0.f;
        if (cy0 && cx0) lds_add(&s_acc[ch][ly][lx], g * wx0 * wy0);
        if (cy0 && cx1) lds_add(&s_acc[ch][ly][lx + 1], g * tp.wx1 * wy0);
        if (cy1 && cx0) lds_add(&s_acc[ch][ly + 1][lx], g * wx0 * tp.wy1);
        if (cy1 && cx1) lds_add(&s_acc[ch][ly + 1][lx + 1], g * tp.wx1 * tp.wy1);
      }
    }
    __syncthreads();
  }
  // every virtual pixel of the tile, every channel (channels >= n_out get zeros): plain coalesced stores
  const size_t plane = (size_t)Hv * Wv;
  for (int e = threadIdx.x; e < VTY * VTX; e += BLK) {
    const int ly = e / VTX, lx = e - ly * VTX;
    const int gx = vx0 + lx, gy = vy0 + ly;
    if (gx < Wv && gy < Hv) {
      for (int ch = 0; ch < C; ch++)
        gvr[ch * plane + (size_t)gy * Wv + gx] = (ch < NACC && ch < n_out) ? s_acc[ch < NACC ? ch : 0][ly][lx] : 0.f;
    }
  }
}

// ---- tile kernel, second form (round 4): bucketed gather, no float atomics ----
// The first form spends its time in ds_add_f32: four taps x n_out channels per output pixel, ~2 LDS cycles per lane each
// (profiles/r04 resample probes: 72-107 us per backward at 1024^2). Here a pixel that lands in the tile costs two INTEGER LDS
// atomics: the workgroup takes its candidate output tiles four at a time (1024 pixels),
//   count : every pixel whose north-west tap cell (x0, y0) lies in [vx0-1, vx0+VTX) x [vy0-1, vy0+VTY) — the cells whose taps can
//           reach the tile — adds one to the counter of that cell's BUCKET;
//   scan  : exclusive prefix of the (VTX+1) x (VTY+1) counters;
//   fill  : the same pixels again, each takes the next place of its bucket and parks {wx1, wy1, g[0..3]} there;
//   gather: every virtual cell (cx, cy) of the tile — a few per thread, accumulators in registers — adds the entries of its four
//           buckets (x0, y0) in {cx-1, cx} x {cy-1, cy} with the tap weight that cell has in them,
// and writes its cells once with plain stores. Every tap of every pixel is counted exactly once, by the tile that owns its
// cell; taps outside the virtual image have no cell. The sums run in a FIXED order (bit-reproducible, unlike the first form and
// unlike grid_sample's backward in the reference): the candidate tiles are listed in tile order (ballot ranks, not an atomic
// cursor), and inside a bucket — whose places the integer atomics hand out in any order — every entry ranks itself among the
// bucket's pixel ids before it parks its payload, so a bucket's entries lie in pixel order.
// (registers: held to the occupancy each variant had before the fixed-order fill — the compiler otherwise keeps a few more
// values live across the two new barriers and drops a wave per SIMD, +50 % on the one-channel kernel)
template <int NACC, int VX, int VY>
__global__ __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu((NACC == 4 && VX == 32) ? 4 : 3)))
void resample_bwd_gather_kernel(int C, int Hv, int Wv, int H, int W, int n_out,
                                                                  const float* __restrict__ uva, const float* __restrict__ M,
                                                                  int fill_channel, const float* __restrict__ gs,
                                                                  const int4* __restrict__ bbox, int ntx, int nty,
                                                                  float* __restrict__ gvr) {
  constexpr int BXN = VX + 1, BYN = VY + 1, NB = BXN * BYN;  // buckets: north-west cells x0 = vx0-1 .. vx0+VX-1
  constexpr int CHT = NACC == 1 ? 8 : 4;                       // candidate output tiles per chunk
  // boxes per scan round, each thread's loads independent. One channel: the whole 1024^2 grid in one round (38.9 us against
  // 46.5 with rounds of 1024 at 2048^2); four channels: rounds of 1024 — the parked entries already take 24 KB of LDS and the
  // larger candidate list costs the fourth resident workgroup (70.9 -> 78.2 us, 31.8 -> 39.1 at 1024^2)
  constexpr int RB = NACC == 1 ? 4096 : 1024;
  constexpr int CAP = CHT * OT * OT;                           // entries a chunk can park
  constexpr int PAY = 2 + NACC;                                // floats per entry: wx1, wy1, g[NACC]
  constexpr int CPT = VX * VY / BLK;                           // cells per thread
  static_assert(VX * VY % BLK == 0 && BLK == OT * OT, "cells dealt evenly; one thread per pixel of an output tile");
  __shared__ uint32_t s_cnt[NB + 1];
  __shared__ uint32_t s_start[NB + 1];
  __shared__ float s_pay[CAP * PAY];
  __shared__ uint16_t s_list[RB];  // candidates of the round, as offsets into it
  __shared__ uint32_t s_n;
  __shared__ uint32_t s_wsum[BLK / 64];
  __shared__ uint32_t s_wc[64];  // candidates per (scan step, wave), then their exclusive prefix
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int vx0 = blockIdx.x * VX, vy0 = blockIdx.y * VY;
  const int HW = H * W;
  const int nt = ntx * nty;
  float acc[CPT][NACC];
#pragma unroll
  for (int c = 0; c < CPT; c++)
#pragma unroll
    for (int ch = 0; ch < NACC; ch++) acc[c][ch] = 0.f;
  for (int e = t; e <= NB; e += BLK) s_cnt[e] = 0u;

  for (int scanned = 0; scanned < nt; scanned += RB) {  // candidate output tiles: RB boxes per round (one round at 1024^2)
    {
      constexpr int RBN = RB / BLK, NW = BLK / 64;
      static_assert(RBN * NW <= 64, "the per-wave candidate counts are scanned by one wave");
      int4 bb[RBN];
#pragma unroll
      for (int i = 0; i < RBN; i++) {
        const int k = scanned + i * BLK + t;
        bb[i] = k < nt ? bbox[k] : make_int4(1, 1, 0, 0);  // (empty box: x0 > x1)
      }
      // the candidates in tile order: per (i, wave) ballot counts, their exclusive scan, ballot ranks
      uint32_t pred = 0;
#pragma unroll
      for (int i = 0; i < RBN; i++) {
        const bool c = bb[i].x <= bb[i].z && bb[i].x <= vx0 + VX - 1 && bb[i].z >= vx0 && bb[i].y <= vy0 + VY - 1 && bb[i].w >= vy0;
        pred |= (c ? 1u : 0u) << i;
        const unsigned long long m = __ballot(c);
        if (lane == 0) s_wc[i * NW + wv] = (uint32_t)__popcll(m);
      }
      __syncthreads();
      if (t < 64) {
        const uint32_t v = t < RBN * NW ? s_wc[t] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t nb2 = __shfl_up(inc, o, 64);
          if (lane >= o) inc += nb2;
        }
        if (t < RBN * NW) s_wc[t] = inc - v;
        if (t == 63) s_n = inc;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < RBN; i++) {
        const bool c = (pred >> i) & 1u;
        const unsigned long long m = __ballot(c);
        if (c) s_list[s_wc[i * NW + wv] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(i * BLK + t);
      }
    }
    __syncthreads();
    const uint32_t n = s_n;
    for (uint32_t k0 = 0; k0 < n; k0 += CHT) {
      const uint32_t kn = min(n - k0, (uint32_t)CHT);
      // This thread's pixel of each of the chunk's candidate tiles. The loads of all CHT pixels are issued before anything
      // waits for one of them (the first version walked the candidates one by one and spent its time in four dependent
      // round trips per pass: 98 us per backward at 2048^2 against 113 for the atomics it had been written to avoid).
      size_t pix[CHT];
      float ua[CHT], ub[CHT], uc[CHT];
      bool in_img[CHT];
#pragma unroll
      for (int k = 0; k < CHT; k++) {
        const int tile = (uint32_t)k < kn ? scanned + (int)s_list[k0 + k] : 0;
        const int x = (tile % ntx) * OT + (t & (OT - 1)), y = (tile / ntx) * OT + (t >> 4);
        in_img[k] = (uint32_t)k < kn && x < W && y < H;
        pix[k] = in_img[k] ? (size_t)y * W + x : 0;
        ua[k] = uva[3 * pix[k]]; ub[k] = uva[3 * pix[k] + 1]; uc[k] = uva[3 * pix[k] + 2];
      }
      int bucket[CHT];
      float wx1[CHT], wy1[CHT];
      bool outside[CHT];
      // count
#pragma unroll
      for (int k = 0; k < CHT; k++) {
        const float u = M[0] * ua[k] + M[1] * ub[k] + M[2] * uc[k], v = M[3] * ua[k] + M[4] * ub[k] + M[5] * uc[k];
        const Taps tp = make_taps(u, v, Wv, Hv);
        const int lx = tp.x0 - vx0, ly = tp.y0 - vy0;
        const bool lands = in_img[k] && lx >= -1 && lx < VX && ly >= -1 && ly < VY;
        bucket[k] = lands ? (ly + 1) * BXN + (lx + 1) : -1;
        wx1[k] = tp.wx1; wy1[k] = tp.wy1;
        outside[k] = fabsf(u) > 1.f || fabsf(v) > 1.f;
        if (lands) atomicAdd(&s_cnt[bucket[k]], 1u);
      }
      // the gradients the fill pass parks: requested now, used after the scan
      float gq[CHT][NACC];
#pragma unroll
      for (int k = 0; k < CHT; k++)
#pragma unroll
        for (int ch = 0; ch < NACC; ch++) {
          float g = (bucket[k] >= 0 && ch < n_out) ? gs[(size_t)ch * HW + pix[k]] : 0.f;
          if (ch == fill_channel && outside[k]) g = 0.f;  // the value was overwritten by a constant
          gq[k][ch] = g;
        }
      __syncthreads();
      // exclusive scan of the NB counters: thread t owns a contiguous run, waves chained through s_wsum
      {
        constexpr int PER = (NB + BLK - 1) / BLK;
        uint32_t loc[PER], sum = 0;
#pragma unroll
        for (int i = 0; i < PER; i++) {
          const int e = t * PER + i;
          loc[i] = e < NB ? s_cnt[e] : 0u;
          sum += loc[i];
        }
        uint32_t inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t nb2 = __shfl_up(inc, o, 64);
          if (lane >= o) inc += nb2;
        }
        if (lane == 63) s_wsum[wv] = inc;
        __syncthreads();
        uint32_t pre = inc - sum;
#pragma unroll
        for (int k = 0; k < BLK / 64; k++)
          if (k < wv) pre += s_wsum[k];
#pragma unroll
        for (int i = 0; i < PER; i++) {
          const int e = t * PER + i;
          if (e < NB) { s_start[e] = pre; s_cnt[e] = 0u; }
          pre += loc[i];
        }
      }
      __syncthreads();
      // fill, in pixel order inside every bucket: the atomics hand out the bucket's places to the pixel IDS (candidate k of the
      // chunk, thread t), each pixel then counts the smaller ids of its bucket (a bucket holds a few entries; one alone skips
      // the loop) and parks its payload at that rank
      uint32_t* s_id = reinterpret_cast<uint32_t*>(s_pay);
#pragma unroll
      for (int k = 0; k < CHT; k++)
        if (bucket[k] >= 0) s_id[(s_start[bucket[k]] + atomicAdd(&s_cnt[bucket[k]], 1u)) * PAY] = (uint32_t)(k * BLK + t);
      __syncthreads();
      uint32_t pos[CHT];
#pragma unroll
      for (int k = 0; k < CHT; k++) {
        pos[k] = 0u;
        if (bucket[k] >= 0) {
          const uint32_t st = s_start[bucket[k]], cnt = s_cnt[bucket[k]], me = (uint32_t)(k * BLK + t);
          uint32_t r = 0u;
          if (cnt > 1u)
            for (uint32_t e = st; e < st + cnt; e++) r += s_id[e * PAY] < me ? 1u : 0u;
          pos[k] = st + r;
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < CHT; k++) {
        if (bucket[k] >= 0) {
          float* e = s_pay + pos[k] * PAY;
          e[0] = wx1[k]; e[1] = wy1[k];
#pragma unroll
          for (int ch = 0; ch < NACC; ch++) e[2 + ch] = gq[k][ch];
        }
      }
      __syncthreads();
      // gather: cell (lx, ly) <- buckets (lx + dx, ly + dy), dx, dy in {0, 1}: bucket column lx holds x0 = cell - 1 (the cell is
      // the EAST tap: weight wx1), column lx + 1 holds x0 = cell (WEST tap: 1 - wx1); rows likewise
#pragma unroll
      for (int c = 0; c < CPT; c++) {
        const int cell = c * BLK + t, lx = cell % VX, ly = cell / VX;
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
#pragma unroll
          for (int dx = 0; dx < 2; dx++) {
            const int bk = (ly + dy) * BXN + lx + dx;
            const uint32_t cnt = s_cnt[bk];
            if (cnt == 0u) continue;
            const uint32_t st = s_start[bk];
            for (uint32_t e = st; e < st + cnt; e++) {
              const float* q = s_pay + e * PAY;
              const float wx = dx ? 1.f - q[0] : q[0], wy = dy ? 1.f - q[1] : q[1];
              const float wgt = wx * wy;
#pragma unroll
              for (int ch = 0; ch < NACC; ch++) acc[c][ch] += q[2 + ch] * wgt;
            }
          }
      }
      __syncthreads();
      for (int e = t; e <= NB; e += BLK) s_cnt[e] = 0u;  // (ordered against the next count pass by the barrier that follows)
      __syncthreads();
    }
  }
  // every virtual pixel of the tile, every channel (channels >= n_out get zeros): plain coalesced stores
  const size_t plane = (size_t)Hv * Wv;
#pragma unroll
  for (int c = 0; c < CPT; c++) {
    const int cell = c * BLK + t, lx = cell % VX, ly = cell / VX;
    const int gx = vx0 + lx, gy = vy0 + ly;
    if (gx < Wv && gy < Hv) {
      for (int ch = 0; ch < C; ch++) {
        float v = 0.f;
#pragma unroll
        for (int a = 0; a < NACC; a++)
          if (a == ch && ch < n_out) v = acc[c][a];
        gvr[ch * plane + (size_t)gy * Wv + gx] = v;
      }
    }
  }
}

}  // namespace

void launch_resample_fwd(int C, int Hv, int Wv, int H, int W, int n_out, const float* vr, const float* uva,
                         const float* M, int fill_channel, float fill_value, float* sample, float* uv, hipStream_t s) {
  const int HW = H * W;
  hipLaunchKernelGGL(resample_fwd_kernel, dim3((HW + BLK - 1) / BLK), dim3(BLK), 0, s, C, Hv, Wv, HW, n_out, vr, uva, M,
                     fill_channel, fill_value, sample, uv);
}

size_t resample_bwd_ws_bytes(int H, int W) {
  const size_t nt = (size_t)((W + OT - 1) / OT) * ((H + OT - 1) / OT);
  return (nt * sizeof(int4) + 255) / 256 * 256 + 256;
}

void launch_resample_bwd(int C, int Hv, int Wv, int H, int W, int n_out, const float* vr, const float* uva,
                         const float* M, int fill_channel, const float* gs, const float* guv, float* gvr, float* guva,
                         void* ws, hipStream_t s) {
  const int ntx = (W + OT - 1) / OT, nty = (H + OT - 1) / OT;
  if (n_out > 4 || !ws) {  // more channels than the LDS tile holds (the reference keeps 4): atomic scatter
    (void)hipMemsetAsync(gvr, 0, (size_t)C * Hv * Wv * sizeof(float), s);
    hipLaunchKernelGGL(resample_bwd_pixel_kernel<true>, dim3(ntx, nty), dim3(BLK), 0, s, Hv, Wv, H, W, n_out, vr, uva, M,
                       fill_channel, gs, guv, gvr, guva, (int4*)nullptr);
    return;
  }
  int4* bbox = reinterpret_cast<int4*>(ws_base(ws));
  hipLaunchKernelGGL(resample_bwd_pixel_kernel<false>, dim3(ntx, nty), dim3(BLK), 0, s, Hv, Wv, H, W, n_out, vr, uva, M,
                     fill_channel, gs, guv, gvr, guva, bbox);
  static const int form = [] {  // EOGS_RESAMPLE_BWD=1: the first form (LDS float atomics); default: the bucketed gather
    const char* e = getenv("EOGS_RESAMPLE_BWD");
    return e ? atoi(e) : 2;
  }();
  if (form == 1) {
    hipLaunchKernelGGL(resample_bwd_tile_kernel<4>, dim3((Wv + VTX - 1) / VTX, (Hv + VTY - 1) / VTY), dim3(BLK), 0, s, C, Hv, Wv,
                       H, W, n_out, uva, M, fill_channel, gs, bbox, ntx, nty, gvr);
    return;
  }
  // virtual tile 64 x 32 for large virtual images, 32 x 32 below 1.5 M cells (a 1024^2 image would be 512 workgroups on 256 CUs)
  const bool big = (size_t)Hv * Wv > 1500000;
  auto* kern = n_out == 1 ? (big ? resample_bwd_gather_kernel<1, 64, 32> : resample_bwd_gather_kernel<1, 32, 32>)
                          : (big ? resample_bwd_gather_kernel<4, 64, 32> : resample_bwd_gather_kernel<4, 32, 32>);
  const int vx = big ? 64 : 32, vy = 32;
  hipLaunchKernelGGL(kern, dim3((Wv + vx - 1) / vx, (Hv + vy - 1) / vy), dim3(BLK), 0, s, C, Hv, Wv, H, W, n_out, uva, M,
                     fill_channel, gs, bbox, ntx, nty, gvr);
}
