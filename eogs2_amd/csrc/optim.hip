// optim.hip — multi-tensor Adam and row compaction for the Gaussian parameter tensors.
// Reference semantics: torch.optim.Adam as configured at src/gaussiansplatting/scene/gaussian_model.py:228-262, and the
// boolean-mask gathers of _prune_optimizer / prune_points (gaussian_model.py:466-505).
// Both are pure HBM streams: Adam reads 16 and writes 12 bytes per element in one launch for all groups; compaction
// reads each kept row once and writes it once, all tensors in one launch, positions from one ballot/prefix scan.
#include "common.h"

namespace {

constexpr int ADAM_VEC = 4;               // elements per thread
constexpr int ADAM_CHUNK = BLK * ADAM_VEC;  // elements per workgroup

struct AdamTable {
  eogs_adam_tensor t[EOGS_ADAM_MAX_TENSORS];
  uint32_t first_block[EOGS_ADAM_MAX_TENSORS + 1];  // workgroup range of each tensor
  int n;
};

__global__ __launch_bounds__(BLK) void adam_kernel(AdamTable tab, float w1, float beta2, float w2, float eps, float inv_bc1,
                                                   float sqrt_bc2) {
  // which tensor does this workgroup belong to (n <= 16: linear search on kernel arguments, wave-uniform)
  int ti = 0;
  while (ti + 1 < tab.n && blockIdx.x >= tab.first_block[ti + 1]) ti++;
  const eogs_adam_tensor T = tab.t[ti];
  const int64_t i0 = ((int64_t)(blockIdx.x - tab.first_block[ti]) * BLK + threadIdx.x) * ADAM_VEC;
  if (i0 >= T.numel) return;
  const float step_size = T.lr * inv_bc1;
  float p[ADAM_VEC], g[ADAM_VEC], m[ADAM_VEC], v[ADAM_VEC];
  const bool full = i0 + ADAM_VEC <= T.numel && ((((uintptr_t)T.param | (uintptr_t)T.grad | (uintptr_t)T.exp_avg |
                                                   (uintptr_t)T.exp_avg_sq) & 15u) == 0);
  if (full) {
    const float4 a = *reinterpret_cast<const float4*>(T.param + i0), b = *reinterpret_cast<const float4*>(T.grad + i0);
    const float4 c = *reinterpret_cast<const float4*>(T.exp_avg + i0), d = *reinterpret_cast<const float4*>(T.exp_avg_sq + i0);
    p[0] = a.x; p[1] = a.y; p[2] = a.z; p[3] = a.w; g[0] = b.x; g[1] = b.y; g[2] = b.z; g[3] = b.w;
    m[0] = c.x; m[1] = c.y; m[2] = c.z; m[3] = c.w; v[0] = d.x; v[1] = d.y; v[2] = d.z; v[3] = d.w;
  } else {
#pragma unroll
    for (int k = 0; k < ADAM_VEC; k++) {
      const bool in = i0 + k < T.numel;
      p[k] = in ? T.param[i0 + k] : 0.f; g[k] = in ? T.grad[i0 + k] : 0.f;
      m[k] = in ? T.exp_avg[i0 + k] : 0.f; v[k] = in ? T.exp_avg_sq[i0 + k] : 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < ADAM_VEC; k++) {
    // torch: exp_avg.lerp_(grad, 1-b1); exp_avg_sq.mul_(b2).addcmul_(grad, grad, 1-b2);
    //        denom = exp_avg_sq.sqrt() / sqrt(bc2) + eps; param.addcdiv_(exp_avg, denom, value=-step_size)
    m[k] = m[k] + w1 * (g[k] - m[k]);
    v[k] = v[k] * beta2 + w2 * (g[k] * g[k]);
    const float denom = sqrtf(v[k]) / sqrt_bc2 + eps;
    p[k] = p[k] - step_size * (m[k] / denom);
  }
  if (full) {
    *reinterpret_cast<float4*>(T.param + i0) = make_float4(p[0], p[1], p[2], p[3]);
    *reinterpret_cast<float4*>(T.exp_avg + i0) = make_float4(m[0], m[1], m[2], m[3]);
    *reinterpret_cast<float4*>(T.exp_avg_sq + i0) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int k = 0; k < ADAM_VEC; k++)
      if (i0 + k < T.numel) {
        T.param[i0 + k] = p[k]; T.exp_avg[i0 + k] = m[k]; T.exp_avg_sq[i0 + k] = v[k];
      }
  }
}

// ---- dst += src0 (+ src1 ...) for several tensors in one launch (include/eogs_optim.h eogs_sum_into) ----
struct SumTable {
  eogs_sum_tensor t[EOGS_SUM_MAX_TENSORS];
  uint32_t first_block[EOGS_SUM_MAX_TENSORS + 1];
  int n, nsrc;
};

__global__ __launch_bounds__(BLK) void sum_into_kernel(SumTable tab) {
  int ti = 0;
  while (ti + 1 < tab.n && blockIdx.x >= tab.first_block[ti + 1]) ti++;
  const eogs_sum_tensor T = tab.t[ti];
  const int64_t i0 = ((int64_t)(blockIdx.x - tab.first_block[ti]) * BLK + threadIdx.x) * ADAM_VEC;
  if (i0 >= T.numel) return;
  uintptr_t al = (uintptr_t)T.dst;
  for (int s = 0; s < tab.nsrc; s++) al |= (uintptr_t)T.src[s];
  if (i0 + ADAM_VEC <= T.numel && (al & 15u) == 0) {
    float4 a = *reinterpret_cast<const float4*>(T.dst + i0);
    for (int s = 0; s < tab.nsrc; s++) {  // (in order: the sum autograd would have made source by source)
      const float4 b = *reinterpret_cast<const float4*>(T.src[s] + i0);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    *reinterpret_cast<float4*>(T.dst + i0) = a;
  } else {
    for (int k = 0; k < ADAM_VEC; k++)
      if (i0 + k < T.numel) {
        float a = T.dst[i0 + k];
        for (int s = 0; s < tab.nsrc; s++) a += T.src[s][i0 + k];
        T.dst[i0 + k] = a;
      }
  }
}

// ---- compaction ----
constexpr int COMPACT_ROWS = BLK;  // rows per workgroup

__global__ __launch_bounds__(BLK) void compact_count_kernel(const uint8_t* __restrict__ keep, int64_t n,
                                                            uint32_t* __restrict__ blk) {
  __shared__ uint32_t s_w[BLK / 64];
  const int64_t i = (int64_t)blockIdx.x * COMPACT_ROWS + threadIdx.x;
  const bool k = i < n && keep[i] != 0;
  const unsigned long long b = __ballot(k);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = (uint32_t)__popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) blk[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// single workgroup: exclusive scan of the per-workgroup counts in place, total appended at blk[nblk]
__global__ __launch_bounds__(BLK) void compact_scan_kernel(uint32_t* __restrict__ blk, uint32_t nblk) {
  __shared__ uint32_t s_w[BLK / 64];
  __shared__ uint32_t s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (uint32_t b0 = 0; b0 < nblk; b0 += BLK) {
    const uint32_t i = b0 + threadIdx.x;
    const uint32_t v = i < nblk ? blk[i] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t nb = __shfl_up(inc, o, 64);
      if (lane >= o) inc += nb;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    const uint32_t pre = (w > 0 ? s_w[0] : 0u) + (w > 1 ? s_w[1] : 0u) + (w > 2 ? s_w[2] : 0u);
    const uint32_t carry = s_carry;
    if (i < nblk) blk[i] = carry + pre + inc - v;
    __syncthreads();
    if (threadIdx.x == BLK - 1) s_carry = carry + pre + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) blk[nblk] = s_carry;
}

struct CompactTable {
  const char* src[EOGS_COMPACT_MAX_TENSORS];
  char* dst[EOGS_COMPACT_MAX_TENSORS];
  int row_words[EOGS_COMPACT_MAX_TENSORS];
  int n;
};

// One workgroup = 256 consecutive rows. The kept rows of the workgroup are consecutive in every destination, so each
// tensor's kept rows are staged in LDS in output order and written as one contiguous run.
__global__ __launch_bounds__(BLK) void compact_apply_kernel(CompactTable tab, const uint8_t* __restrict__ keep, int64_t n,
                                                            const uint32_t* __restrict__ blk) {
  __shared__ uint32_t s_w[BLK / 64];
  __shared__ uint16_t s_srcrow[BLK];  // kept-row rank -> local source row
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * COMPACT_ROWS;
  const int64_t i = row0 + threadIdx.x;
  const bool k = i < n && keep[i] != 0;
  const unsigned long long b = __ballot(k);
  if (lane == 0) s_w[w] = (uint32_t)__popcll(b);
  __syncthreads();
  const uint32_t pre = (w > 0 ? s_w[0] : 0u) + (w > 1 ? s_w[1] : 0u) + (w > 2 ? s_w[2] : 0u);
  const uint32_t kept = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  const uint32_t rank = pre + (uint32_t)__popcll(b & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
  if (k) s_srcrow[rank] = (uint16_t)threadIdx.x;
  __syncthreads();
  if (kept == 0) return;
  const size_t out0 = blk[blockIdx.x];
  for (int t = 0; t < tab.n; t++) {
    const int rw = tab.row_words[t];
    const uint32_t* src = reinterpret_cast<const uint32_t*>(tab.src[t]) + (size_t)row0 * rw;
    uint32_t* dst = reinterpret_cast<uint32_t*>(tab.dst[t]) + out0 * rw;
    const uint32_t words = kept * (uint32_t)rw;
    for (uint32_t e = threadIdx.x; e < words; e += BLK) {  // consecutive lanes -> consecutive output words
      const uint32_t r = e / (uint32_t)rw, c = e - r * (uint32_t)rw;
      dst[e] = src[(size_t)s_srcrow[r] * rw + c];
    }
  }
}

struct PackTable {
  eogs_pack_tensor t[EOGS_PACK_MAX_TENSORS];
  int n;
};

// One workgroup = 256 consecutive rows. A tensor's 256 x width block and the bucket's 256 x K block are both contiguous in
// memory, so every global access is a contiguous run (lane = consecutive word); the column shuffle happens in LDS.
// (One lane per row - 14 scalar stores 56 bytes apart per lane - ran at 0.093 ms for 1 M rows, slower than torch.cat.)
#define PACK_MAX_COLS 16
template <bool UNPACK>
__global__ __launch_bounds__(BLK) void pack_columns_kernel(PackTable tab, int64_t rows, float* __restrict__ packed, int K) {
  __shared__ float s_pk[BLK * PACK_MAX_COLS];
  const int64_t row0 = (int64_t)blockIdx.x * BLK;
  const int nrows = (int)((rows - row0) < (int64_t)BLK ? (rows - row0) : (int64_t)BLK);
  float* pblk = packed + (size_t)row0 * K;
  if (UNPACK) {
    for (int e = threadIdx.x; e < nrows * K; e += BLK) s_pk[e] = pblk[e];
    __syncthreads();
  }
  int o = 0;
  for (int t = 0; t < tab.n; t++) {
    const int wd = tab.t[t].width, c0 = tab.t[t].col0, nc = tab.t[t].ncols;
    float* blk = tab.t[t].data + (size_t)row0 * wd;
    for (int e = threadIdx.x; e < nrows * wd; e += BLK) {
      const int r = e / wd, c = e - r * wd - c0;
      if (c >= 0 && c < nc) {
        if (UNPACK) blk[e] = s_pk[r * K + o + c];
        else s_pk[r * K + o + c] = blk[e];
      }
    }
    o += nc;
  }
  if (!UNPACK) {
    __syncthreads();
    for (int e = threadIdx.x; e < nrows * K; e += BLK) pblk[e] = s_pk[e];
  }
}

}  // namespace

void launch_pack_columns(int64_t rows, int n, const eogs_pack_tensor* tensors, float* packed, int packed_cols, int unpack,
                         hipStream_t s) {
  PackTable tab;
  tab.n = n;
  for (int i = 0; i < n; i++) tab.t[i] = tensors[i];
  const unsigned blocks = (unsigned)((rows + BLK - 1) / BLK);
  if (unpack) hipLaunchKernelGGL(pack_columns_kernel<true>, dim3(blocks), dim3(BLK), 0, s, tab, rows, packed, packed_cols);
  else hipLaunchKernelGGL(pack_columns_kernel<false>, dim3(blocks), dim3(BLK), 0, s, tab, rows, packed, packed_cols);
}

int launch_adam(int n, const eogs_adam_tensor* tensors, double beta1, double beta2, double eps, int64_t step, hipStream_t s) {
  AdamTable tab;
  tab.n = 0;
  uint64_t blocks = 0;
  for (int i = 0; i < n; i++) {
    if (tensors[i].numel <= 0) continue;
    tab.t[tab.n] = tensors[i];
    tab.first_block[tab.n] = (uint32_t)blocks;
    blocks += (uint64_t)((tensors[i].numel + ADAM_CHUNK - 1) / ADAM_CHUNK);
    tab.n++;
  }
  tab.first_block[tab.n] = (uint32_t)blocks;
  if (tab.n == 0) return 0;
  if (blocks > 0x7FFFFFFFull) return -1;
  // bias corrections in double, like Python floats in torch.optim.adam._single_tensor_adam
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3((uint32_t)blocks), dim3(BLK), 0, s, tab, (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), (float)eps, (float)(1.0 / bc1), (float)sqrt(bc2));
  return 0;
}

int launch_sum_into(int n, const eogs_sum_tensor* tensors, int nsrc, hipStream_t s) {
  SumTable tab;
  tab.n = 0;
  tab.nsrc = nsrc;
  uint64_t blocks = 0;
  for (int i = 0; i < n; i++) {
    if (tensors[i].numel <= 0) continue;
    tab.t[tab.n] = tensors[i];
    tab.first_block[tab.n] = (uint32_t)blocks;
    blocks += (uint64_t)((tensors[i].numel + ADAM_CHUNK - 1) / ADAM_CHUNK);
    tab.n++;
  }
  tab.first_block[tab.n] = (uint32_t)blocks;
  if (tab.n == 0 || nsrc == 0) return 0;
  if (blocks > 0x7FFFFFFFull) return -1;
  hipLaunchKernelGGL(sum_into_kernel, dim3((uint32_t)blocks), dim3(BLK), 0, s, tab);
  return 0;
}

CompactWS compact_layout(char* base, int64_t n_rows) {
  CompactWS w;
  w.nblk = (uint32_t)((n_rows + COMPACT_ROWS - 1) / COMPACT_ROWS);
  w.blk = reinterpret_cast<uint32_t*>(base);
  w.bytes = (((size_t)w.nblk + 1) * sizeof(uint32_t) + 255) / 256 * 256 + 256;
  return w;
}

void launch_compact_plan(const CompactWS& w, int64_t n_rows, const uint8_t* keep, hipStream_t s) {
  if (w.nblk) hipLaunchKernelGGL(compact_count_kernel, dim3(w.nblk), dim3(BLK), 0, s, keep, n_rows, w.blk);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(BLK), 0, s, w.blk, w.nblk);
}

void launch_compact_apply(const CompactWS& w, int64_t n_rows, const uint8_t* keep, int n_tensors, const void* const* src,
                          void* const* dst, const int* row_bytes, hipStream_t s) {
  for (int t0 = 0; t0 < n_tensors; t0 += EOGS_COMPACT_MAX_TENSORS) {
    CompactTable tab;
    tab.n = 0;
    for (int t = t0; t < n_tensors && tab.n < EOGS_COMPACT_MAX_TENSORS; t++) {
      if (row_bytes[t] == 0) continue;
      tab.src[tab.n] = (const char*)src[t];
      tab.dst[tab.n] = (char*)dst[t];
      tab.row_words[tab.n] = row_bytes[t] / 4;
      tab.n++;
    }
    if (tab.n && w.nblk) hipLaunchKernelGGL(compact_apply_kernel, dim3(w.nblk), dim3(BLK), 0, s, tab, keep, n_rows, w.blk);
  }
}
