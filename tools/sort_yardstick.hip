// Yardstick for the binning stage's sorts (tools only: never linked into the product library).
//
//  (1) rocPRIM's radix_sort_pairs (its one-sweep implementation, ROCm 7.2) on key sets shaped like the pipeline's:
//        depth sort:  P = 1,048,576 keys = float bits of 200 - altitude, payload u32 (Gaussian id), bits [0,24) and [0,32);
//        tile sort:   R = 4,236,528 keys = 14-bit internal-tile ids emitted in depth order, payload 8 bytes {id, slot}.
//      What the product does for the same jobs: DGR/cuda_rasterizer/rasterizer_impl.cu:280,306-311 are the two CUB calls
//      this stage replaces; eogs2_amd/csrc/binning.hip holds the hand-written passes the numbers are compared with.
//  (2) a decoupled look-back chain: nblk workgroups take a ticket, publish 256 per-digit counts as self-tagged 4-byte
//      granules (agent-scope relaxed stores = `sc1`, the per-XCD L2s are not coherent) and resolve their exclusive prefix
//      by walking back over their predecessors' granules. Timed with and without the walk: the difference is what a
//      one-launch radix pass pays for learning its predecessors' digit counts inside the launch.
//
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/sort_yardstick tools/sort_yardstick.hip ; run on the GPU box.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint32_t rnd() {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 32);
}
static inline float rndf() { return (rnd() >> 8) * (1.0f / 16777216.0f); }

template <typename VT>
static double time_rocprim(const uint32_t* d_kin, const VT* d_vin, uint32_t* d_kout, VT* d_vout, size_t n, int b0, int b1,
                           int iters, const std::vector<uint32_t>& h_keys, bool check) {
  size_t tb = 0;
  CK(rocprim::radix_sort_pairs(nullptr, tb, d_kin, d_kout, d_vin, d_vout, n, b0, b1, 0));
  void* tmp; CK(hipMalloc(&tmp, tb));
  for (int i = 0; i < 5; i++) CK(rocprim::radix_sort_pairs(tmp, tb, d_kin, d_kout, d_vin, d_vout, n, b0, b1, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; i++) CK(rocprim::radix_sort_pairs(tmp, tb, d_kin, d_kout, d_vin, d_vout, n, b0, b1, 0));
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  if (check) {
    std::vector<uint32_t> out(n);
    CK(hipMemcpy(out.data(), d_kout, n * 4, hipMemcpyDeviceToHost));
    const uint32_t m = b1 - b0 >= 32 ? 0xFFFFFFFFu : ((1u << (b1 - b0)) - 1u) << b0;
    for (size_t i = 1; i < n; i++) if ((out[i - 1] & m) > (out[i] & m)) { printf("  NOT SORTED at %zu\n", i); break; }
  }
  CK(hipFree(tmp));
  return ms * 1e3 / iters;
}

// ---- decoupled look-back chain ----
#define LB_AGG 0x40000000u
#define LB_INC 0x80000000u
#define LB_VAL 0x3FFFFFFFu
__global__ __launch_bounds__(256) void lookback_kernel(uint32_t* __restrict__ ticket, uint32_t* __restrict__ state,
                                                       uint32_t* __restrict__ out, int walk, int spin_work) {
  __shared__ uint32_t s_tile;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);  // dispatch order is not promised: order = ticket order
  __syncthreads();
  const uint32_t tile = s_tile, t = threadIdx.x;
  uint32_t local = (tile * 131u + t * 7u) & 1023u;  // stands for this workgroup's count of digit t
  // stand-in for the key loads + ranking that precede the publication in a real pass
  for (int i = 0; i < spin_work; i++) local = (local * 1664525u + 1013904223u) & 1023u;
  __hip_atomic_store(&state[(size_t)tile * 256 + t], local | (tile == 0 ? LB_INC : LB_AGG), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t excl = 0;
  if (walk && tile > 0) {
    int p = (int)tile - 1;
    while (true) {
      const uint32_t v = __hip_atomic_load(&state[(size_t)p * 256 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((v >> 30) == 0u) { __builtin_amdgcn_s_sleep(1); continue; }
      excl += v & LB_VAL;
      if (v & LB_INC) break;
      p--;
    }
    __hip_atomic_store(&state[(size_t)tile * 256 + t], ((excl + local) & LB_VAL) | LB_INC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  out[(size_t)tile * 256 + t] = excl;
}

static void run_lookback(int nblk, int spin_work) {
  uint32_t *ticket, *state, *out;
  CK(hipMalloc(&ticket, 256)); CK(hipMalloc(&state, (size_t)nblk * 1024)); CK(hipMalloc(&out, (size_t)nblk * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double us[2];
  for (int walk = 0; walk < 2; walk++) {
    const int iters = 200;
    for (int rep = 0; rep < 2; rep++) {  // rep 0 = warm-up
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; i++) {
        CK(hipMemsetAsync(ticket, 0, 4, 0));
        CK(hipMemsetAsync(state, 0, (size_t)nblk * 1024, 0));
        hipLaunchKernelGGL(lookback_kernel, dim3(nblk), dim3(256), 0, 0, ticket, state, out, walk, spin_work);
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    us[walk] = ms * 1e3 / iters;
  }
  // check the prefixes of the last walked launch
  std::vector<uint32_t> h((size_t)nblk * 256);
  CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (int t = 0; t < 256; t++) {
    uint32_t run = 0;
    for (int b = 0; b < nblk; b++) {
      if (h[(size_t)b * 256 + t] != run) bad++;
      uint32_t local = ((uint32_t)b * 131u + (uint32_t)t * 7u) & 1023u;
      for (int i = 0; i < spin_work; i++) local = (local * 1664525u + 1013904223u) & 1023u;
      run += local;
    }
  }
  printf("lookback nblk=%5d work=%4d: 2 memsets + launch without walk %7.2f us, with walk %7.2f us (walk costs %+6.2f us)  prefixes wrong: %zu\n",
         nblk, spin_work, us[0], us[1], us[1] - us[0], bad);
  CK(hipFree(ticket)); CK(hipFree(state)); CK(hipFree(out));
}

int main(int argc, char** argv) {
  const size_t P = 1048576;
  const int iters = 100;
  // depth keys: altitude uniform in [-17.5, 52.5] (eogs2_amd/synthetic.py: z in [-0.05, 0.15] x 350), depth = 200 - altitude
  std::vector<uint32_t> hk(P), hv(P);
  for (size_t i = 0; i < P; i++) {
    const float d = 200.0f - (-17.5f + 70.0f * rndf());
    memcpy(&hk[i], &d, 4);
    hv[i] = (uint32_t)i;
  }
  uint32_t kmin = ~0u, kmax = 0;
  for (size_t i = 0; i < P; i++) { kmin = std::min(kmin, hk[i]); kmax = std::max(kmax, hk[i]); }
  printf("depth keys: min %08x max %08x (varying bits: %d)\n", kmin, kmax, 32 - __builtin_clz(kmin ^ kmax));
  uint32_t *dk, *dv, *dk2, *dv2;
  CK(hipMalloc(&dk, P * 4)); CK(hipMalloc(&dv, P * 4)); CK(hipMalloc(&dk2, P * 4)); CK(hipMalloc(&dv2, P * 4));
  CK(hipMemcpy(dk, hk.data(), P * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dv, hv.data(), P * 4, hipMemcpyHostToDevice));
  printf("rocPRIM radix_sort_pairs u32+u32, n=%zu, bits [0,24): %8.2f us\n", P, time_rocprim<uint32_t>(dk, dv, dk2, dv2, P, 0, 24, iters, hk, true));
  printf("rocPRIM radix_sort_pairs u32+u32, n=%zu, bits [0,32): %8.2f us\n", P, time_rocprim<uint32_t>(dk, dv, dk2, dv2, P, 0, 32, iters, hk, true));
  printf("rocPRIM radix_sort_pairs u32+u32, n=%zu, bits [0,16): %8.2f us\n", P, time_rocprim<uint32_t>(dk, dv, dk2, dv2, P, 0, 16, iters, hk, true));
  printf("rocPRIM radix_sort_pairs u32+u32, n=%zu, bits [0, 8): %8.2f us\n", P, time_rocprim<uint32_t>(dk, dv, dk2, dv2, P, 0, 8, iters, hk, true));

  // tile keys: every Gaussian lists a 2x2 neighbourhood of the 128x128 internal tiles (4.04 listed tiles per Gaussian)
  const size_t R = 4236528;
  std::vector<uint32_t> tk(R);
  std::vector<uint64_t> tv(R);
  size_t r = 0;
  while (r < R) {
    const uint32_t x = rnd() % 127, y = rnd() % 127, id = rnd() % P;
    for (int q = 0; q < 4 && r < R; q++, r++) {
      tk[r] = (y + (q >> 1)) * 128 + x + (q & 1);
      tv[r] = ((uint64_t)r << 32) | id;
    }
  }
  uint32_t *tk1, *tk2; uint64_t *tv1, *tv2;
  CK(hipMalloc(&tk1, R * 4)); CK(hipMalloc(&tk2, R * 4)); CK(hipMalloc(&tv1, R * 8)); CK(hipMalloc(&tv2, R * 8));
  CK(hipMemcpy(tk1, tk.data(), R * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(tv1, tv.data(), R * 8, hipMemcpyHostToDevice));
  printf("rocPRIM radix_sort_pairs u32+u64, n=%zu, bits [0,14): %8.2f us\n", R, time_rocprim<uint64_t>(tk1, tv1, tk2, tv2, R, 0, 14, iters, tk, true));
  printf("rocPRIM radix_sort_pairs u32+u64, n=%zu, bits [0,16): %8.2f us\n", R, time_rocprim<uint64_t>(tk1, tv1, tk2, tv2, R, 0, 16, iters, tk, true));
  printf("rocPRIM radix_sort_pairs u32+u64, n=%zu, bits [0, 8): %8.2f us\n", R, time_rocprim<uint64_t>(tk1, tv1, tk2, tv2, R, 0, 8, iters, tk, true));
  // the reference's own job: 64-bit keys (tile << 32 | depth bits), 46 bits, u32 payload, on the reference's pair count
  {
    const size_t RR = 6790000;
    std::vector<uint64_t> k64(RR); std::vector<uint32_t> v32(RR);
    for (size_t i = 0; i < RR; i++) { k64[i] = ((uint64_t)(rnd() & 4095u) << 32) | hk[rnd() % P]; v32[i] = rnd() % P; }
    uint64_t *a, *b; uint32_t *c, *d;
    CK(hipMalloc(&a, RR * 8)); CK(hipMalloc(&b, RR * 8)); CK(hipMalloc(&c, RR * 4)); CK(hipMalloc(&d, RR * 4));
    CK(hipMemcpy(a, k64.data(), RR * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(c, v32.data(), RR * 4, hipMemcpyHostToDevice));
    size_t tb = 0;
    CK(rocprim::radix_sort_pairs(nullptr, tb, a, b, c, d, RR, 0, 46, 0));
    void* tmp; CK(hipMalloc(&tmp, tb));
    for (int i = 0; i < 3; i++) CK(rocprim::radix_sort_pairs(tmp, tb, a, b, c, d, RR, 0, 46, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; i++) CK(rocprim::radix_sort_pairs(tmp, tb, a, b, c, d, RR, 0, 46, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("rocPRIM radix_sort_pairs u64+u32, n=%zu, bits [0,46) (the reference's sort at its own pair count): %8.2f us\n", RR, ms * 1e3 / 20);
  }
  for (int nblk : {128, 512, 1024, 2048, 4096})
    for (int work : {0, 256}) run_lookback(nblk, work);
  return 0;
}
