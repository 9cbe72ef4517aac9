// Achievable HBM bandwidth on this MI355X next to the 8 TB/s the roofline is quoted against ("report both
// peak and achievable"): a read-only sum, a copy and a write-only fill over 1 GiB per array (far beyond the 256 MiB of
// last-level cache), float4 per lane, grid-stride, best of 10 after 3 warm-up launches, HIP events.
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_stream.hip -o /tmp/hbm_stream && /tmp/hbm_stream
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, size_t n, float* out) {
  float s = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = a[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) out[0] = s;  // (never true: keeps the loads alive)
}
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ b, size_t n) {
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    b[i] = v;
}

template <class F>
static double best_ms(F launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) launch();
  double best = 1e30;
  for (int i = 0; i < 10; i++) {
    CK(hipEventRecord(e0, 0));
    launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  return best;
}

int main() {
  const size_t bytes = (size_t)1 << 30, n = bytes / sizeof(float4);
  float4 *a, *b;
  float* out;
  CK(hipMalloc((void**)&a, bytes));
  CK(hipMalloc((void**)&b, bytes));
  CK(hipMalloc((void**)&out, 4));
  CK(hipMemset(a, 1, bytes));
  CK(hipMemset(b, 0, bytes));
  for (int wg_per_cu : {4, 8, 16, 32}) {
    const int grid = 256 * wg_per_cu;
    const double r = best_ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, out); });
    const double c = best_ms([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); });
    const double w = best_ms([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n); });
    printf("%2d workgroups per CU: read %.2f TB/s   copy %.2f TB/s (read + write)   write %.2f TB/s\n", wg_per_cu, bytes / r / 1e9,
           2.0 * bytes / c / 1e9, bytes / w / 1e9);
  }
  return 0;
}
