// Tuning aid (round 5): how fast does a wave read 48-byte records when every LANE owns four consecutive records (gaussian_bwd's
// pattern: twelve 16-byte loads per lane, lane stride 192 bytes — every lane of a load instruction in a different cache line)
// against the same bytes read with consecutive lanes on consecutive 16-byte quarters?   hipcc -O3 --offload-arch=gfx950 -o gp gather_pattern.hip && ./gp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int MODE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ src, float* __restrict__ out, size_t nlanes) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nlanes) return;
  float4 v[12];
  if (MODE == 0) {  // lane-owned: records 4 i .. 4 i + 3, quarters 0..2 each
#pragma unroll
    for (int j = 0; j < 12; j++) v[j] = src[i * 12 + j];
  } else {  // coalesced: the wave's 64 x 12 quarters, consecutive lanes on consecutive quarters
    const size_t w = i >> 6, l = i & 63;
#pragma unroll
    for (int j = 0; j < 12; j++) v[j] = src[(w * 12 + j) * 64 + l];
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 12; j++) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  out[i] = s;
}
int main() {
  const size_t nlanes = 1 << 20;  // one lane per Gaussian, four records each: 201 MB
  float4* src; float* out;
  CK(hipMalloc(&src, nlanes * 12 * sizeof(float4)));
  CK(hipMalloc(&out, nlanes * sizeof(float)));
  CK(hipMemset(src, 0, nlanes * 12 * sizeof(float4)));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int mode = 0; mode < 2; mode++) {
    for (int rep = 0; rep < 3; rep++) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nlanes / 256), dim3(256), 0, 0, src, out, nlanes);
      else hipLaunchKernelGGL(k<1>, dim3(nlanes / 256), dim3(256), 0, 0, src, out, nlanes);
    }
    CK(hipEventRecord(a));
    const int N = 20;
    for (int rep = 0; rep < N; rep++) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nlanes / 256), dim3(256), 0, 0, src, out, nlanes);
      else hipLaunchKernelGGL(k<1>, dim3(nlanes / 256), dim3(256), 0, 0, src, out, nlanes);
    }
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%s: %.1f us per launch, %.2f TB/s\n", mode == 0 ? "lane-owned records (stride 192 B)" : "coalesced quarters           ", ms / N * 1e3,
           nlanes * 12.0 * 16 / (ms / N * 1e-3) / 1e12);
  }
  return 0;
}
