// knn.hip — mean squared distance to the three nearest neighbours of every point.
// Reference semantics: src/gaussiansplatting/submodules/simple-knn/simple_knn.cu (Morton order :47-72, boxes of sorted
// points :80-121, conservative box pruning + exact scan :147-185, orchestration :187-222). Init-only (called once).
// Here: bounding box by two reduction kernels (no host sync), 30-bit Morton codes sorted with the library's own stable
// radix passes (binning.hip), points gathered into Morton order once so every later access is coalesced, boxes of 256
// sorted points, one lane per point: +-3 Morton neighbours give the rejection radius, then every box nearer than it is
// scanned. Neighbouring lanes are neighbouring points, so a wave scans nearly the same boxes.
#include "common.h"

namespace {

constexpr int KBOX = 256;       // sorted points per box (= one workgroup)
constexpr float KNN_FAR = 1e37f;  // the reference's FLT_MAX (simple_knn.cu:27)

struct Box { float lo[3], hi[3]; };

__device__ inline float wave_min(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// per-workgroup min/max of the rows [row0, row0+BLK) (given by index through `order` if non-NULL) -> out[blockIdx]
__global__ __launch_bounds__(BLK) void knn_box_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order,
                                                      Box* __restrict__ out) {
  __shared__ float s_v[6][BLK / 64];
  const int i = blockIdx.x * BLK + threadIdx.x;
  float lo[3] = {KNN_FAR, KNN_FAR, KNN_FAR}, hi[3] = {-KNN_FAR, -KNN_FAR, -KNN_FAR};
  if (i < P) {
    const size_t r = order ? order[i] : (uint32_t)i;
#pragma unroll
    for (int k = 0; k < 3; k++) lo[k] = hi[k] = pts[3 * r + k];
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    lo[k] = wave_min(lo[k]);
    hi[k] = wave_max(hi[k]);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) { s_v[k][w] = lo[k]; s_v[3 + k][w] = hi[k]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Box b;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      b.lo[k] = fminf(fminf(s_v[k][0], s_v[k][1]), fminf(s_v[k][2], s_v[k][3]));
      b.hi[k] = fmaxf(fmaxf(s_v[3 + k][0], s_v[3 + k][1]), fmaxf(s_v[3 + k][2], s_v[3 + k][3]));
    }
    out[blockIdx.x] = b;
  }
}

// single workgroup: bounding box of all per-workgroup boxes -> scene[0]
__global__ __launch_bounds__(BLK) void knn_scene_kernel(const Box* __restrict__ boxes, int nb, Box* __restrict__ scene) {
  __shared__ float s_v[6][BLK / 64];
  float lo[3] = {KNN_FAR, KNN_FAR, KNN_FAR}, hi[3] = {-KNN_FAR, -KNN_FAR, -KNN_FAR};
  for (int b = threadIdx.x; b < nb; b += BLK) {
#pragma unroll
    for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], boxes[b].lo[k]); hi[k] = fmaxf(hi[k], boxes[b].hi[k]); }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) { lo[k] = wave_min(lo[k]); hi[k] = wave_max(hi[k]); }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) { s_v[k][w] = lo[k]; s_v[3 + k][w] = hi[k]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Box b;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      b.lo[k] = fminf(fminf(s_v[k][0], s_v[k][1]), fminf(s_v[k][2], s_v[k][3]));
      b.hi[k] = fmaxf(fmaxf(s_v[3 + k][0], s_v[3 + k][1]), fmaxf(s_v[3 + k][2], s_v[3 + k][3]));
    }
    scene[0] = b;
  }
}

__device__ inline uint32_t spread10(uint32_t x) {  // simple_knn.cu:47-54
  x = (x | (x << 16)) & 0x030000FFu;
  x = (x | (x << 8)) & 0x0300F00Fu;
  x = (x | (x << 4)) & 0x030C30C3u;
  x = (x | (x << 2)) & 0x09249249u;
  return x;
}

__global__ __launch_bounds__(BLK) void knn_morton_kernel(int P, const float* __restrict__ pts, const Box* __restrict__ scene,
                                                         uint32_t* __restrict__ codes, uint32_t* __restrict__ ids) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= P) return;
  const Box sc = scene[0];
  uint32_t q[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float ext = sc.hi[k] - sc.lo[k];
    const float t = ext > 0.f ? (pts[3 * (size_t)i + k] - sc.lo[k]) / ext : 0.f;  // a flat axis maps to cell 0
    q[k] = (uint32_t)fminf(fmaxf(t * 1023.f, 0.f), 1023.f);
  }
  codes[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
  ids[i] = (uint32_t)i;
}

__global__ __launch_bounds__(BLK) void knn_gather_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order,
                                                         float4* __restrict__ sorted) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= P) return;
  const size_t r = order[i];
  sorted[i] = make_float4(pts[3 * r], pts[3 * r + 1], pts[3 * r + 2], 0.f);
}

__device__ inline void keep3(float d, float best[3]) {  // updateKBest<3> (simple_knn.cu:133-145)
#pragma unroll
  for (int j = 0; j < 3; j++)
    if (best[j] > d) {
      const float t = best[j];
      best[j] = d;
      d = t;
    }
}
__device__ inline float dist2(const float4& a, const float4& b) {
  const float x = a.x - b.x, y = a.y - b.y, z = a.z - b.z;
  return x * x + y * y + z * z;
}

__global__ __launch_bounds__(BLK) void knn_search_kernel(int P, const float4* __restrict__ sorted,
                                                         const uint32_t* __restrict__ order, const Box* __restrict__ boxes,
                                                         int nb, float* __restrict__ out) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  const bool live = i < P;
  const float4 me = live ? sorted[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  float best[3] = {KNN_FAR, KNN_FAR, KNN_FAR};
  if (live)
    for (int j = max(0, i - 3); j <= min(P - 1, i + 3); j++)
      if (j != i) keep3(dist2(me, sorted[j]), best);
  const float reject = best[2];  // simple_knn.cu:163-166: the bound from the Morton neighbours, then a fresh search
  best[0] = best[1] = best[2] = KNN_FAR;
  for (int b = 0; b < nb; b++) {
    const Box bx = boxes[b];  // wave-uniform address
    float dd = 0.f;
    const float c[3] = {me.x, me.y, me.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float t = c[k] < bx.lo[k] ? bx.lo[k] - c[k] : (c[k] > bx.hi[k] ? c[k] - bx.hi[k] : 0.f);
      dd += t * t;
    }
    const bool want = live && !(dd > reject || dd > best[2]);
    if (__ballot(want) == 0ull) continue;  // no lane of the wave needs this box
    const int j0 = b * KBOX, j1 = min(P, j0 + KBOX);
    for (int j = j0; j < j1; j++) {
      const float4 o = sorted[j];  // wave-uniform address
      if (want && j != i) keep3(dist2(me, o), best);
    }
  }
  if (live) out[order[i]] = (best[0] + best[1] + best[2]) / 3.0f;
}

}  // namespace

KnnWS knn_layout(char* base, int P) {
  KnnWS w;
  size_t n = (size_t)P, o = 0;
  w.nblk = ceil_div_u32(n, BLK * SORTP_ITEMS);
  w.nbox = ceil_div_u32(n, KBOX);
  o = ws_carve(base, o, w.keyA, n);
  o = ws_carve(base, o, w.keyB, n);
  o = ws_carve(base, o, w.valA, n);
  o = ws_carve(base, o, w.valB, n);
  o = ws_carve(base, o, w.hist, (size_t)256 * (w.nblk ? w.nblk : 1));
  o = ws_carve(base, o, w.dtotal, 256);
  o = ws_carve(base, o, w.sorted, n);
  o = ws_carve(base, o, w.boxes, (size_t)(w.nbox + 1) * 6);
  w.bytes = ws_align(o) + 256;
  return w;
}

void launch_knn(const KnnWS& w, int P, const float* pts, float* out, hipStream_t s) {
  const uint32_t nb = w.nbox;
  Box* boxes = reinterpret_cast<Box*>(w.boxes);
  Box* scene = boxes + nb;
  // scene bounding box (unsorted boxes are only a stepping stone for the reduction)
  hipLaunchKernelGGL(knn_box_kernel, dim3(nb), dim3(BLK), 0, s, P, pts, (const uint32_t*)nullptr, boxes);
  hipLaunchKernelGGL(knn_scene_kernel, dim3(1), dim3(BLK), 0, s, boxes, (int)nb, scene);
  hipLaunchKernelGGL(knn_morton_kernel, dim3(nb), dim3(BLK), 0, s, P, pts, scene, w.keyA, w.valA);
  launch_sort_u32(w.keyA, w.valA, w.keyB, w.valB, (uint32_t)P, 4, w.hist, w.nblk, w.dtotal, s);  // 4 passes: back in A
  hipLaunchKernelGGL(knn_gather_kernel, dim3(nb), dim3(BLK), 0, s, P, pts, w.valA, w.sorted);
  hipLaunchKernelGGL(knn_box_kernel, dim3(nb), dim3(BLK), 0, s, P, pts, w.valA, boxes);
  hipLaunchKernelGGL(knn_search_kernel, dim3(nb), dim3(BLK), 0, s, P, w.sorted, w.valA, boxes, (int)nb, out);
}
