// A C++ host of the C-ABI (include/eogs_rast.h) without torch: the step a maintainer's own C++ trainer would run.
//   1. eager forward + backward the way the reference's forward works (wait for num_rendered, size the workspace);
//   2. the same step recorded with hipStreamBeginCapture — EOGS_FLAG_DEFER_COUNTS | EOGS_FLAG_NO_READBACK, a capacity token,
//      the count mirror — and replayed: bit-identical image and gradients, counts read while the replay runs;
//   3. the replay again after the Gaussians were moved and enlarged in place: the capacity check reports whether the recorded
//      workspaces held the new counts, and the result equals an eager step on the new inputs whenever they did;
//   4. a replay that outgrows the workspaces renders the background and says so.
// Built and run by tests/test_gpu_capi.py (hipcc, linked against eogs2_amd/libeogs_rast_hip.so). Exit code 0 = all checks
// passed; every failed check prints a line.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "eogs_rast.h"

#define HIPCK(x)                                                                         \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                           \
    }                                                                                    \
  } while (0)
#define RCK(x)                                                                              \
  do {                                                                                      \
    int r_ = (x);                                                                           \
    if (r_ != EOGS_OK) {                                                                    \
      fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #x, r_, eogs_rast_last_error()); \
      exit(3);                                                                              \
    }                                                                                       \
  } while (0)

static int g_failed = 0;
#define CHECK(cond, what)                          \
  do {                                             \
    if (!(cond)) {                                 \
      printf("FAILED: %s (%s)\n", what, #cond);    \
      g_failed++;                                  \
    } else {                                       \
      printf("ok: %s\n", what);                    \
    }                                              \
  } while (0)

struct Rng {  // xorshift: the scene only has to be the same in every phase
  uint64_t s;
  float uni() {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (float)((s >> 40) & 0xFFFFFF) / 16777216.0f;
  }
};

template <class T>
struct Dev {
  T* p = nullptr;
  size_t n = 0;
  explicit Dev(size_t n_) : n(n_) { HIPCK(hipMalloc((void**)&p, (n_ ? n_ : 1) * sizeof(T))); }
  ~Dev() { (void)hipFree(p); }
  void up(const std::vector<T>& h) { HIPCK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); }
  std::vector<T> down() const {
    std::vector<T> h(n);
    HIPCK(hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost));
    return h;
  }
};

struct Scene {
  int P, H, W;
  std::vector<float> xyz, scales, rot, opac, colors, vm, bg, dL;
  Scene(int P_, int H_, int W_, uint64_t seed, float scale) : P(P_), H(H_), W(W_) {
    Rng r{seed * 2654435761ull + 88172645463325252ull};
    xyz.resize(3 * (size_t)P); scales.resize(3 * (size_t)P); rot.resize(4 * (size_t)P); opac.resize(P);
    colors.resize(5 * (size_t)P);
    for (int i = 0; i < P; i++) {
      xyz[3 * i] = -0.9f + 1.8f * r.uni(); xyz[3 * i + 1] = -0.9f + 1.8f * r.uni(); xyz[3 * i + 2] = -0.05f + 0.2f * r.uni();
      for (int k = 0; k < 3; k++) scales[3 * i + k] = scale * (0.6f + 0.8f * r.uni());
      float q[4] = {1.f + 0.3f * (r.uni() - 0.5f), 0.3f * (r.uni() - 0.5f), 0.3f * (r.uni() - 0.5f), 0.3f * (r.uni() - 0.5f)};
      const float nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      for (int k = 0; k < 4; k++) rot[4 * i + k] = q[k] / nq;
      opac[i] = 0.05f + 0.9f * r.uni();
      for (int k = 0; k < 3; k++) colors[5 * i + k] = r.uni();
      colors[5 * i + 4] = 1.f;
    }
    // the transposed affine camera [[A^T, 0], [b^T, 1]] (affine_cameras.py:151-157): nadir view, altitude scale 350, a shear
    vm = {0.f, 1.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.05f, -0.03f, 350.f, 0.f, 0.f, 0.f, 0.f, 1.f};
    for (int i = 0; i < P; i++)  // altitude channel = (xyz @ vm[:3, :3] + vm[3, :3])[2]
      colors[5 * i + 3] = xyz[3 * i] * vm[2] + xyz[3 * i + 1] * vm[6] + xyz[3 * i + 2] * vm[10] + vm[14];
    bg = {0.2f, 0.4f, 0.6f, -20.f, 0.f};
    dL.resize(5 * (size_t)H * W);
    for (auto& v : dL) v = (r.uni() - 0.5f) / (float)(H * W);
  }
};

struct Buffers {
  int P, H, W;
  Dev<float> xyz, scales, rot, opac, colors, vm, bg, dL, out_color, out_inv;
  Dev<int> radii;
  Dev<float> g_m2, g_col, g_op, g_m3, g_sc, g_rot;
  Dev<uint8_t> geom, image, scratch;
  size_t geom_b, image_b, scratch_b;
  Buffers(int P_, int H_, int W_, size_t gb, size_t ib, size_t sb)
      : P(P_), H(H_), W(W_), xyz(3 * (size_t)P_), scales(3 * (size_t)P_), rot(4 * (size_t)P_), opac(P_), colors(5 * (size_t)P_), vm(16),
        bg(5), dL(5 * (size_t)H_ * W_), out_color(5 * (size_t)H_ * W_), out_inv((size_t)H_ * W_), radii(P_), g_m2(3 * (size_t)P_),
        g_col(5 * (size_t)P_), g_op(P_), g_m3(3 * (size_t)P_), g_sc(3 * (size_t)P_), g_rot(4 * (size_t)P_), geom(gb), image(ib), scratch(sb),
        geom_b(gb), image_b(ib), scratch_b(sb) {}
  void load(const Scene& s) {
    xyz.up(s.xyz); scales.up(s.scales); rot.up(s.rot); opac.up(s.opac); colors.up(s.colors); vm.up(s.vm); bg.up(s.bg); dL.up(s.dL);
  }
  int prepare(unsigned flags, int64_t* R, hipStream_t st) {
    return eogs_rast_forward_prepare(P, H, W, xyz.p, scales.p, rot.p, nullptr, opac.p, colors.p, 1.0f, vm.p, vm.p, nullptr, flags,
                                     radii.p, geom.p, geom_b, scratch.p, scratch_b, R, st);
  }
  int render(int64_t R, void* binning, size_t binning_b, hipStream_t st) {
    return eogs_rast_forward_render(P, H, W, R, bg.p, 0u, geom.p, geom_b, binning, binning_b, image.p, image_b, scratch.p, scratch_b,
                                    out_color.p, out_inv.p, st);
  }
  int backward(int64_t R, void* binning, size_t binning_b, hipStream_t st) {
    return eogs_rast_backward(P, H, W, R, bg.p, xyz.p, radii.p, colors.p, opac.p, scales.p, rot.p, 1.0f, nullptr, vm.p, vm.p, nullptr,
                              0u, out_color.p, out_inv.p, dL.p, nullptr, geom.p, geom_b, binning, binning_b, image.p, image_b,
                              g_m2.p, g_col.p, g_op.p, g_m3.p, nullptr, g_sc.p, g_rot.p, nullptr, nullptr, nullptr, 0, st);
  }
};

struct Result {
  std::vector<float> color, g_m2, g_col, g_op, g_m3, g_sc, g_rot;
  std::vector<int> radii;
  bool operator==(const Result& o) const {
    auto same = [](const std::vector<float>& a, const std::vector<float>& b) {
      return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0;
    };
    return same(color, o.color) && same(g_m2, o.g_m2) && same(g_col, o.g_col) && same(g_op, o.g_op) && same(g_m3, o.g_m3) &&
           same(g_sc, o.g_sc) && same(g_rot, o.g_rot) && radii == o.radii;
  }
};
static Result fetch(const Buffers& b) {
  return Result{b.out_color.down(), b.g_m2.down(), b.g_col.down(), b.g_op.down(), b.g_m3.down(), b.g_sc.down(), b.g_rot.down(), b.radii.down()};
}

// the reference's shape of a step: wait for the count, size the binning workspace, render, backward
static Result eager_step(Buffers& b, hipStream_t st, int64_t* exact) {
  int64_t R = 0;
  RCK(b.prepare(0u, &R, st));
  size_t nb = 0;
  RCK(eogs_rast_binning_bytes(b.P, b.H, b.W, R, &nb));
  Dev<uint8_t> binning(nb);
  RCK(b.render(R, binning.p, nb, st));
  RCK(b.backward(R, binning.p, nb, st));
  HIPCK(hipStreamSynchronize(st));
  *exact = R;
  return fetch(b);
}

int main() {
  const int P = 40000, H = 224, W = 200;
  printf("ABI %d, backend %s\n", eogs_rast_abi_version(), eogs_rast_backend());
  size_t gb, ib, sb;
  RCK(eogs_rast_geom_bytes(P, &gb));
  RCK(eogs_rast_image_bytes(H, W, &ib));
  RCK(eogs_rast_scratch_bytes(P, H, W, &sb));
  Buffers b(P, H, W, gb, ib, sb);
  hipStream_t st;
  HIPCK(hipStreamCreate(&st));

  const Scene first(P, H, W, 1, 0.004f), moved(P, H, W, 2, 0.0045f), huge(P, H, W, 3, 0.02f);
  b.load(first);
  int64_t exact_first = 0;
  const Result eager_first = eager_step(b, st, &exact_first);
  CHECK(exact_first > 0, "eager step lists something");

  // ---- record the step ----
  int64_t cap = 0;
  RCK(eogs_rast_capacity_token(P, exact_first, 0.25, 1, 0, &cap, nullptr));
  size_t nb = 0;
  RCK(eogs_rast_binning_bytes(P, H, W, cap, &nb));
  Dev<uint8_t> binning(nb);
  void* mirror = nullptr;
  HIPCK(hipHostMalloc(&mirror, EOGS_MIRROR_BYTES, hipHostMallocDefault));
  hipGraph_t graph;
  hipGraphExec_t exec;
  HIPCK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  int64_t zero = -1;
  RCK(b.prepare(EOGS_FLAG_DEFER_COUNTS | EOGS_FLAG_NO_READBACK, &zero, st));
  RCK(eogs_rast_mirror_counts(P, b.geom.p, gb, mirror, st));
  RCK(b.render(cap, binning.p, nb, st));
  RCK(b.backward(cap, binning.p, nb, st));
  HIPCK(hipStreamEndCapture(st, &graph));
  HIPCK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CHECK(zero == 0, "a deferred forward_prepare hands back no token");
  size_t nodes = 0;
  HIPCK(hipGraphGetNodes(graph, nullptr, &nodes));
  printf("recorded %zu graph nodes\n", nodes);

  auto replay = [&](int64_t* exact, int* fits, bool* early) {
    RCK(eogs_rast_mirror_arm(mirror));
    HIPCK(hipGraphLaunch(exec, st));
    int arrived = 0;
    long polls = 0;
    while (!arrived && polls < 200000000L) {  // (the counts arrive a few kernels into the forward)
      RCK(eogs_rast_mirror_token(P, H, W, mirror, 1, exact, &arrived));
      polls++;
    }
    *early = hipStreamQuery(st) == hipErrorNotReady;  // the verdict was there before the replay had finished
    HIPCK(hipStreamSynchronize(st));
    if (!arrived) RCK(eogs_rast_read_counts(P, H, W, b.geom.p, gb, 1, st, exact));
    int64_t c2 = 0;
    RCK(eogs_rast_capacity_token(P, exact_first, 0.25, 1, *exact, &c2, fits));
    return fetch(b);
  };

  int64_t ex = 0;
  int fits = 0;
  bool early = false;
  Result r = replay(&ex, &fits, &early);
  CHECK(ex == exact_first && fits == 1, "replay: mirrored counts equal the eager counts and fit");
  CHECK(r == eager_first, "replay == eager step, bit for bit (image, radii, six gradients)");
  printf("counts known before the replay finished: %s\n", early ? "yes" : "no");

  // ---- new inputs in place ----
  b.load(moved);
  r = replay(&ex, &fits, &early);
  int64_t exact_moved = 0;
  const Result eager_moved = eager_step(b, st, &exact_moved);
  CHECK(ex == exact_moved, "replay on new inputs: mirrored counts equal the eager counts of the new inputs");
  CHECK(fits == 1, "the moved scene fits the recorded workspaces (25 % slack)");
  CHECK(r == eager_moved, "replay on new inputs == eager step on them, bit for bit");
  CHECK(!(eager_moved == eager_first), "(the two scenes do differ)");

  // ---- a replay that outgrows the workspaces ----
  b.load(huge);
  r = replay(&ex, &fits, &early);
  CHECK(fits == 0, "a scene with 5x the footprint does not fit and is reported");
  bool all_bg = true;
  for (int ch = 0; ch < 5 && all_bg; ch++)
    for (size_t i = 0; i < (size_t)H * W; i++)
      if (r.color[(size_t)ch * H * W + i] != huge.bg[ch]) { all_bg = false; break; }
  CHECK(all_bg, "nothing was blended into the undersized workspaces: the image is the background");
  bool zero_grads = true;
  for (const std::vector<float>* v : {&r.g_m2, &r.g_col, &r.g_op, &r.g_m3, &r.g_sc, &r.g_rot})
    for (float x : *v)
      if (x != 0.f) { zero_grads = false; break; }
  CHECK(zero_grads, "and its backward read no record slot beyond them: every gradient is zero");
  int64_t exact_huge = 0;
  const Result eager_huge = eager_step(b, st, &exact_huge);
  CHECK(ex == exact_huge, "the counts it reported are the eager counts");
  b.load(first);
  r = replay(&ex, &fits, &early);
  CHECK(fits == 1 && r == eager_first, "and the graph still replays the first scene bit for bit afterwards");

  // ---- for the record: what a C++ host pays per step either way (not a pass criterion) ----
  {
    const int n = 200;
    hipEvent_t e0, e1;
    HIPCK(hipEventCreate(&e0));
    HIPCK(hipEventCreate(&e1));
    size_t nb_exact = 0;
    RCK(eogs_rast_binning_bytes(P, H, W, exact_first, &nb_exact));
    Dev<uint8_t> bin_exact(nb_exact);
    auto eager_loop = [&](int iters) {
      for (int i = 0; i < iters; i++) {
        int64_t R = 0;
        RCK(b.prepare(0u, &R, st));  // waits for the counts, like the reference's forward
        RCK(b.render(R, bin_exact.p, nb_exact, st));
        RCK(b.backward(R, bin_exact.p, nb_exact, st));
      }
    };
    auto replay_loop = [&](int iters) {
      for (int i = 0; i < iters; i++) {
        RCK(eogs_rast_mirror_arm(mirror));
        HIPCK(hipGraphLaunch(exec, st));
        int arrived = 0;
        int64_t t = 0;
        while (!arrived) RCK(eogs_rast_mirror_token(P, H, W, mirror, 1, &t, &arrived));  // capacity check, every replay
      }
    };
    float ms_eager = 0.f, ms_graph = 0.f;
    eager_loop(20);
    HIPCK(hipStreamSynchronize(st));
    HIPCK(hipEventRecord(e0, st));
    eager_loop(n);
    HIPCK(hipEventRecord(e1, st));
    HIPCK(hipEventSynchronize(e1));
    HIPCK(hipEventElapsedTime(&ms_eager, e0, e1));
    replay_loop(20);
    HIPCK(hipStreamSynchronize(st));
    HIPCK(hipEventRecord(e0, st));
    replay_loop(n);
    HIPCK(hipEventRecord(e1, st));
    HIPCK(hipEventSynchronize(e1));
    HIPCK(hipEventElapsedTime(&ms_graph, e0, e1));
    printf("C++ host, %d Gaussians / %dx%d, fwd+bwd per step: eager (count wait) %.3f ms, graph replay + capacity check %.3f ms\n", P, H, W,
           ms_eager / n, ms_graph / n);
    HIPCK(hipEventDestroy(e0));
    HIPCK(hipEventDestroy(e1));
  }

  HIPCK(hipGraphExecDestroy(exec));
  HIPCK(hipGraphDestroy(graph));
  HIPCK(hipHostFree(mirror));
  HIPCK(hipStreamDestroy(st));
  printf("%s\n", g_failed ? "SOME CHECKS FAILED" : "ALL CHECKS PASSED");
  return g_failed ? 1 : 0;
}
