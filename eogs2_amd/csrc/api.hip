// api.hip — the extern "C" boundary declared in include/eogs_rast.h (device pointers + hipStream_t).
// Orchestration only: argument checks, workspace carving, kernel launches. The library never allocates
// device memory; the only host allocation is a small pinned staging buffer for the num_rendered readback.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "common.h"

namespace {

thread_local char g_err[512] = "";
thread_local uint32_t* g_pinned = nullptr;  // MISC_WORDS u32, pinned host memory (one per calling thread)
// the forward_prepare whose count readback is still pending on this thread (EOGS_FLAG_DEFER_COUNTS)
struct PendingCounts { bool valid; int P, H, W; bool have_scratch; uint32_t sort_cap; hipStream_t side; bool alt; };
thread_local PendingCounts g_pending_counts = {false, 0, 0, 0, false, 0u, nullptr};
// the side stream of a count copy into g_pinned that nobody has waited for yet (a deferred forward whose counts were never
// asked for): the next forward_prepare drains it before it arms g_pinned again, or the old copy would land as the new counts
thread_local hipStream_t g_copy_in_flight = nullptr;
#define MIRROR_PENDING 0xFFFFFFFFu  // sentinel of a count word that has not arrived (never a legitimate high word of a count below 2^31)

// per calling thread and device: a non-blocking side stream + event for the num_rendered readback
struct Side { int dev; hipStream_t stream; hipEvent_t ev; };
thread_local Side g_side[16];
thread_local int g_nside = 0;
Side* side_for_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  for (int i = 0; i < g_nside; i++)
    if (g_side[i].dev == dev) return &g_side[i];
  if (g_nside == 16) return nullptr;
  Side sd;
  sd.dev = dev;
  if (hipStreamCreateWithFlags(&sd.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&sd.ev, hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(sd.stream); return nullptr; }
  g_side[g_nside] = sd;
  return &g_side[g_nside++];
}

int fail(int code, const char* fmt, const char* detail = "") {
  snprintf(g_err, sizeof g_err, fmt, detail);
  return code;
}

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) return fail(EOGS_ERR_DEVICE, #expr ": %s", hipGetErrorString(e_)); \
  } while (0)

// after a group of launches: always catch launch errors; in debug mode also synchronise (auxiliary.h:178-185)
int check_launch(hipStream_t s, bool debug, const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && debug) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return EOGS_ERR_DEVICE;
  }
  return EOGS_OK;
}
// ---- optional per-kernel-group timing with hipEvents on the launch stream ----
enum { PS_PREPROCESS, PS_DEPTH_SORT, PS_BINNING, PS_RENDER_FWD, PS_RENDER_BWD, PS_GAUSS_BWD, PS_LOSS_FWD, PS_LOSS_BWD, PS_ADAM,
       PS_COMPACT, PS_RESAMPLE_FWD, PS_RESAMPLE_BWD, PS_KNN, PS_SHADE_FWD, PS_SHADE_BWD, PS_MLOSS_FWD, PS_MLOSS_BWD, PS_TSDF, PS_COUNT };
const char* const kSlotNames[PS_COUNT] = {"preprocess_fwd", "depth_sort", "binning", "render_fwd", "render_bwd", "gaussian_bwd",
                                          "loss_fwd", "loss_bwd", "adam", "compact", "resample_fwd", "resample_bwd", "knn",
                                          "shade_fwd", "shade_bwd", "mloss_fwd", "mloss_bwd", "tsdf"};
struct Pending { int slot; hipEvent_t a, b; };
// process-wide (autograd runs backward on its own thread), guarded by g_prof_mu
std::mutex g_prof_mu;
std::atomic<bool> g_prof_on{false};
std::atomic<uint32_t> g_prof_mask{0xFFFFFFFFu};  // bit per slot: which kernel groups are bracketed
double g_prof_ms[PS_COUNT];
int64_t g_prof_n[PS_COUNT];
Pending g_pending[4096];
int g_npending = 0;

// events are pooled: creating / destroying a hipEvent per bracket costs several microseconds of host time each
hipEvent_t g_pool[8192];
int g_npool = 0;
bool prof_get_event(hipEvent_t* e) {  // caller holds g_prof_mu
  if (g_npool > 0) { *e = g_pool[--g_npool]; return true; }
  return hipEventCreate(e) == hipSuccess;
}
void prof_put_event(hipEvent_t e) {  // caller holds g_prof_mu
  if (g_npool < 8192) g_pool[g_npool++] = e;
  else (void)hipEventDestroy(e);
}

void prof_drain() {  // caller holds g_prof_mu
  for (int i = 0; i < g_npending; i++) {
    float ms = 0.f;
    if (hipEventSynchronize(g_pending[i].b) == hipSuccess && hipEventElapsedTime(&ms, g_pending[i].a, g_pending[i].b) == hipSuccess) {
      g_prof_ms[g_pending[i].slot] += ms;
      g_prof_n[g_pending[i].slot] += 1;
    }
    prof_put_event(g_pending[i].a);
    prof_put_event(g_pending[i].b);
  }
  g_npending = 0;
}
// The two events live in the scope object and the {slot, a, b} triple is queued only once both are recorded: forward
// and autograd's backward run on different threads, and a drain triggered by another scope must never see (or recycle)
// a half-recorded pair.
struct ProfScope {
  hipStream_t s; int slot; bool on = false; hipEvent_t a{}, b{};
  ProfScope(int slot_, hipStream_t st) : s(st), slot(slot_) {
    if (!g_prof_on.load(std::memory_order_relaxed) || !((g_prof_mask.load(std::memory_order_relaxed) >> slot) & 1u)) return;
    {
      std::lock_guard<std::mutex> lk(g_prof_mu);
      if (!prof_get_event(&a)) return;
      if (!prof_get_event(&b)) { prof_put_event(a); return; }
    }
    on = true;
    (void)hipEventRecord(a, s);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(b, s);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_npending == 4096) prof_drain();
    g_pending[g_npending++] = Pending{slot, a, b};
  }
};

#define LAUNCH_TRY(s, dbg, what)              \
  do {                                        \
    int rc_ = check_launch((s), (dbg), what); \
    if (rc_ != EOGS_OK) return rc_;           \
  } while (0)

}  // namespace

extern "C" {

const char* eogs_rast_last_error(void) { return g_err; }
int eogs_rast_abi_version(void) { return EOGS_RAST_ABI_VERSION; }
const char* eogs_rast_backend(void) { return "hip-gfx950"; }

int eogs_rast_geom_bytes(int P, size_t* bytes) {
  if (P < 0 || !bytes) return fail(EOGS_ERR_INVALID_ARG, "geom_bytes: bad argument");
  *bytes = geom_layout(nullptr, P).bytes;
  return EOGS_OK;
}
int eogs_rast_image_bytes(int H, int W, size_t* bytes) {
  if (H < 0 || W < 0 || !bytes) return fail(EOGS_ERR_INVALID_ARG, "image_bytes: bad argument");
  *bytes = img_layout(nullptr, H, W).bytes;
  return EOGS_OK;
}
int eogs_rast_binning_bytes(int P, int H, int W, int64_t R, size_t* bytes) {
  (void)P;
  if (R < 0 || H < 0 || W < 0 || !bytes) return fail(EOGS_ERR_INVALID_ARG, "binning_bytes: bad argument");
  *bytes = bin_layout(nullptr, H, W, R).bytes;
  return EOGS_OK;
}
int eogs_rast_scratch_bytes(int P, int H, int W, size_t* bytes) {
  (void)H; (void)W;
  if (P < 0 || !bytes) return fail(EOGS_ERR_INVALID_ARG, "scratch_bytes: bad argument");
  *bytes = sort_layout(nullptr, ent_cap(P)).bytes;
  return EOGS_OK;
}

int eogs_rast_forward_prepare(int P, int H, int W, const float* means3D, const float* scales, const float* rotations,
                              const float* cov3D_precomp, const float* opacities, const float* colors,
                              float scale_modifier, const float* viewmatrix, const float* projmatrix,
                              const float* alt_affine, unsigned flags, int* radii, void* geom,
                              size_t geom_bytes, void* scratch, size_t scratch_bytes, int64_t* num_rendered, void* stream) {
  (void)projmatrix;
  g_err[0] = 0;
  g_pending_counts.valid = false;
  if (P < 0 || H <= 0 || W <= 0 || !num_rendered) return fail(EOGS_ERR_INVALID_ARG, "forward_prepare: bad sizes");
  *num_rendered = 0;
  if (P == 0) return EOGS_OK;
  if (((W + TILE - 1) / TILE) > 32767 || ((H + TILE - 1) / TILE) > 32767)
    return fail(EOGS_ERR_INVALID_ARG, "forward_prepare: image too large for 16-bit internal tile coordinates");
  if ((uint64_t)macro_grid_x(W, BLOCK_BIG) * macro_grid_y(H, BLOCK_BIG) > MAX_BLOCKS)
    return fail(EOGS_ERR_INVALID_ARG, "forward_prepare: image too large (more than 65536 blocks of 32 x 32 pixels)");
  if (!means3D || !opacities || !viewmatrix || !radii || !geom) return fail(EOGS_ERR_INVALID_ARG, "forward_prepare: NULL input");
  if (!colors) return fail(EOGS_ERR_NO_COLORS, "For non-RGB, provide precomputed Gaussian colors!");
  const bool have_sr = scales && rotations, have_cov = cov3D_precomp != nullptr;
  if (have_sr == have_cov || (!!scales != !!rotations))
    return fail(EOGS_ERR_INVALID_ARG,
                "forward_prepare: provide exactly one of either scale/rotation pair or precomputed 3D covariance");
  const bool raw = (flags & EOGS_FLAG_RAW_PARAMS) != 0;
  if (raw && (!have_sr || !alt_affine))
    return fail(EOGS_ERR_INVALID_ARG, "forward_prepare: EOGS_FLAG_RAW_PARAMS needs scales, rotations and alt_affine");
  char* base = ws_base(geom);
  const GeomWS g = geom_layout(base, P);
  if ((size_t)(base - (char*)geom) + g.bytes - 256 > geom_bytes)
    return fail(EOGS_ERR_WORKSPACE, "forward_prepare: geom workspace too small");
  // The entry sort runs here, before the host knows how many entries there are, when the caller hands over a scratch
  // buffer of the size eogs_rast_scratch_bytes() names; otherwise (or when the entries exceed its capacity) it runs in
  // forward_render inside the binning workspace.
  SortWS sw = sort_layout(nullptr, 0);
  bool have_scratch = false;
  if (scratch) {
    char* sb = ws_base(scratch);
    sw = sort_layout(sb, ent_cap(P));
    have_scratch = (size_t)(sb - (char*)scratch) + sw.bytes - 256 <= scratch_bytes;
    if (!have_scratch) return fail(EOGS_ERR_WORKSPACE, "forward_prepare: scratch smaller than eogs_rast_scratch_bytes()");
  }
  hipStream_t s = (hipStream_t)stream;
  const bool debug = flags & EOGS_FLAG_DEBUG;
  const bool readback = !(flags & EOGS_FLAG_NO_READBACK);
  if (!readback && !(flags & EOGS_FLAG_DEFER_COUNTS))
    return fail(EOGS_ERR_INVALID_ARG, "forward_prepare: EOGS_FLAG_NO_READBACK needs EOGS_FLAG_DEFER_COUNTS");
  if (!readback && debug) return fail(EOGS_ERR_INVALID_ARG, "forward_prepare: EOGS_FLAG_DEBUG waits for the stream: not with EOGS_FLAG_NO_READBACK");
  if (readback && !g_pinned) HIP_TRY(hipHostMalloc((void**)&g_pinned, MISC_WORDS * sizeof(uint32_t), hipHostMallocDefault));

  // (g.misc needs no clearing: the scan writes every word the host reads, the error flag included)
  FwdPrepArgs a{P, H, W, means3D, scales, rotations, cov3D_precomp, opacities, colors, viewmatrix, scale_modifier,
                (flags & EOGS_FLAG_ANTIALIASING) != 0, radii, raw, alt_affine};
  { ProfScope ps(PS_PREPROCESS, s); launch_preprocess_fwd(a, g, s); }
  LAUNCH_TRY(s, debug, "preprocess_fwd");
  { ProfScope ps(PS_BINNING, s); launch_pblock_scan(g, P, s); }
  LAUNCH_TRY(s, debug, "pblock_scan");
  // The counts are read back on a side stream that waits only for the two kernels above, so the host wakes up while the
  // caller's stream is busy with the entry sort (which reads the entry count on the device) and has the rest of the
  // forward queued before the GPU runs dry.
  Side* sd = nullptr;
  if (readback) {
    sd = side_for_current_device();
    if (!sd) return fail(EOGS_ERR_DEVICE, "forward_prepare: cannot create the readback stream");
    // The host does not sleep on this copy (hipStreamSynchronize parks the thread after a short spin, and the wake-up — an
    // interrupt plus the scheduler — is on the critical path of every eager forward: a late host is an idle GPU). It arms the
    // pinned words with a sentinel here and polls them in eogs_rast_forward_counts, as a graph's host does with its mirror.
    if (g_copy_in_flight) { HIP_TRY(hipStreamSynchronize(g_copy_in_flight)); g_copy_in_flight = nullptr; }
    for (int i = 0; i < MISC_READBACK; i++) reinterpret_cast<volatile uint32_t*>(g_pinned)[i] = MIRROR_PENDING;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    HIP_TRY(hipEventRecord(sd->ev, s));
    HIP_TRY(hipStreamWaitEvent(sd->stream, sd->ev, 0));
    HIP_TRY(hipMemcpyAsync(g_pinned, g.misc, MISC_READBACK * sizeof(uint32_t), hipMemcpyDeviceToHost, sd->stream));
    g_copy_in_flight = sd->stream;
  }
  if (have_scratch) {
    ProfScope ps(PS_BINNING, s);
    launch_entry_sort(g, sw, P, H, W, s);
  }
  LAUNCH_TRY(s, debug, "entry_sort");
  if (!readback) return EOGS_OK;  // (the caller reads the counts itself: eogs_rast_read_counts)
  g_pending_counts = PendingCounts{true, P, H, W, have_scratch, sw.cap, sd->stream, (flags & EOGS_FLAG_ALT_ONLY) != 0};
  if (flags & EOGS_FLAG_DEFER_COUNTS) return EOGS_OK;  // the caller asks for the token later (eogs_rast_forward_counts)
  return eogs_rast_forward_counts(num_rendered);
}

}  // extern "C"
namespace {
// exact token of a forward from its count words (host copy of GeomWS::misc)
int token_from_counts(const uint32_t* m, int P, int H, int W, bool have_scratch, uint32_t sort_cap, int64_t* num_rendered, bool alt) {
  const uint64_t total = (uint64_t)m[MISC_TOTAL_LO] | ((uint64_t)m[MISC_TOTAL_HI] << 32);
  const uint64_t entries = (uint64_t)m[MISC_MACRO_LO] | ((uint64_t)m[MISC_MACRO_HI] << 32);
  // List granularity the render kernels read (common.h "blocks"). Per-tile lists while footprints are small: every entry a
  // wave reads is one it uses. Lists per 32 x 32-pixel block once a Gaussian is listed in many tiles: the lists are then
  // 4-7x shorter to write and re-read, which can outweigh the block-list scan in the render waves. Since the lists are built
  // per block and split locally (round 3) per-tile lists with the quad kernels win up to at least 13.8 listed tiles per
  // Gaussian (1024^2 trained 0.694 against 0.704 ms, 300 k / 1600^2 0.679 against 0.722); block lists still win at 31
  // (2048^2 trained: 1.536 against 1.596): the switch sits between — profiles/r03_regime_scan.txt.
  static const double block_switch = [] {  // tuning aid: EOGS_BLOCK_SWITCH=<tiles per Gaussian> overrides the default
    const char* e = getenv("EOGS_BLOCK_SWITCH");
    return e ? atof(e) : (double)EOGS_BLOCK_SWITCH;
  }();
  // Second criterion: termination. With opaque Gaussians a pixel stops after ~ln(1e4) / (mean alpha) list entries, so a
  // tile's list (length L = pairs / tiles) is only rendered to a depth ~1 / opacity while binning still writes all of it.
  // While per-tile binning SORTED every pair (rounds 1-2) block lists won when L * (mean pair opacity) exceeded ~115; now a
  // pair costs one 8-byte list item and the criterion loses or ties everywhere it used to fire (2 M Gaussians at opacity
  // 0.3: 1.24 ms per-tile against 1.29; 2 M / 4 M trained: equal), so it is off by default.
  static const double depth_switch = [] {  // EOGS_DEPTH_SWITCH=<L * mean opacity> overrides; 0 disables the criterion
    const char* e = getenv("EOGS_DEPTH_SWITCH");
    return e ? atof(e) : (double)EOGS_DEPTH_SWITCH;
  }();
  const uint64_t opw = (uint64_t)m[MISC_OPW_LO] | ((uint64_t)m[MISC_OPW_HI] << 32);
  const double ntiles8 = (double)macro_grid_x(W, 1) * (double)macro_grid_y(H, 1);
  const double list_depth = (double)opw / 64.0 / ntiles8;  // = L * mean pair opacity
  const bool by_footprint = (double)total > block_switch * (double)P;
  const bool by_depth = depth_switch > 0.0 && list_depth > depth_switch && total >= 2 * entries;  // blocks must merge entries
  // (an altitude-only forward — EOGS_FLAG_ALT_ONLY — runs the quad kernels' one-channel variants: per-tile lists)
  const int block = alt ? 1 : ((by_footprint || by_depth) ? BLOCK_BIG : 1);
  if (m[MISC_ERR] & 1u) return fail(EOGS_ERR_ALTITUDE, "Point is too high: altitude > 200");
  if (total >= ((uint64_t)1 << 31) || entries >= ((uint64_t)1 << 27)) return fail(EOGS_ERR_OVERFLOW, "the forward lists more than the token holds: 2^31 record slots (tile, Gaussian) or 2^27 list "
                                                                                                   "entries (32-px block, Gaussian)");
  // block_lists_kernel's 8-item build pays when the average block holds 2800 ... 6000 entries (csrc/binning.hip)
  const double per_block = (double)entries / ((double)macro_grid_x(W, BLOCK_BIG) * (double)macro_grid_y(H, BLOCK_BIG));
  // (nothing listed: the token is 0, as include/eogs_rast.h says — callers test it whole, every R > 0 shortcut applies)
  // (... except for an altitude-only forward, whose token must still say so: the render launches pick their variant by it)
  *num_rendered = (total == 0 && entries == 0 && !alt) ? 0 : nr_pack((uint32_t)total, (uint32_t)entries, block, have_scratch && entries <= (uint64_t)sort_cap,
                          per_block > 2800.0 && per_block <= 6000.0, list_depth <= (double)GB_WIDE_DEPTH, alt);
  return EOGS_OK;
}


}  // namespace
extern "C" {

int eogs_rast_forward_counts(int64_t* num_rendered) {
  g_err[0] = 0;
  if (!num_rendered) return fail(EOGS_ERR_INVALID_ARG, "forward_counts: NULL argument");
  *num_rendered = 0;
  if (!g_pending_counts.valid) return fail(EOGS_ERR_INVALID_ARG, "forward_counts: no forward_prepare pending on this thread");
  const PendingCounts pc = g_pending_counts;
  g_pending_counts.valid = false;
  const int P = pc.P, H = pc.H, W = pc.W;
  // Poll the armed words (forward_prepare); every 256 looks ask the stream, so a failed copy or a faulted kernel ends the wait
  // with its error instead of spinning for ever. Arrived = every word replaced (eogs_rast_mirror_token's rule) and the tag.
  const volatile uint32_t* v = g_pinned;
  for (uint32_t spin = 1;; spin++) {
    bool arrived = true;
    for (int i = 0; i < MISC_READBACK; i++)
      if (i != MISC_KEY_NMIN && v[i] == MIRROR_PENDING) { arrived = false; break; }
    if (arrived) break;
    if ((spin & 255u) == 0u) {
      const hipError_t q = hipStreamQuery(pc.side);
      if (q == hipSuccess) {  // the copy is done: its words are in memory
        HIP_TRY(hipStreamSynchronize(pc.side));
        break;
      }
      if (q != hipErrorNotReady) HIP_TRY(q);
    }
    __builtin_ia32_pause();
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  g_copy_in_flight = nullptr;
  uint32_t m[MISC_READBACK];
  for (int i = 0; i < MISC_READBACK; i++) m[i] = v[i];
  if (m[MISC_TAG] != MISC_TAG_VALUE) return fail(EOGS_ERR_DEVICE, "forward_counts: the count words did not arrive");
  return token_from_counts(m, P, H, W, pc.have_scratch, pc.sort_cap, num_rendered, pc.alt);
}

int eogs_rast_read_counts(int P, int H, int W, const void* geom, size_t geom_bytes, int have_scratch, void* stream,
                          int64_t* num_rendered) {
  g_err[0] = 0;
  if (P <= 0 || H <= 0 || W <= 0 || !geom || !num_rendered) return fail(EOGS_ERR_INVALID_ARG, "read_counts: bad argument");
  *num_rendered = 0;
  char* base = ws_base(const_cast<void*>(geom));
  const GeomWS g = geom_layout(base, P);
  if ((size_t)(base - (const char*)geom) + g.bytes - 256 > geom_bytes) return fail(EOGS_ERR_WORKSPACE, "read_counts: geom workspace too small");
  uint32_t m[MISC_WORDS];
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemcpyAsync(m, g.misc, MISC_READBACK * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return token_from_counts(m, P, H, W, (have_scratch & 1) != 0, ent_cap(P), num_rendered, (have_scratch & 2) != 0);
}

static_assert(MISC_READBACK * sizeof(uint32_t) <= EOGS_MIRROR_BYTES, "mirror buffer");

int eogs_rast_mirror_arm(void* host) {
  if (!host || ((uintptr_t)host & 63u)) return fail(EOGS_ERR_INVALID_ARG, "mirror_arm: NULL or unaligned host buffer");
  volatile uint32_t* m = (volatile uint32_t*)host;
  for (int i = 0; i < MISC_READBACK; i++) m[i] = MIRROR_PENDING;
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  return EOGS_OK;
}

int eogs_rast_mirror_counts(int P, const void* geom, size_t geom_bytes, void* host, void* stream) {
  g_err[0] = 0;
  if (P <= 0 || !geom || !host || ((uintptr_t)host & 63u)) return fail(EOGS_ERR_INVALID_ARG, "mirror_counts: bad argument");
  char* base = ws_base(const_cast<void*>(geom));
  const GeomWS g = geom_layout(base, P);
  if ((size_t)(base - (const char*)geom) + g.bytes - 256 > geom_bytes) return fail(EOGS_ERR_WORKSPACE, "mirror_counts: geom workspace too small");
  HIP_TRY(hipMemcpyAsync(host, g.misc, MISC_READBACK * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  return EOGS_OK;
}

int eogs_rast_mirror_token(int P, int H, int W, const void* host, int have_scratch, int64_t* num_rendered, int* arrived) {
  g_err[0] = 0;
  if (P <= 0 || H <= 0 || W <= 0 || !host || !num_rendered || !arrived) return fail(EOGS_ERR_INVALID_ARG, "mirror_token: bad argument");
  *arrived = 0;
  const volatile uint32_t* v = (const volatile uint32_t*)host;
  // Arrived = EVERY word of the 40-byte copy has replaced its sentinel, so nothing rests on the copy landing as one
  // transaction. None of them can legitimately hold the sentinel — the high words of counts below 2^31, low words
  // of counts that would otherwise be refused as overflow, a few error bits, MISC_TAG_VALUE, a positive float's bits — except
  // MISC_KEY_NMIN (~min depth key = ~0 for a depth of exactly 0.0f), which sits between two checked words of its 16 bytes.
  for (int i = 0; i < MISC_READBACK; i++)
    if (i != MISC_KEY_NMIN && v[i] == MIRROR_PENDING) return EOGS_OK;
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  uint32_t m[MISC_READBACK];
  for (int i = 0; i < MISC_READBACK; i++) m[i] = v[i];
  if (m[MISC_TAG] != MISC_TAG_VALUE) return fail(EOGS_ERR_INVALID_ARG, "mirror_token: the mirrored words are not a forward's counts");
  *arrived = 1;
  return token_from_counts(m, P, H, W, (have_scratch & 1) != 0, ent_cap(P), num_rendered, (have_scratch & 2) != 0);
}

int eogs_rast_capacity_token(int P, int64_t num_rendered, double slack, int have_scratch, int64_t exact, int64_t* capacity,
                             int* fits) {
  if (P < 0 || num_rendered < 0 || !(slack >= 0.0) || !capacity) return fail(EOGS_ERR_INVALID_ARG, "capacity_token: bad argument");
  uint64_t slots = (uint64_t)((double)nr_slots(num_rendered) * (1.0 + slack)) + 4096u;
  uint64_t ents = (uint64_t)((double)nr_entries(num_rendered) * (1.0 + slack)) + 1024u;
  if (slots > 0x7FFFFFFFull) slots = 0x7FFFFFFFull;
  if (ents > 0x07FFFFFFull) ents = 0x07FFFFFFull;
  // "sorted in scratch" only if every forward that fits this capacity also fits the scratch (then forward_prepare did sort)
  const int sorted = (have_scratch & 1) && ents <= (uint64_t)ent_cap(P);
  const int alt = (have_scratch & 2) != 0 || nr_alt(num_rendered);  // (an altitude-only forward: per-tile lists)
  *capacity = nr_pack((uint32_t)slots, (uint32_t)ents, alt ? 1 : nr_block(num_rendered), sorted, nr_wide(num_rendered),
                      nr_shallow(num_rendered), alt);
  // A capacity token carries the list granularity and the 8-item build of the EARLIER forward: speed only, every choice computes
  // the same values (ABI 3-7 also carried a choice of backward formulation that did matter for parity; gone with ABI 8).
  if (fits) *fits = exact >= 0 && nr_slots(exact) <= slots && nr_entries(exact) <= ents;
  return EOGS_OK;
}

int eogs_rast_forward_render(int P, int H, int W, int64_t R, const float* bg, unsigned flags,
                             void* geom, size_t geom_bytes, void* binning, size_t binning_bytes, void* image,
                             size_t image_bytes, void* scratch, size_t scratch_bytes, float* out_color, float* out_invdepth,
                             void* stream) {
  g_err[0] = 0;
  if (P < 0 || H <= 0 || W <= 0 || R < 0 || !out_color || !bg || !image)
    return fail(EOGS_ERR_INVALID_ARG, "forward_render: bad argument");
  if (R > 0 && (P == 0 || !geom || !binning)) return fail(EOGS_ERR_INVALID_ARG, "forward_render: NULL workspace");
  if (((flags & EOGS_FLAG_ALT_ONLY) != 0) != (nr_alt(R) != 0))
    return fail(EOGS_ERR_INVALID_ARG, "forward_render: EOGS_FLAG_ALT_ONLY and the token disagree (the flag goes to forward_prepare too; "
                                      "tokens built without flags take it as have_scratch | 2)");
  if (nr_alt(R) && nr_block(R) > 1) return fail(EOGS_ERR_INVALID_ARG, "forward_render: malformed altitude-only token");
  hipStream_t s = (hipStream_t)stream;
  const bool debug = flags & EOGS_FLAG_DEBUG;
  char* ibase = ws_base(image);
  const ImgWS im = img_layout(ibase, H, W);
  if ((size_t)(ibase - (char*)image) + im.bytes - 256 > image_bytes)
    return fail(EOGS_ERR_WORKSPACE, "forward_render: image workspace too small");
  GeomWS g = geom_layout(nullptr, 0);
  BinWS b = bin_layout(nullptr, H, W, 0);
  if (P > 0) {
    char* gb = ws_base(geom);
    g = geom_layout(gb, P);
    if ((size_t)(gb - (char*)geom) + g.bytes - 256 > geom_bytes)
      return fail(EOGS_ERR_WORKSPACE, "forward_render: geom workspace too small");
  }
  if (R > 0) {
    char* bb = ws_base(binning);
    b = bin_layout(bb, H, W, R);
    if ((size_t)(bb - (char*)binning) + b.bytes - 256 > binning_bytes)
      return fail(EOGS_ERR_WORKSPACE, "forward_render: binning workspace too small");
  }
  SortWS sw = b.sort;
  if (R > 0 && nr_sorted(R)) {  // forward_prepare sorted the entries in the caller's scratch: the same buffer, untouched since
    if (!scratch) return fail(EOGS_ERR_INVALID_ARG, "forward_render: the scratch buffer handed to forward_prepare is required");
    char* sb = ws_base(scratch);
    sw = sort_layout(sb, ent_cap(P));
    if ((size_t)(sb - (char*)scratch) + sw.bytes - 256 > scratch_bytes)
      return fail(EOGS_ERR_WORKSPACE, "forward_render: scratch smaller than eogs_rast_scratch_bytes()");
  } else if (R > 0 && nr_entries(R)) {
    ProfScope ps(PS_BINNING, s);
    launch_entry_sort(g, sw, P, H, W, s);
  }
  LAUNCH_TRY(s, debug, "entry_sort");
  { ProfScope ps(PS_DEPTH_SORT, s); launch_block_lists(g, sw, b, im, P, H, W, R, s); }
  LAUNCH_TRY(s, debug, "block_lists");
  { ProfScope ps(PS_RENDER_FWD, s); launch_render_fwd(g, b, im, P, H, W, R, bg, out_color, out_invdepth, s); }
  LAUNCH_TRY(s, debug, "render_fwd");
  return EOGS_OK;
}

int eogs_rast_backward_range(int P, int H, int W, int64_t R, const float* bg, const float* means3D, const int* radii,
                             const float* colors, const float* opacities, const float* scales, const float* rotations,
                             float scale_modifier, const float* cov3D_precomp, const float* viewmatrix,
                             const float* projmatrix, const float* alt_affine, unsigned flags, const float* out_color,
                             const float* out_invdepth,
                             const float* dL_dout_color, const float* dL_dout_invdepth, const void* geom, size_t geom_bytes,
                             const void* binning, size_t binning_bytes, const void* image, size_t image_bytes,
                             float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D,
                             float* dL_dscales, float* dL_drotations, float* dL_dT_sum, float* dL_dvm_mean,
                             float* dL_dcolors_lead, int lead_cols, int p_begin, int p_end, void* stream) {
  g_err[0] = 0;
  if (P < 0 || H <= 0 || W <= 0 || R < 0) return fail(EOGS_ERR_INVALID_ARG, "backward: bad sizes");
  if (p_begin < 0 || p_end < p_begin || p_end > P || (p_begin % BLK) != 0 || (p_end != P && (p_end % BLK) != 0))
    return fail(EOGS_ERR_INVALID_ARG, "backward: the Gaussian range must lie in [0, P] with multiples of 256 as inner bounds");
  hipStream_t s = (hipStream_t)stream;
  const bool debug = flags & EOGS_FLAG_DEBUG;
  if (P == 0) {
    if (dL_dT_sum) HIP_TRY(hipMemsetAsync(dL_dT_sum, 0, 6 * sizeof(float), s));
    if (dL_dvm_mean) HIP_TRY(hipMemsetAsync(dL_dvm_mean, 0, 12 * sizeof(float), s));
    return EOGS_OK;
  }
  const bool raw = (flags & EOGS_FLAG_RAW_PARAMS) != 0;
  if (!means3D || !radii || !opacities || !viewmatrix || !projmatrix || !dL_dout_color || !geom ||
      !image || !dL_dmeans2D || !dL_dcolors || !dL_dopacity || !dL_dmeans3D || (!raw && !colors) ||
      (cov3D_precomp && !dL_dcov3D))
    return fail(EOGS_ERR_INVALID_ARG, "backward: NULL argument");
  (void)out_color; (void)out_invdepth;  // (ABI <= 7 read the rendered images back; the back-to-front walk starts from final_T in the image workspace)
  const bool have_sr = scales && rotations;
  if (have_sr == (cov3D_precomp != nullptr)) return fail(EOGS_ERR_INVALID_ARG, "backward: scale/rotation xor cov3D_precomp");
  if (have_sr && (!dL_dscales || !dL_drotations)) return fail(EOGS_ERR_INVALID_ARG, "backward: NULL scale/rotation gradient");
  if (raw && (!have_sr || !alt_affine))
    return fail(EOGS_ERR_INVALID_ARG, "backward: EOGS_FLAG_RAW_PARAMS needs scales, rotations and alt_affine");
  if (R > 0 && !binning) return fail(EOGS_ERR_INVALID_ARG, "backward: NULL binning workspace");
  if (R > 0 && !bg) return fail(EOGS_ERR_INVALID_ARG, "backward: bg is required");
  if (R > 0 && nr_alt(R) && dL_dout_invdepth)
    return fail(EOGS_ERR_INVALID_ARG, "backward: an altitude-only render (EOGS_FLAG_ALT_ONLY) has no inverse-depth output");
  if (R > 0 && nr_alt(R) && nr_block(R) > 1)
    return fail(EOGS_ERR_INVALID_ARG, "backward: malformed altitude-only token");
  if (dL_dcolors_lead && (lead_cols <= 0 || lead_cols > (raw ? 3 : NCH))) return fail(EOGS_ERR_INVALID_ARG, "backward: bad lead_cols");

  char* gb = ws_base(geom);
  const GeomWS g = geom_layout(gb, P);
  char* ib = ws_base(image);
  const ImgWS im = img_layout(ib, H, W);
  BinWS b = bin_layout(nullptr, H, W, 0);
  if (R > 0) {
    char* bb = ws_base(binning);
    b = bin_layout(bb, H, W, R);
    if ((size_t)(bb - (char*)binning) + b.bytes - 256 > binning_bytes)
      return fail(EOGS_ERR_WORKSPACE, "backward: binning workspace too small");
  }
  if ((size_t)(gb - (char*)geom) + g.bytes - 256 > geom_bytes || (size_t)(ib - (char*)image) + im.bytes - 256 > image_bytes)
    return fail(EOGS_ERR_WORKSPACE, "backward: workspace too small");

  if (R > 0 && p_begin == 0) {  // the per-pixel pass covers the whole image: once, with the first range
    { ProfScope ps(PS_RENDER_BWD, s); launch_render_bwd(g, b, im, P, H, W, R, dL_dout_color, dL_dout_invdepth, bg, raw, s); }
    LAUNCH_TRY(s, debug, "render_bwd");
  }
  GaussBwdArgs a{P, H, W, means3D, have_sr ? scales : nullptr, have_sr ? rotations : nullptr, cov3D_precomp, opacities,
                 viewmatrix, projmatrix, radii, scale_modifier, (flags & EOGS_FLAG_ANTIALIASING) != 0,
                 dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, have_sr ? dL_dscales : nullptr,
                 have_sr ? dL_drotations : nullptr, dL_dT_sum, dL_dvm_mean, raw, alt_affine, dL_dcolors_lead,
                 dL_dcolors_lead ? lead_cols : 0, nr_alt(R) != 0, R > 0 ? render_bwd_noflag_ok(b.block, R, P) : 0,
                 R > 0 ? gaussian_bwd_wide(R, P) : 0};
  { ProfScope ps(PS_GAUSS_BWD, s); launch_gaussian_bwd(a, g, b, p_begin, p_end, s); }
  LAUNCH_TRY(s, debug, "gaussian_bwd");
  return EOGS_OK;
}

int eogs_rast_backward(int P, int H, int W, int64_t R, const float* bg, const float* means3D, const int* radii,
                       const float* colors, const float* opacities, const float* scales, const float* rotations,
                       float scale_modifier, const float* cov3D_precomp, const float* viewmatrix,
                       const float* projmatrix, const float* alt_affine, unsigned flags, const float* out_color,
                       const float* out_invdepth,
                       const float* dL_dout_color, const float* dL_dout_invdepth, const void* geom, size_t geom_bytes,
                       const void* binning, size_t binning_bytes, const void* image, size_t image_bytes,
                       float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D,
                       float* dL_dscales, float* dL_drotations, float* dL_dT_sum, float* dL_dvm_mean,
                       float* dL_dcolors_lead, int lead_cols, void* stream) {
  return eogs_rast_backward_range(P, H, W, R, bg, means3D, radii, colors, opacities, scales, rotations, scale_modifier,
                                  cov3D_precomp, viewmatrix, projmatrix, alt_affine, flags, out_color, out_invdepth,
                                  dL_dout_color, dL_dout_invdepth, geom, geom_bytes, binning, binning_bytes, image,
                                  image_bytes, dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dscales,
                                  dL_drotations, dL_dT_sum, dL_dvm_mean, dL_dcolors_lead, lead_cols, 0, P < 0 ? 0 : P, stream);
}

int eogs_rast_path_info(int P, int64_t R, int* list_block_px, int* fwd_kernel, int* bwd_kernel) {
  if (P < 0 || R < 0 || !list_block_px || !fwd_kernel || !bwd_kernel) return fail(EOGS_ERR_INVALID_ARG, "path_info: bad argument");
  const int block = nr_block(R);
  *list_block_px = block * SUBX;
  *fwd_kernel = render_fwd_variant(block, R, P);
  *bwd_kernel = render_bwd_variant(block, R, P);
  return EOGS_OK;
}

int eogs_rast_backward_info(int P, int64_t R, int* gaussian_bwd_wide_out) {
  if (P < 0 || R < 0 || !gaussian_bwd_wide_out) return fail(EOGS_ERR_INVALID_ARG, "backward_info: bad argument");
  *gaussian_bwd_wide_out = R > 0 ? gaussian_bwd_wide(R, P) : 0;
  return EOGS_OK;
}

// checkFrustum's predicate has its culling commented out (DGR/cuda_rasterizer/auxiliary.h:151-176): all visible.
int eogs_rast_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                           uint8_t* present, void* stream) {
  (void)means3D; (void)viewmatrix; (void)projmatrix;
  g_err[0] = 0;
  if (P < 0 || (P > 0 && !present)) return fail(EOGS_ERR_INVALID_ARG, "mark_visible: bad argument");
  if (P > 0) HIP_TRY(hipMemsetAsync(present, 1, (size_t)P, (hipStream_t)stream));
  return EOGS_OK;
}

int eogs_rast_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!on) prof_drain();
  g_prof_on = on != 0;
  return EOGS_OK;
}
int eogs_rast_profile_select(unsigned slot_mask) {
  g_prof_mask = slot_mask;
  return EOGS_OK;
}
int eogs_rast_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  prof_drain();
  for (int i = 0; i < PS_COUNT; i++) { g_prof_ms[i] = 0.0; g_prof_n[i] = 0; }
  return EOGS_OK;
}
int eogs_rast_profile_slots(void) { return PS_COUNT; }
int eogs_rast_profile_get(int slot, double* total_ms, int64_t* launches, const char** name) {
  if (slot < 0 || slot >= PS_COUNT || !total_ms || !launches || !name) return fail(EOGS_ERR_INVALID_ARG, "profile_get: bad argument");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  prof_drain();
  *total_ms = g_prof_ms[slot];
  *launches = g_prof_n[slot];
  *name = kSlotNames[slot];
  return EOGS_OK;
}

int eogs_rast_selftest(void* scratch, unsigned* failed, void* stream) {
  g_err[0] = 0;
  if (!scratch || !failed) return fail(EOGS_ERR_INVALID_ARG, "selftest: bad argument");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(scratch, 0, 4, s));
  launch_selftest((uint32_t*)scratch, s);
  LAUNCH_TRY(s, true, "selftest");
  HIP_TRY(hipMemcpy(failed, scratch, 4, hipMemcpyDeviceToHost));
  return EOGS_OK;
}

// ---- include/eogs_loss.h ----
int eogs_loss_bytes(int planes, int H, int W, unsigned mode, size_t* bytes) {
  if (planes < 0 || H < 0 || W < 0 || !bytes || !(mode & (EOGS_LOSS_L1 | EOGS_LOSS_SSIM)))
    return fail(EOGS_ERR_INVALID_ARG, "loss_bytes: bad argument");
  *bytes = loss_layout(nullptr, planes, H, W, mode).bytes;
  return EOGS_OK;
}

static int loss_check(const char* who, int planes, int H, int W, const void* img, const void* gt, unsigned mode,
                      const void* ws, size_t ws_bytes, LossWS* out) {
  if (planes <= 0 || H <= 0 || W <= 0 || !(mode & (EOGS_LOSS_L1 | EOGS_LOSS_SSIM)) ||
      (mode & ~(EOGS_LOSS_L1 | EOGS_LOSS_SSIM)))
    return fail(EOGS_ERR_INVALID_ARG, "%s: bad sizes or mode", who);
  if (planes > 65535 || (H + 31) / 32 > 65535) return fail(EOGS_ERR_INVALID_ARG, "%s: too many planes / rows for one launch", who);
  if (!img || !gt || !ws) return fail(EOGS_ERR_INVALID_ARG, "%s: NULL argument", who);
  char* base = ws_base(const_cast<void*>(ws));
  *out = loss_layout(base, planes, H, W, mode);
  if ((size_t)(base - (const char*)ws) + out->bytes - 256 > ws_bytes) return fail(EOGS_ERR_WORKSPACE, "%s: workspace too small", who);
  return EOGS_OK;
}

int eogs_loss_forward(int planes, int H, int W, const float* img, const float* gt, unsigned mode, float w_l1,
                      float w_ssim, float bias, float* out, float* plane_sums, void* ws, size_t ws_bytes, void* stream) {
  g_err[0] = 0;
  LossWS w;
  const int rc = loss_check("loss_forward", planes, H, W, img, gt, mode, ws, ws_bytes, &w);
  if (rc != EOGS_OK) return rc;
  if (!out) return fail(EOGS_ERR_INVALID_ARG, "loss_forward: NULL out");
  hipStream_t s = (hipStream_t)stream;
  { ProfScope ps(PS_LOSS_FWD, s); launch_loss_fwd(w, planes, H, W, img, gt, mode, w_l1, w_ssim, bias, out, plane_sums, s); }
  LAUNCH_TRY(s, false, "loss_fwd");
  return EOGS_OK;
}

int eogs_loss_backward(int planes, int H, int W, const float* img, const float* gt, unsigned mode, float w_l1,
                       float w_ssim, const float* upstream, const float* plane_grad, const void* ws, size_t ws_bytes,
                       float* dL_dimg, void* stream) {
  g_err[0] = 0;
  LossWS w;
  const int rc = loss_check("loss_backward", planes, H, W, img, gt, mode, ws, ws_bytes, &w);
  if (rc != EOGS_OK) return rc;
  if (!dL_dimg) return fail(EOGS_ERR_INVALID_ARG, "loss_backward: NULL dL_dimg");
  hipStream_t s = (hipStream_t)stream;
  { ProfScope ps(PS_LOSS_BWD, s); launch_loss_bwd(w, planes, H, W, img, gt, mode, w_l1, w_ssim, upstream, plane_grad, dL_dimg, s); }
  LAUNCH_TRY(s, false, "loss_bwd");
  return EOGS_OK;
}

// ---- include/eogs_optim.h ----
int eogs_adam_step(int n, const eogs_adam_tensor* tensors, double beta1, double beta2, double eps, int64_t step, void* stream) {
  g_err[0] = 0;
  if (n < 0 || n > EOGS_ADAM_MAX_TENSORS || (n > 0 && !tensors) || step < 1)
    return fail(EOGS_ERR_INVALID_ARG, "adam_step: bad argument (at most 16 tensors, step >= 1)");
  for (int i = 0; i < n; i++)
    if (tensors[i].numel < 0 || (tensors[i].numel > 0 && (!tensors[i].param || !tensors[i].grad || !tensors[i].exp_avg ||
                                                          !tensors[i].exp_avg_sq)))
      return fail(EOGS_ERR_INVALID_ARG, "adam_step: NULL tensor");
  hipStream_t s = (hipStream_t)stream;
  int rc;
  { ProfScope ps(PS_ADAM, s); rc = launch_adam(n, tensors, beta1, beta2, eps, step, s); }
  if (rc) return fail(EOGS_ERR_OVERFLOW, "adam_step: too many elements for one launch");
  LAUNCH_TRY(s, false, "adam");
  return EOGS_OK;
}

int eogs_sum_into(int n, const eogs_sum_tensor* tensors, int nsrc, void* stream) {
  g_err[0] = 0;
  if (n < 0 || n > EOGS_SUM_MAX_TENSORS || nsrc < 0 || nsrc > EOGS_SUM_MAX_SOURCES || (n > 0 && !tensors))
    return fail(EOGS_ERR_INVALID_ARG, "sum_into: bad argument (at most 8 tensors with at most 4 sources each)");
  for (int i = 0; i < n; i++) {
    if (tensors[i].numel < 0 || (tensors[i].numel > 0 && !tensors[i].dst)) return fail(EOGS_ERR_INVALID_ARG, "sum_into: NULL tensor");
    for (int k = 0; k < nsrc; k++)
      if (tensors[i].numel > 0 && !tensors[i].src[k]) return fail(EOGS_ERR_INVALID_ARG, "sum_into: NULL source");
  }
  hipStream_t s = (hipStream_t)stream;
  if (launch_sum_into(n, tensors, nsrc, s)) return fail(EOGS_ERR_OVERFLOW, "sum_into: too many elements for one launch");
  LAUNCH_TRY(s, false, "sum_into");
  return EOGS_OK;
}

int eogs_pack_columns(int64_t rows, int n, const eogs_pack_tensor* tensors, float* packed, int packed_cols, int unpack,
                      void* stream) {
  g_err[0] = 0;
  if (rows < 0 || n < 0 || n > EOGS_PACK_MAX_TENSORS || packed_cols < 0 || packed_cols > 16)
    return fail(EOGS_ERR_INVALID_ARG, "pack_columns: bad sizes");
  if (rows == 0 || n == 0) return EOGS_OK;
  if (!tensors || !packed) return fail(EOGS_ERR_INVALID_ARG, "pack_columns: NULL argument");
  int total = 0;
  for (int i = 0; i < n; i++) {
    const eogs_pack_tensor& t = tensors[i];
    if (!t.data || t.width <= 0 || t.col0 < 0 || t.ncols <= 0 || t.col0 + t.ncols > t.width)
      return fail(EOGS_ERR_INVALID_ARG, "pack_columns: bad tensor descriptor");
    total += t.ncols;
  }
  if (total != packed_cols) return fail(EOGS_ERR_INVALID_ARG, "pack_columns: packed_cols is not the sum of the column counts");
  hipStream_t s = (hipStream_t)stream;
  launch_pack_columns(rows, n, tensors, packed, packed_cols, unpack, s);
  LAUNCH_TRY(s, false, "pack_columns");
  return EOGS_OK;
}

int eogs_compact_bytes(int64_t n_rows, size_t* bytes) {
  if (n_rows < 0 || !bytes) return fail(EOGS_ERR_INVALID_ARG, "compact_bytes: bad argument");
  *bytes = compact_layout(nullptr, n_rows).bytes;
  return EOGS_OK;
}

static int compact_check(const char* who, int64_t n_rows, const void* keep, const void* ws, size_t ws_bytes, CompactWS* w) {
  if (n_rows < 0 || n_rows > (int64_t)0x7FFFFFFF * 128) return fail(EOGS_ERR_INVALID_ARG, "%s: bad row count", who);
  if ((n_rows > 0 && !keep) || !ws) return fail(EOGS_ERR_INVALID_ARG, "%s: NULL argument", who);
  char* base = ws_base(const_cast<void*>(ws));
  *w = compact_layout(base, n_rows);
  if ((size_t)(base - (const char*)ws) + w->bytes - 256 > ws_bytes) return fail(EOGS_ERR_WORKSPACE, "%s: workspace too small", who);
  return EOGS_OK;
}

int eogs_compact_plan(int64_t n_rows, const uint8_t* keep, void* ws, size_t ws_bytes, int64_t* n_keep, void* stream) {
  g_err[0] = 0;
  CompactWS w;
  const int rc = compact_check("compact_plan", n_rows, keep, ws, ws_bytes, &w);
  if (rc != EOGS_OK) return rc;
  if (!n_keep) return fail(EOGS_ERR_INVALID_ARG, "compact_plan: NULL n_keep");
  hipStream_t s = (hipStream_t)stream;
  { ProfScope ps(PS_COMPACT, s); launch_compact_plan(w, n_rows, keep, s); }
  LAUNCH_TRY(s, false, "compact_plan");
  uint32_t total = 0;
  HIP_TRY(hipMemcpyAsync(&total, w.blk + w.nblk, sizeof total, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  *n_keep = (int64_t)total;
  return EOGS_OK;
}

int eogs_compact_apply(int64_t n_rows, const uint8_t* keep, int n_tensors, const void* const* src, void* const* dst,
                       const int* row_bytes, const void* ws, size_t ws_bytes, void* stream) {
  g_err[0] = 0;
  CompactWS w;
  const int rc = compact_check("compact_apply", n_rows, keep, ws, ws_bytes, &w);
  if (rc != EOGS_OK) return rc;
  if (n_tensors < 0 || (n_tensors > 0 && (!src || !dst || !row_bytes))) return fail(EOGS_ERR_INVALID_ARG, "compact_apply: bad tensor list");
  for (int t = 0; t < n_tensors; t++)
    if (row_bytes[t] < 0 || row_bytes[t] > 256 || (row_bytes[t] & 3) || (row_bytes[t] > 0 && n_rows > 0 && (!src[t] || !dst[t])))
      return fail(EOGS_ERR_INVALID_ARG, "compact_apply: row sizes must be multiples of 4 up to 256 bytes, pointers non-NULL");
  hipStream_t s = (hipStream_t)stream;
  { ProfScope ps(PS_COMPACT, s); launch_compact_apply(w, n_rows, keep, n_tensors, src, dst, row_bytes, s); }
  LAUNCH_TRY(s, false, "compact_apply");
  return EOGS_OK;
}

// ---- include/eogs_resample.h ----
static int resample_check(const char* who, int C, int Hv, int Wv, int H, int W, int n_out, int fill_channel) {
  if (C <= 0 || Hv <= 0 || Wv <= 0 || H <= 0 || W <= 0 || n_out <= 0 || n_out > C || fill_channel >= n_out ||
      (int64_t)H * W > 0x7FFFFFFF || (int64_t)Hv * Wv > 0x7FFFFFFF)
    return fail(EOGS_ERR_INVALID_ARG, "%s: bad sizes", who);
  return EOGS_OK;
}

int eogs_resample_forward(int C, int Hv, int Wv, int H, int W, int n_out, const float* virtual_render, const float* uva,
                          const float* cam2virt, int fill_channel, float fill_value, float* sample, float* uv,
                          void* stream) {
  g_err[0] = 0;
  const int rc = resample_check("resample_forward", C, Hv, Wv, H, W, n_out, fill_channel);
  if (rc != EOGS_OK) return rc;
  if (!virtual_render || !uva || !cam2virt || !sample || !uv) return fail(EOGS_ERR_INVALID_ARG, "resample_forward: NULL argument");
  hipStream_t s = (hipStream_t)stream;
  { ProfScope ps(PS_RESAMPLE_FWD, s); launch_resample_fwd(C, Hv, Wv, H, W, n_out, virtual_render, uva, cam2virt, fill_channel, fill_value, sample, uv, s); }
  LAUNCH_TRY(s, false, "resample_fwd");
  return EOGS_OK;
}

int eogs_resample_bytes(int H, int W, size_t* bytes) {
  if (H <= 0 || W <= 0 || !bytes) return fail(EOGS_ERR_INVALID_ARG, "resample_bytes: bad argument");
  *bytes = resample_bwd_ws_bytes(H, W);
  return EOGS_OK;
}

int eogs_resample_backward(int C, int Hv, int Wv, int H, int W, int n_out, const float* virtual_render, const float* uva,
                           const float* cam2virt, int fill_channel, const float* dL_dsample, const float* dL_duv,
                           float* dL_dvirtual, float* dL_duva, void* ws, size_t ws_bytes, void* stream) {
  g_err[0] = 0;
  const int rc = resample_check("resample_backward", C, Hv, Wv, H, W, n_out, fill_channel);
  if (rc != EOGS_OK) return rc;
  if (!virtual_render || !uva || !cam2virt || !dL_dsample || !dL_dvirtual || !dL_duva)
    return fail(EOGS_ERR_INVALID_ARG, "resample_backward: NULL argument");
  if (ws && ws_bytes < resample_bwd_ws_bytes(H, W)) return fail(EOGS_ERR_WORKSPACE, "resample_backward: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  { ProfScope ps(PS_RESAMPLE_BWD, s); launch_resample_bwd(C, Hv, Wv, H, W, n_out, virtual_render, uva, cam2virt, fill_channel, dL_dsample, dL_duv, dL_dvirtual, dL_duva, ws, s); }
  LAUNCH_TRY(s, false, "resample_bwd");
  return EOGS_OK;
}

// ---- include/eogs_knn.h ----
int eogs_knn_bytes(int P, size_t* bytes) {
  if (P < 0 || !bytes) return fail(EOGS_ERR_INVALID_ARG, "knn_bytes: bad argument");
  *bytes = knn_layout(nullptr, P).bytes;
  return EOGS_OK;
}

int eogs_knn_mean_dist2(int P, const float* points, float* mean_dist2, void* ws, size_t ws_bytes, void* stream) {
  g_err[0] = 0;
  if (P < 0) return fail(EOGS_ERR_INVALID_ARG, "knn_mean_dist2: bad size");
  if (P == 0) return EOGS_OK;
  if (!points || !mean_dist2 || !ws) return fail(EOGS_ERR_INVALID_ARG, "knn_mean_dist2: NULL argument");
  char* base = ws_base(ws);
  const KnnWS w = knn_layout(base, P);
  if ((size_t)(base - (char*)ws) + w.bytes - 256 > ws_bytes) return fail(EOGS_ERR_WORKSPACE, "knn_mean_dist2: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  { ProfScope ps(PS_KNN, s); launch_knn(w, P, points, mean_dist2, s); }
  LAUNCH_TRY(s, false, "knn");
  return EOGS_OK;
}

// ---- include/eogs_shade.h ----
int eogs_shade_bytes(int H, int W, size_t* bytes) {
  g_err[0] = 0;
  if (H <= 0 || W <= 0 || !bytes) return fail(EOGS_ERR_INVALID_ARG, "shade_bytes: bad argument");
  *bytes = shade_ws_bytes();
  return EOGS_OK;
}

int eogs_shade_forward(int H, int W, const float* raw, const float* alt_diff, const float* M, const float* inshadow,
                       float* cc, float* shaded, float* shadow, void* stream) {
  g_err[0] = 0;
  if (H <= 0 || W <= 0) return fail(EOGS_ERR_INVALID_ARG, "shade_forward: bad sizes");
  if (!raw || !M || !shaded) return fail(EOGS_ERR_INVALID_ARG, "shade_forward: NULL argument");
  if ((alt_diff != nullptr) != (shadow != nullptr) || (alt_diff && !inshadow))
    return fail(EOGS_ERR_INVALID_ARG, "shade_forward: alt_diff, inshadow and shadow go together");
  hipStream_t s = (hipStream_t)stream;
  { ProfScope ps(PS_SHADE_FWD, s); launch_shade_fwd(H, W, raw, alt_diff, M, inshadow, cc, shaded, shadow, s); }
  LAUNCH_TRY(s, false, "shade_fwd");
  return EOGS_OK;
}

int eogs_shade_backward(int H, int W, const float* raw, const float* alt_diff, const float* M, const float* inshadow,
                        const float* g_shaded, const float* g_cc, const float* g_shadow, float* g_raw,
                        float* g_alt_diff, float* g_params, void* ws, size_t ws_bytes, void* stream) {
  g_err[0] = 0;
  if (H <= 0 || W <= 0) return fail(EOGS_ERR_INVALID_ARG, "shade_backward: bad sizes");
  if (!raw || !M || !g_shaded || !g_raw || !g_params || !ws) return fail(EOGS_ERR_INVALID_ARG, "shade_backward: NULL argument");
  if ((alt_diff != nullptr) != (g_alt_diff != nullptr) || (alt_diff && !inshadow) || (!alt_diff && g_shadow))
    return fail(EOGS_ERR_INVALID_ARG, "shade_backward: alt_diff, inshadow and g_alt_diff go together");
  if (ws_bytes < shade_ws_bytes()) return fail(EOGS_ERR_WORKSPACE, "shade_backward: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  { ProfScope ps(PS_SHADE_BWD, s);
    launch_shade_bwd(H, W, raw, alt_diff, M, inshadow, g_shaded, g_cc, g_shadow, g_raw, g_alt_diff, g_params, ws, s); }
  LAUNCH_TRY(s, false, "shade_bwd");
  return EOGS_OK;
}

int eogs_mloss_forward(int H, int W, int mode, const float* alt_diff, const float* rgb_a, const float* rgb_b,
                       const float* uv, float* out, void* ws, size_t ws_bytes, void* stream) {
  g_err[0] = 0;
  if (H <= 0 || W <= 0 || (mode != EOGS_MLOSS_SUN && mode != EOGS_MLOSS_RANDOM))
    return fail(EOGS_ERR_INVALID_ARG, "mloss_forward: bad sizes or mode");
  if (!alt_diff || !rgb_a || !rgb_b || !uv || !out || !ws) return fail(EOGS_ERR_INVALID_ARG, "mloss_forward: NULL argument");
  if (ws_bytes < shade_ws_bytes()) return fail(EOGS_ERR_WORKSPACE, "mloss_forward: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  { ProfScope ps(PS_MLOSS_FWD, s); launch_mloss_fwd(H, W, mode, alt_diff, rgb_a, rgb_b, uv, out, ws, s); }
  LAUNCH_TRY(s, false, "mloss_fwd");
  return EOGS_OK;
}

int eogs_mloss_backward(int H, int W, int mode, const float* alt_diff, const float* rgb_a, const float* rgb_b,
                        const float* uv, const float* out, const float* upstream, float* g_alt_diff, float* g_rgb_a,
                        float* g_rgb_b, void* stream) {
  g_err[0] = 0;
  if (H <= 0 || W <= 0 || (mode != EOGS_MLOSS_SUN && mode != EOGS_MLOSS_RANDOM))
    return fail(EOGS_ERR_INVALID_ARG, "mloss_backward: bad sizes or mode");
  if (!alt_diff || !rgb_a || !rgb_b || !uv || !out || !upstream || !g_alt_diff || !g_rgb_a)
    return fail(EOGS_ERR_INVALID_ARG, "mloss_backward: NULL argument");
  hipStream_t s = (hipStream_t)stream;
  { ProfScope ps(PS_MLOSS_BWD, s);
    launch_mloss_bwd(H, W, mode, alt_diff, rgb_a, rgb_b, uv, out, upstream, g_alt_diff, g_rgb_a, g_rgb_b, s); }
  LAUNCH_TRY(s, false, "mloss_bwd");
  return EOGS_OK;
}

int eogs_tshadow_forward(int64_t n, const float* a, float* out, void* ws, size_t ws_bytes, void* stream) {
  g_err[0] = 0;
  if (n <= 0) return fail(EOGS_ERR_INVALID_ARG, "tshadow_forward: bad size");
  if (!a || !out || !ws) return fail(EOGS_ERR_INVALID_ARG, "tshadow_forward: NULL argument");
  if (ws_bytes < shade_ws_bytes()) return fail(EOGS_ERR_WORKSPACE, "tshadow_forward: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  launch_tshadow_fwd(n, a, out, ws, s);
  LAUNCH_TRY(s, false, "tshadow_fwd");
  return EOGS_OK;
}

int eogs_tshadow_backward(int64_t n, const float* a, const float* upstream, float* g_a, void* stream) {
  g_err[0] = 0;
  if (n <= 0) return fail(EOGS_ERR_INVALID_ARG, "tshadow_backward: bad size");
  if (!a || !upstream || !g_a) return fail(EOGS_ERR_INVALID_ARG, "tshadow_backward: NULL argument");
  hipStream_t s = (hipStream_t)stream;
  launch_tshadow_bwd(n, a, upstream, g_a, s);
  LAUNCH_TRY(s, false, "tshadow_bwd");
  return EOGS_OK;
}

// ---- include/eogs_tsdf.h ----
int eogs_tsdf_integrate(int nx, int ny, int nz, const float* ax, const float* ay, const float* az, const float* affine,
                        float model_scale, float trunc_margin, int H, int W, const float* altitude, const float* weight,
                        float* tsdf_vol, float* weight_vol, void* stream) {
  g_err[0] = 0;
  if (nx < 0 || ny < 0 || nz < 0 || H <= 0 || W <= 0) return fail(EOGS_ERR_INVALID_ARG, "tsdf_integrate: bad sizes");
  if ((size_t)nx * ny * nz == 0) return EOGS_OK;
  if ((uint64_t)nx * ny * nz > ((uint64_t)1 << 40)) return fail(EOGS_ERR_OVERFLOW, "tsdf_integrate: volume too large");
  if (!ax || !ay || !az || !affine || !altitude || !weight || !tsdf_vol || !weight_vol)
    return fail(EOGS_ERR_INVALID_ARG, "tsdf_integrate: NULL argument");
  if (!(model_scale != 0.f) || !(trunc_margin > 0.f)) return fail(EOGS_ERR_INVALID_ARG, "tsdf_integrate: bad scale or truncation");
  hipStream_t s = (hipStream_t)stream;
  { ProfScope ps(PS_TSDF, s);
    launch_tsdf_integrate(nx, ny, nz, ax, ay, az, affine, model_scale, trunc_margin, H, W, altitude, weight, tsdf_vol, weight_vol, s); }
  LAUNCH_TRY(s, false, "tsdf_integrate");
  return EOGS_OK;
}

}  // extern "C"
