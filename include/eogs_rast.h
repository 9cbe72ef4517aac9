/*
 * eogs_rast.h — C-ABI of the MI355X-native differentiable Gaussian-splatting
 * rasterizer (drop-in for the hot path of gardiens/EOGS2).
 *
 * Every entry point takes plain pointers + sizes (no torch types) and returns
 * an int status (0 = ok, <0 = error; message via eogs_rast_last_error()).
 * The HIP library (eogs2_amd/csrc -> libeogs_rast_hip.so) takes DEVICE pointers
 * and a hipStream_t passed as `void* stream`.  The CPU oracle
 * (oracle/ -> librast_oracle.so, test infrastructure only) exports the same
 * symbols over HOST pointers and ignores `stream`.
 *
 * Path shorthands for the reference interface each entry point replaces:
 *   DGR/ = src/gaussiansplatting/submodules/diff-gaussian-rasterization/
 *
 * Layout contract (identical to the reference at the API):
 *   means3D   f32[P,3] row-major        scales   f32[P,3]     rotations f32[P,4] (r,x,y,z)
 *   opacities f32[P]                    colors   f32[P,5]     cov3D_precomp f32[P,6] or NULL
 *   viewmatrix / projmatrix f32[16] = the torch tensor's row-major storage, which holds the
 *     TRANSPOSED affine [[A^T,0],[b^T,1]]: uva = xyz @ vm[:3,:3] + vm[3,:3]
 *   bg f32[5]; out_color f32[5,H,W] planar; out_invdepth f32[1,H,W]; radii i32[P]
 * The three workspaces (geom, binning, image) are opaque byte buffers owned by
 * the caller, sized by the *_bytes queries, written by forward and re-read by
 * backward (the reference's geomBuffer / binningBuffer / imgBuffer,
 * DGR/rasterize_points.cu:78-85).  The library never allocates device memory.
 * A fourth buffer, `scratch`, is transient: it holds one forward's list entries while they are sorted
 * (the role of the reference's list_sorting_space, DGR/cuda_rasterizer/rasterizer_impl.cu:186-192) and is
 * dead when eogs_rast_forward_render returns, so one buffer serves every forward on a stream and autograd
 * never keeps it.
 */
#ifndef EOGS_RAST_H_INCLUDED
#define EOGS_RAST_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EOGS_RAST_ABI_VERSION 8 /* 8: every backward walks back to front (the reference's recursion): `bg` is required by eogs_rast_backward,
                                 * out_color / out_invdepth are ignored there, token bit 60 now carries the per-Gaussian kernel hint,
                                 * n_contrib counts positions of the list a render wave walks; eogs_rast_backward_info, eogs_sum_into.
                                 * 7: EOGS_FLAG_ALT_ONLY, token bit 59 (list entries now 27 bits), `have_scratch | 2`, tile schedule in the workspaces */
#define EOGS_RAST_CHANNELS 5 /* DGR/cuda_rasterizer/config.h:15 NUM_CHANNELS */
#define EOGS_RAST_TILE 16    /* DGR/cuda_rasterizer/config.h:16-17 BLOCK_X/BLOCK_Y */

/* status codes */
#define EOGS_OK 0
#define EOGS_ERR_INVALID_ARG (-1) /* bad shape / NULL where data is required            */
#define EOGS_ERR_DEVICE (-2)      /* a HIP call or kernel failed                        */
#define EOGS_ERR_WORKSPACE (-3)   /* workspace smaller than the *_bytes query           */
#define EOGS_ERR_ALTITUDE (-4)    /* a Gaussian's altitude > 200 (DGR/cuda_rasterizer/forward.cu:267-272 traps) */
#define EOGS_ERR_OVERFLOW (-5)    /* a forward lists more than the token holds: 2^31 (tile, Gaussian) record slots or, since ABI 7,
                                   * 2^27 (32-px block, Gaussian) list entries (ABI 6: 2^28) — about 22 M Gaussians of EOGS footprint;
                                   * the reference's own limit is 2^31 (tile, Gaussian) pairs (int num_rendered) */
#define EOGS_ERR_NO_COLORS (-6)   /* colors_precomp missing (DGR/cuda_rasterizer/rasterizer_impl.cu:244-247 throws) */

/* flags */
#define EOGS_FLAG_ANTIALIASING 1u /* raster_settings.antialiasing */
#define EOGS_FLAG_DEBUG 2u        /* raster_settings.debug: sync + check after every kernel (DGR/cuda_rasterizer/auxiliary.h:178-185) */
/* Opt-in (no counterpart in the reference's extension): the per-Gaussian inputs are the model's RAW parameters
 * and the activations + feature assembly that the reference performs in PyTorch before every render are applied
 * inside the per-Gaussian kernels (and chained through in backward):
 *   scales    = log-scales,  s = exp(.)            (src/gaussiansplatting/scene/gaussian_model.py:41,109-111)
 *   rotations = raw quats,   q = r / max(|r|,1e-12) (gaussian_model.py:52,113-115, F.normalize)
 *   opacities = logits,      o = sigmoid(.)        (gaussian_model.py:49,135-137)
 *   colors    = f_dc f32[P,3]; feature = [0.28209479177387814 f_dc + 0.5, altitude, 1] with
 *               altitude = xyz . alt_affine[0:3] + alt_affine[3]
 *               (src/gaussiansplatting/gaussian_renderer/renderer.py:91-96, utils/sh_utils.py:125-126,
 *                scene/cameras/affine_cameras.py:432-438: alt_affine = affine[0:4, 2])
 * cov3D_precomp must be NULL. Backward then returns gradients with respect to the raw parameters:
 * dL_dscales = d/d log-scale, dL_drotations = d/d raw quat, dL_dopacity = d/d logit, dL_dcolors = f32[P,3] d/d f_dc,
 * dL_dmeans3D includes the altitude-feature path; dL_dcov3D may be NULL. */
#define EOGS_FLAG_RAW_PARAMS 4u
/* forward_prepare only: do not wait for the count readback. *num_rendered is left 0; the caller may go on queueing —
 * eogs_rast_forward_render with a CAPACITY token (eogs_rast_capacity_token: workspaces sized from an earlier forward of the
 * same shape) — and must then call eogs_rast_forward_counts, which waits for the readback (by then usually complete) and
 * returns the exact token: if the forward's counts exceed the capacity the device has built NO lists (the image is the
 * background) and the caller repeats forward_render with the exact token. The reference has no counterpart: it blocks
 * on the 4-byte readback inside forward (DGR/cuda_rasterizer/rasterizer_impl.cu:284). */
#define EOGS_FLAG_DEFER_COUNTS 8u
/* forward_prepare only, with EOGS_FLAG_DEFER_COUNTS: no readback at all — the call only queues kernels on `stream`, so it
 * may be recorded into a HIP graph (hipStreamBeginCapture). eogs_rast_forward_counts is not available for such a forward;
 * after the stream's work (or a replay of the graph) has been queued, eogs_rast_read_counts returns its counts. */
#define EOGS_FLAG_NO_READBACK 16u
/* forward_prepare: an ALTITUDE-ONLY render. The reference's training iteration renders the sun camera at 2H x 2W and, by
 * default, consumes that render through its altitude channel alone (GS/train_pan.py:305-324 with
 * gs_config/train.yaml:123 `iterstart_L_sun_resample: 9999999999`; renderer_cc_shadow.py:28-50): the most expensive
 * render of the iteration blends five channels to use one. With this flag the forward blends and stores only feature
 * channel 3: `out_color` of forward_render and `out_color` / `dL_dout_color` of backward are then SINGLE planes f32[H,W]
 * (the altitude image and its gradient), the backward's per-pair work and records shrink accordingly, and dL_dcolors
 * comes back with column 3 alone non-zero (raw mode: a zero f_dc gradient; the altitude's dependence on xyz is chained).
 * The choice travels in the token (bit 59): pass the flag to forward_prepare, hand `have_scratch | 2` to the calls that
 * build a token without flags (read_counts, mirror_token, capacity_token). Such a forward always takes per-tile lists and
 * the quad kernels. Equal to the full render's channel 3 / to a full backward with zero upstream gradient
 * on the other channels (tests/test_gpu_altonly.py). */
#define EOGS_FLAG_ALT_ONLY 32u

/* Thread-local message of the last failing call on this thread ("" if none). */
const char* eogs_rast_last_error(void);
int eogs_rast_abi_version(void);
/* "hip-gfx950" for the product library, "cpu-oracle" for the test oracle. */
const char* eogs_rast_backend(void);

/* Workspace size queries.
 * Replace required<GeometryState/ImageState/BinningState>() —
 * DGR/cuda_rasterizer/rasterizer_impl.h:64-72, rasterizer_impl.cu:155-194,227,240,286. */
int eogs_rast_geom_bytes(int P, size_t* bytes);
int eogs_rast_image_bytes(int H, int W, size_t* bytes);
int eogs_rast_binning_bytes(int P, int H, int W, int64_t num_rendered, size_t* bytes);
/* Size of the transient scratch buffer of forward_prepare / forward_render (about 192 bytes per Gaussian). */
int eogs_rast_scratch_bytes(int P, int H, int W, size_t* bytes);

/* Forward, phase 1: per-Gaussian preprocess + overlap count.
 * Replaces the first half of CudaRasterizer::Rasterizer::forward
 * (DGR/cuda_rasterizer/rasterizer_impl.cu:198-288: FORWARD::preprocess, InclusiveSum,
 * and the blocking 4-byte D2H of num_rendered at :284 — this call synchronises
 * `stream` once for the same reason: the binning workspace size depends on it).
 * Exactly one of (scales, rotations) / cov3D_precomp must be non-NULL. `colors` (colors_precomp, f32[P,5]) is
 * required when P > 0 (EOGS_ERR_NO_COLORS otherwise, DGR/cuda_rasterizer/rasterizer_impl.cu:244-247): the
 * per-Gaussian render record is written once, whole, by the preprocess kernel.
 * alt_affine: f32[4], required with EOGS_FLAG_RAW_PARAMS, otherwise ignored (pass NULL).
 * scratch: eogs_rast_scratch_bytes() bytes, or NULL. With it the list entries are expanded and sorted by this call,
 *   behind the count readback, so the device stays busy while the host sizes the binning workspace; the SAME buffer,
 *   untouched, must then be handed to forward_render. Without it (or when a forward has more entries than the buffer
 *   holds: six per Gaussian) forward_render does that work inside a correspondingly larger binning workspace.
 * Writes radii[P] and *num_rendered (host). num_rendered is an opaque token for the three calls below (it packs this
 * library's pair counts, list granularity and where the entries were sorted, see csrc/common.h nr_pack: record slots in
 * bits 0..30, list entries in bits 32..58, flags above); 0 means nothing is listed.
 * One of the flag bits (60) says whether the forward's lists are shallow in opacity (mean list depth x mean pair opacity):
 * eogs_rast_backward picks between builds of its per-Gaussian kernel by it (a capacity token inherits it). A hint only: every
 * build computes the same bits. */
int eogs_rast_forward_prepare(
    int P, int H, int W,
    const float* means3D, const float* scales, const float* rotations,
    const float* cov3D_precomp, const float* opacities, const float* colors, float scale_modifier,
    const float* viewmatrix, const float* projmatrix, const float* alt_affine, unsigned flags,
    int* radii, void* geom, size_t geom_bytes, void* scratch, size_t scratch_bytes,
    int64_t* num_rendered, void* stream);

/* Exact token of the last eogs_rast_forward_prepare(..., EOGS_FLAG_DEFER_COUNTS) on this thread: waits for its count
 * readback — by polling the pinned words the copy lands in (the calling thread spins, it is never parked; the stream is
 * asked every 256 looks, so a failed copy or a faulted kernel ends the wait with EOGS_ERR_DEVICE). A forward_prepare
 * WITHOUT the flag ends in this same wait. Errors as forward_prepare reports them (EOGS_ERR_ALTITUDE, EOGS_ERR_OVERFLOW). */
int eogs_rast_forward_counts(int64_t* num_rendered);
/* Exact token (and error state, as forward_prepare reports it) of the forward that last ran over `geom`: copies its count
 * words back on `stream` and WAITS for the stream. For forwards queued with EOGS_FLAG_NO_READBACK — a replayed graph — whose
 * caller checks afterwards that the capacity it recorded was enough (eogs_rast_capacity_token's `fits`). */
int eogs_rast_read_counts(int P, int H, int W, const void* geom, size_t geom_bytes, int have_scratch, void* stream,
                          int64_t* num_rendered);
/* The same without waiting for the stream: the count words of a forward mirrored into host memory by a copy that is part
 * of the stream's work (and so of a captured graph). `host`: EOGS_MIRROR_BYTES of pinned (hipHostMalloc, coherent) memory,
 * 64-byte aligned, owned by the caller.
 *   eogs_rast_mirror_arm    marks `host` "not arrived" (call before the work — the replay — is queued);
 *   eogs_rast_mirror_counts queues the copy on `stream`; call it after forward_prepare(..., EOGS_FLAG_NO_READBACK) of the
 *                           forward over `geom` (any time before the next forward_prepare over the same `geom`);
 *   eogs_rast_mirror_token  *arrived = 0 while the copy has not landed (poll; the counts are final a few kernels into the
 *                           forward, long before the stream is idle), else 1 and the exact token / error as read_counts. */
#define EOGS_MIRROR_BYTES 64
int eogs_rast_mirror_arm(void* host);
int eogs_rast_mirror_counts(int P, const void* geom, size_t geom_bytes, void* host, void* stream);
int eogs_rast_mirror_token(int P, int H, int W, const void* host, int have_scratch, int64_t* num_rendered, int* arrived);
/* A token whose workspaces hold `slack` (e.g. 0.25) more record slots and list entries than `num_rendered` (an exact token
 * of an earlier forward with the same P, H, W) describes, for a deferred-count forward; *fits (optional; pass NULL when there
 * is no exact token yet) receives whether the exact token `exact` fits inside it — its counts do (0 = nothing listed always
 * does). The capacity token keeps the earlier forward's list granularity and kernel hints: speed only, every choice computes
 * the same values. have_scratch: the caller passes a scratch buffer to both calls. */
int eogs_rast_capacity_token(int P, int64_t num_rendered, double slack, int have_scratch, int64_t exact,
                             int64_t* capacity, int* fits);

/* Forward, phase 2: duplicate-with-keys, (tile,depth) sort, tile ranges, alpha blend.
 * Replaces DGR/cuda_rasterizer/rasterizer_impl.cu:290-340 (duplicateWithKeys,
 * cub::DeviceRadixSort::SortPairs, identifyTileRanges, FORWARD::render).
 * out_invdepth may be NULL: the inverse-depth image is then not blended at all (the reference's render() drops it,
 * gaussian_renderer/renderer.py:101,126; eogs2_amd.render.render passes NULL). Asynchronous on `stream`. */
int eogs_rast_forward_render(
    int P, int H, int W, int64_t num_rendered,
    const float* bg, unsigned flags,
    void* geom, size_t geom_bytes, void* binning, size_t binning_bytes,
    void* image, size_t image_bytes, void* scratch, size_t scratch_bytes,
    float* out_color, float* out_invdepth, void* stream);

/* Backward.
 * Replaces CudaRasterizer::Rasterizer::backward (DGR/cuda_rasterizer/rasterizer_impl.cu:345-452:
 * BACKWARD::render, computeCov2DCUDA, BACKWARD::preprocessCUDA) and the zero-filled
 * gradient allocation of DGR/rasterize_points.cu:163-174 — every gradient output is
 * fully (over)written here, the caller may pass uninitialised memory.
 *   bg             f32[5], required (the background term of dL/dalpha, DGR/cuda_rasterizer/backward.cu:527-529,617-620)
 *   out_color      forward's rendered image f32[5,H,W]: ignored since ABI 8 (as by the oracle), may be NULL — the walk starts
 *                  from the final transmittance the forward left in the image workspace, as the reference's does
 *   out_invdepth   forward's inverse-depth image f32[H,W]: ignored since ABI 8, may be NULL
 *   dL_dout_color  f32[5,H,W];  dL_dout_invdepth f32[H,W] or NULL
 *   dL_dmeans2D f32[P,3] (NDC units, z = 0)  dL_dcolors f32[P,5]  dL_dopacity f32[P]
 *   dL_dmeans3D f32[P,3]  dL_dcov3D f32[P,6]  dL_dscales f32[P,3]  dL_drotations f32[P,4]
 *     (dL_dscales/dL_drotations may be NULL when cov3D_precomp is used; dL_dcov3D may be NULL when scales/rotations are
 *     used: the reference always materialises it, DGR/rasterize_points.cu:167, and its wrapper then discards it)
 *   dL_dT_sum  f32[6] or NULL: sum over Gaussians of dL/dT (2x3, row-major), T = diag(W/2,H/2)·A[0:2,:],
 *     i.e. the reduction the reference wrapper performs on its [P,6] dL_dT tensor
 *     (DGR/diff_gaussian_rasterization/__init__.py:179-190) under the intended 6*idx layout
 *     (the reference kernel writes dL_dT[idx+k], DGR/cuda_rasterizer/backward.cu:320-325).
 *   dL_dvm_mean f32[12] or NULL: [0:9] = means3D^T @ dL_dmeans2D (3x3 row-major), [9:12] = sum_P dL_dmeans2D
 *     (DGR/diff_gaussian_rasterization/__init__.py:193-201).
 *   dL_dcolors_lead f32[P, lead_cols] or NULL: a second destination for the first lead_cols columns of every dL_dcolors
 *     row, contiguous. A data-parallel caller points it at the f_dc block of its gradient exchange buffer (the three
 *     colour columns of colors_precomp are model parameters, the altitude / constant columns are not), so that no
 *     copy kernel runs between the backward and the collective. dL_dcolors is written in full either way.
 */
int eogs_rast_backward(
    int P, int H, int W, int64_t num_rendered,
    const float* bg, const float* means3D, const int* radii, const float* colors,
    const float* opacities, const float* scales, const float* rotations,
    float scale_modifier, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* alt_affine, unsigned flags,
    const float* out_color, const float* out_invdepth,
    const float* dL_dout_color, const float* dL_dout_invdepth,
    const void* geom, size_t geom_bytes, const void* binning, size_t binning_bytes,
    const void* image, size_t image_bytes,
    float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity,
    float* dL_dmeans3D, float* dL_dcov3D, float* dL_dscales, float* dL_drotations,
    float* dL_dT_sum, float* dL_dvm_mean, float* dL_dcolors_lead, int lead_cols, void* stream);

/* Backward over a range of Gaussians: the same computation as eogs_rast_backward, split so that a data-parallel
 * caller can hand finished gradient rows to the collective while later rows are still being computed
 * (the exchange step of view-sharded training; the reference is single-GPU,
 * src/gaussiansplatting/train_pan.py:252-257). Arguments as eogs_rast_backward; the gradient pointers are the
 * FULL arrays (row 0). A call with p_begin == 0 first runs the per-pixel pass (BACKWARD::render) for the whole
 * image; every call then runs the per-Gaussian pass (computeCov2DCUDA + BACKWARD::preprocessCUDA) for rows
 * [p_begin, p_end) only. The caller covers [0, P) with ascending, adjacent ranges on ONE stream; p_begin must be a
 * multiple of 256. dL_dT_sum / dL_dvm_mean are valid after the call whose p_end == P.
 * eogs_rast_backward(...) == eogs_rast_backward_range(..., 0, P), bit for bit. */
int eogs_rast_backward_range(
    int P, int H, int W, int64_t num_rendered,
    const float* bg, const float* means3D, const int* radii, const float* colors,
    const float* opacities, const float* scales, const float* rotations,
    float scale_modifier, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* alt_affine, unsigned flags,
    const float* out_color, const float* out_invdepth,
    const float* dL_dout_color, const float* dL_dout_invdepth,
    const void* geom, size_t geom_bytes, const void* binning, size_t binning_bytes,
    const void* image, size_t image_bytes,
    float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity,
    float* dL_dmeans3D, float* dL_dcov3D, float* dL_dscales, float* dL_drotations,
    float* dL_dT_sum, float* dL_dvm_mean, float* dL_dcolors_lead, int lead_cols, int p_begin, int p_end, void* stream);

/* Replaces CudaRasterizer::Rasterizer::markVisible (DGR/cuda_rasterizer/rasterizer_impl.cu:141-153).
 * The reference predicate has its culling commented out (DGR/cuda_rasterizer/auxiliary.h:151-176),
 * so every Gaussian is reported visible: present[i] = 1. */
int eogs_rast_mark_visible(int P, const float* means3D, const float* viewmatrix,
                           const float* projmatrix, uint8_t* present, void* stream);

/* ---- Diagnostics (no reference counterpart; the reference has no profiler) ----
 * When enabled, every kernel group launched by the calls above is bracketed by a pair of hipEvents recorded
 * on the caller's stream. eogs_rast_profile_get() waits for the recorded events and returns, for slot
 * 0..eogs_rast_profile_slots()-1, the accumulated device time, the number of bracketed launches and the
 * group's name ("preprocess_fwd", "depth_sort", "binning", "render_fwd", "render_bwd", "gaussian_bwd").
 * The oracle implements them as no-ops (0 slots). Process-wide (backward may run on another thread). */
int eogs_rast_profile_enable(int on);
/* Restricts the bracketing to the slots whose bit is set (default: all). Each bracket costs two event records
 * (about 3.4 us of queue time each on the MI355X), so a timed region brackets only the kernel it needs. */
int eogs_rast_profile_select(unsigned slot_mask);
int eogs_rast_profile_reset(void);
int eogs_rast_profile_slots(void);
int eogs_rast_profile_get(int slot, double* total_ms, int64_t* launches, const char** name);
/* Which kernels a (P, num_rendered) pair selects — the library picks list granularity and render kernel per forward
 * (DESIGN.md §2.3, §2.5); tests use this to record that every kernel was compared with the oracle.
 *   *list_block_px  8 (per-tile lists) or 32 (block lists)
 *   *fwd_kernel / *bwd_kernel  0 = one list per tile (render_*_kernel<1>), 1 = block lists (render_*_kernel<4>),
 *                              2 = quad sub-lists (render_*_quad_kernel); backward only: 6 = the quad kernel's one-channel
 *                              variant of an altitude-only render. (3, 4, 5 — two matrix-pipe experiments and a separate
 *                              back-to-front kernel — existed up to ABI 7; every backward kernel now walks back to front,
 *                              DGR/cuda_rasterizer/backward.cu:536-643.)
 * The oracle reports 16 / -1 / -1 (the reference's 16-px tiles, no kernel variants). */
int eogs_rast_path_info(int P, int64_t num_rendered, int* list_block_px, int* fwd_kernel, int* bwd_kernel);
/* Which build of the per-Gaussian backward kernel eogs_rast_backward would launch for this token right now: 0 = four records
 * in flight per lane (seven waves per SIMD), 1 / 2 = eight (four waves; 2: up to eight listed tiles in one trip). It depends on
 * the token's counts and its hint bit (the forward's list depth x mean pair opacity, see forward_prepare): deep in
 * opacity -> 0. All three compute the same bits; tests use this to see the hint travel. The oracle reports -1. */
int eogs_rast_backward_info(int P, int64_t num_rendered, int* gaussian_bwd_wide);
/* Runs the library's wave64 primitive self-test (DPP reduction, readlane broadcast) on `stream` and returns,
 * after synchronising, a bit mask of failing primitives in *failed (0 = all good). scratch: >= 4 device bytes. */
int eogs_rast_selftest(void* scratch, unsigned* failed, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EOGS_RAST_H_INCLUDED */
