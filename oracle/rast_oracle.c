/*
 * rast_oracle.c — CPU restatement of the reference rasterizer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (eogs2_amd/,
 * diff_gaussian_rasterization/) may import, link or call this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * only as the checker.
 *
 * PARITY UNPINNED (against reference outputs): the reference's arithmetic is CUDA-only (nvcc + CUB + un-vendored
 * glm) and its tests hold no golden vectors for this path,
 * so this oracle cannot be checked against reference outputs.  It is pinned
 * instead (tests/test_oracle_pins.py) against (1) an independent dense
 * PyTorch/autograd renderer (oracle/torch_dense.py) on forward and on every
 * gradient, and (2) the reference's own Python wrapper
 * (DGR/diff_gaussian_rasterization/__init__.py) imported with `_C` stubbed by
 * this library, which pins argument/tuple order and the grad_viewmatrix
 * assembly (fixtures under tests/golden/).
 *
 * It exports the same C-ABI as include/eogs_rast.h, over HOST pointers.
 * Each function cites the reference lines it follows.  DGR/ =
 * src/gaussiansplatting/submodules/diff-gaussian-rasterization/.
 *
 * Arithmetic: fp32 exactly where the reference is fp32, double temporaries
 * exactly where the reference promotes (ndc2Pix, depth).  The only deliberate
 * deviation: per-Gaussian gradient sums over pixels, which the reference
 * accumulates with fp32 atomicAdd in non-deterministic order
 * (DGR/cuda_rasterizer/backward.cu:598-640), are accumulated here in double
 * and rounded once, i.e. the order-independent value every fp32 ordering
 * approximates.
 *
 * Three builds of this one file (oracle/Makefile): librast_oracle.so, the restatement proper (fp32, one rounded operation per
 * operation of the reference); librast_oracle_fma.so, the same with fused multiply-adds allowed (a second valid rounding); and
 * librast_oracle_f64.so (-DORACLE_F64), the ARBITER: the fp32 forward and its decisions, the backward's differentiable quantities
 * recomputed and chained in double — what tests/parity_cases.py measures every fp32 evaluation's distance from when two of them
 * disagree on an ill-conditioned gradient (see "the arbiter build" below). Switches for the tests' attribution
 * (eogs_oracle_threshold_nudge, _accum_float, _pair_noise, _suffix_by_subtraction) are documented where they are defined; none of
 * them is part of include/eogs_rast.h.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * Threads (round 4): the two per-pixel loops run on the host's cores — the forward's pixels are independent; the backward
 * takes its per-Gaussian double sums per band of image rows (ORACLE_BANDS, a constant) and adds the bands in ascending
 * order — so every result is the same number whatever the thread count (eogs_oracle_set_threads; 1 = the single-core run
 * bench.py reports as `scalar_c`). The committed fixtures regenerate bit for bit with it (tests/golden/make_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/eogs_rast.h"

#define C_ EOGS_RAST_CHANNELS
#define TILE EOGS_RAST_TILE
#define BATCH (TILE * TILE)

static __thread char g_err[256];

/* The per-Gaussian sums of BACKWARD::renderCUDA are fp32 atomicAdds in the reference (backward.cu:598-640), in an order that
 * changes from run to run; the restatement accumulates them in double (the order-independent value the fp32 sums scatter
 * around). For the tests' sensitivity attribution (tests/parity_cases.py) the accumulation can be switched to fp32 in
 * pixel order — one of the orders the reference itself can produce — so that the size of the summation-order effect is
 * measured rather than assumed. Test infrastructure only; not part of include/eogs_rast.h. */
static int g_acc_float = 0;
int eogs_oracle_accum_float(int on) { const int old = g_acc_float; g_acc_float = on != 0; return old; }
/* Diagnostic (tools/suffix_probe.py): evaluate dL/dalpha the way the HIP path does — front to back, the sum
 * behind a Gaussian obtained by subtracting the running prefix from the rendered total,
 *   dL/dalpha_j = T_j (g.c_j) - (D_final - D_j) / (1 - alpha_j),   D_final = sum_ch g_ch out_ch,
 * in fp32 — instead of the reference's back-to-front recursion. Algebraically identical; used to measure how much of a
 * HIP-vs-oracle difference is this formulation. Needs forward's out_color (out_invdepth with an invdepth gradient). */
static int g_suffix_by_subtraction = 0;
/* 0: the reference's back-to-front recursion. 1: front to back, sum behind a Gaussian = rendered total (from out_color) minus running
 * prefix — the HIP path's formulation. 2: the same with the total taken from the running sum's own end value (a first walk over the list)
 * instead of the rendered image: tells how much of (1)'s deviation is the mismatch between the two differently associated totals.
 * 3: back to front like the reference, but with the colour behind a Gaussian carried as ONE number per pixel — the reference's accum_rec
 * projected on the pixel's upstream gradient, a_j = g . accum_rec_j, with a_j = a_{j+1} + alpha_{j+1} (g.c_{j+1} - a_{j+1}) — and T
 * recovered with a reciprocal: the HIP path's formulation since round 6 (csrc/render.hip). Same recursion, another association. */
int eogs_oracle_suffix_by_subtraction(int on) { const int old = g_suffix_by_subtraction; g_suffix_by_subtraction = on; return old; }
/* Diagnostic (tests/parity_cases.py arbiter()): one more VALID fp32 evaluation of the backward — every pair's exponent
 * `power = -1/2 (a dx^2 + c dy^2) - b dx dy` and its G = exp(power) carry the rounding of ANOTHER association: power a relative
 * error of up to 2 `ulps` (five rounded operations whose order the reference's compiler, the restatement and the HIP path's
 * log2-domain form (A dx - B dy) dx + C dy^2 each choose differently — G inherits |power| times that), exp() one of up to `ulps`
 * (libm's expf is correctly rounded to well under an ulp; CUDA's expf, which the reference runs, is documented at 2 ulp;
 * v_exp_f32 at 1 ulp), deterministic in (pixel, Gaussian, seed). What the other variants (fp32 sums, FMA contraction, perturbed
 * inputs) do not model is this per-pair noise, which the ill-conditioned per-Gaussian chain amplifies like any other error of
 * the conic-gradient sums. Decisions stay those of the un-noised fp32 forward; the pixel's final transmittance is recomputed
 * from the noised alphas (as the arbiter build recomputes it in double), so the recursion stays self-consistent. 0 = off. */
static float g_pair_noise_ulps = 0.f;
static uint32_t g_pair_noise_seed = 0;
int eogs_oracle_pair_noise(float ulps, uint32_t seed) { g_pair_noise_ulps = ulps < 0.f ? 0.f : ulps; g_pair_noise_seed = seed; return 0; }
static inline uint32_t pair_hash(size_t pix_id, uint32_t id, uint32_t salt) {
  uint32_t h = (uint32_t)pix_id * 0x9E3779B1u ^ (id + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (g_pair_noise_seed + salt) * 0xC2B2AE35u;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}
/* the pair's G under the noise model: exp(power (1 + 2 ulps e r1)) (1 + ulps e r2), r in [-1, 1), e = 2^-23 */
static inline float noised_G(size_t pix_id, uint32_t id, float power) {
  const float r1 = (float)(int32_t)pair_hash(pix_id, id, 0u) * (1.0f / 2147483648.0f);
  const float r2 = (float)(int32_t)pair_hash(pix_id, id, 1u) * (1.0f / 2147483648.0f);
  return expf(power * (1.0f + 2.0f * g_pair_noise_ulps * 1.1920929e-07f * r1)) * (1.0f + g_pair_noise_ulps * 1.1920929e-07f * r2);
}
/* Diagnostic (tests/parity_cases.py, causal attribution of threshold flips): the reference's two data-dependent blend
 * decisions (forward.cu:374-382) evaluated with their thresholds moved by a stated number of ulp, per pixel:
 *   skip    if  alpha < (1/255) (1 + s (ka0 + ka1 M) ulp)     (M = the magnitude of the exponent's terms, alpha_min below)
 *   stop    if  T'    < 1e-4   (1 + s (kT0 + kT1 n) ulp),  n = Gaussians blended so far at the pixel
 * s in {-1, 0, +1}: g_nudge_map[pixel] when a map is installed, else g_nudge_uniform. s = 0 is the reference's arithmetic,
 * bit for bit. Two fp32 implementations that round the exponent differently can land on different sides of a threshold
 * for a pair within these margins; a test that sees a difference re-runs the oracle with the decision moved and must
 * then find agreement. */
#define ORACLE_ULP 1.1920928955078125e-7f
static int g_nudge_uniform = 0;
static signed char* g_nudge_map = NULL;
static size_t g_nudge_n = 0;
static float g_nk[4] = {16.f, 8.f, 16.f, 4.f};
int eogs_oracle_threshold_nudge(int uniform, const signed char* map, size_t n, const float* k4) {
  free(g_nudge_map);
  g_nudge_map = NULL;
  g_nudge_n = 0;
  g_nudge_uniform = uniform < 0 ? -1 : (uniform > 0 ? 1 : 0);
  if (k4) for (int i = 0; i < 4; i++) g_nk[i] = k4[i];
  if (map && n) {
    if (!(g_nudge_map = (signed char*)malloc(n))) return -1;
    memcpy(g_nudge_map, map, n);
    g_nudge_n = n;
  }
  return 0;
}
static inline int nudge_sign(size_t pix_id) { return g_nudge_map ? (pix_id < g_nudge_n ? g_nudge_map[pix_id] : 0) : g_nudge_uniform; }
/* `co`, dx, dy: the pair's conic and offset. The margin grows with the MAGNITUDE of the exponent's terms, M = |a| dx^2 / 2 +
 * |c| dy^2 / 2 + |b dx dy| — what the rounding error of its five operations scales with (about M ulp absolute, i.e. M ulp of alpha
 * relative). M equals |power| unless the cross term cancels the squares: a thin rotated Gaussian evaluated far out along its long
 * axis. Rounds 3-5 used |power|; round 6's fresh sweep range 9000-9399 held two such pairs — seed 9241: terms -1720.0, -1766.7,
 * +3485.8 summing to a power of -0.94, alpha 1563 ulp below 1/255 where the margin was 24 ulp; seed 9376: -5004.2, -4899.6, +9899.0
 * -> -4.75, 3510 ulp above — one pixel each on which the HIP path, which evaluates (A dx - B dy) dx + C dy^2 with a pre-scaled conic,
 * decides the other way; the oracle run with the threshold moved by THIS margin reproduces those pixels channel for channel. */
static inline float alpha_min(int s, const float* co, float dx, float dy) {
  if (!s) return 1.0f / 255.0f;
  const float mag = 0.5f * (fabsf(co[0]) * dx * dx + fabsf(co[2]) * dy * dy) + fabsf(co[1] * dx * dy);
  return (1.0f / 255.0f) * (1.0f + (float)s * (g_nk[0] + g_nk[1] * mag) * ORACLE_ULP);
}
static inline float T_min(int s, uint32_t nblended) {
  return s ? 0.0001f * (1.0f + (float)s * (g_nk[2] + g_nk[3] * (float)nblended) * ORACLE_ULP) : 0.0001f;
}
#define ACC(a, term) do { if (g_acc_float) (a) = (double)((float)(a) + (float)(term)); else (a) += (double)(term); } while (0)

/* Threads of the per-pixel loops (test infrastructure: speeds up the checker, never changes its results — forward pixels are
 * independent, the backward's sums are taken per band of rows in a fixed order whatever the thread count). 0 = as many as the
 * machine offers, at most 16; bench.py's single-core `scalar_c` baseline sets 1. */
#ifdef _OPENMP
#include <omp.h>
#endif
#define ORACLE_BANDS 8 /* bands of image rows the backward's per-Gaussian sums are taken over (backward_activated) */
static int g_threads = 0;
int eogs_oracle_set_threads(int n) { const int old = g_threads; g_threads = n < 0 ? 0 : n; return old; }
static int oracle_threads(void) {
#ifdef _OPENMP
  if (g_threads > 0) return g_threads;
  const int m = omp_get_num_procs();
  return m > 16 ? 16 : (m < 1 ? 1 : m);
#else
  return 1;
#endif
}

static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}

const char* eogs_rast_last_error(void) { return g_err; }
int eogs_rast_abi_version(void) { return EOGS_RAST_ABI_VERSION; }
const char* eogs_rast_backend(void) { return "cpu-oracle"; }

/* ---- workspace carving (same idea as DGR/cuda_rasterizer/rasterizer_impl.h:22-28 `obtain`) ---- */
static size_t align_up(size_t x) { return (x + 127u) & ~(size_t)127u; }

typedef struct {
  float* means2D;       /* [P,2] pixel centre  */
  float* depths;        /* [P]   200 - altitude */
  float* cov3D;         /* [P,6] */
  float* conic_opacity; /* [P,4] */
  uint32_t* tiles_touched; /* [P] */
  uint32_t* point_offsets; /* [P] inclusive scan */
  int* radii;              /* [P] internal copy (GeometryState::internal_radii, rasterizer_impl.h:38) */
  float* colors;           /* [P,5] copy of colors_precomp handed to forward_prepare */
} Geom;

typedef struct {
  uint64_t* keys;     /* [R] sorted */
  uint32_t* values;   /* [R] sorted = point_list */
  uint64_t* keys_tmp;
  uint32_t* values_tmp;
} Binning;

typedef struct {
  uint32_t* ranges;   /* [T,2] */
  float* final_T;     /* [H*W] */
  uint32_t* n_contrib; /* [H*W] */
} Image;

static size_t carve(char* base, size_t off, void** p, size_t bytes) {
  off = align_up(off);
  if (base) *p = base + off;
  return off + bytes;
}

static size_t geom_layout(char* base, int P, Geom* g) {
  size_t o = 0, n = (size_t)P;
  Geom d;
  o = carve(base, o, (void**)&d.means2D, n * 2 * 4);
  o = carve(base, o, (void**)&d.depths, n * 4);
  o = carve(base, o, (void**)&d.cov3D, n * 6 * 4);
  o = carve(base, o, (void**)&d.conic_opacity, n * 4 * 4);
  o = carve(base, o, (void**)&d.tiles_touched, n * 4);
  o = carve(base, o, (void**)&d.point_offsets, n * 4);
  o = carve(base, o, (void**)&d.radii, n * 4);
  o = carve(base, o, (void**)&d.colors, n * C_ * 4);
  if (g) *g = d;
  return align_up(o) + 128;
}

static size_t binning_layout(char* base, int64_t R, Binning* b) {
  size_t o = 0, n = (size_t)R;
  Binning d;
  o = carve(base, o, (void**)&d.keys, n * 8);
  o = carve(base, o, (void**)&d.values, n * 4);
  o = carve(base, o, (void**)&d.keys_tmp, n * 8);
  o = carve(base, o, (void**)&d.values_tmp, n * 4);
  if (b) *b = d;
  return align_up(o) + 128;
}

static size_t image_layout(char* base, int H, int W, Image* im) {
  size_t o = 0, n = (size_t)H * W;
  size_t T = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
  Image d;
  o = carve(base, o, (void**)&d.ranges, T * 2 * 4);
  o = carve(base, o, (void**)&d.final_T, n * 4);
  o = carve(base, o, (void**)&d.n_contrib, n * 4);
  if (im) *im = d;
  return align_up(o) + 128;
}

int eogs_rast_geom_bytes(int P, size_t* bytes) {
  if (P < 0 || !bytes) return fail(EOGS_ERR_INVALID_ARG, "geom_bytes: bad argument");
  *bytes = geom_layout(NULL, P, NULL);
  return EOGS_OK;
}
int eogs_rast_image_bytes(int H, int W, size_t* bytes) {
  if (H < 0 || W < 0 || !bytes) return fail(EOGS_ERR_INVALID_ARG, "image_bytes: bad argument");
  *bytes = image_layout(NULL, H, W, NULL);
  return EOGS_OK;
}
int eogs_rast_binning_bytes(int P, int H, int W, int64_t R, size_t* bytes) {
  (void)P; (void)H; (void)W;
  if (R < 0 || !bytes) return fail(EOGS_ERR_INVALID_ARG, "binning_bytes: bad argument");
  *bytes = binning_layout(NULL, R, NULL);
  return EOGS_OK;
}
/* the oracle sorts inside its binning workspace like the reference: the transient scratch of the ABI is unused */
int eogs_rast_scratch_bytes(int P, int H, int W, size_t* bytes) {
  (void)H; (void)W;
  if (P < 0 || !bytes) return fail(EOGS_ERR_INVALID_ARG, "scratch_bytes: bad argument");
  *bytes = 0;
  return EOGS_OK;
}

/* ---- small helpers ---- */

/* DGR/cuda_rasterizer/auxiliary.h:40-43 — evaluated in double (the literals are double), then narrowed. */
static float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* DGR/cuda_rasterizer/auxiliary.h:45-55 — truncating float->int division by the tile size, clamped to the grid. */
static void get_rect(float px, float py, int max_radius, int gx, int gy,
                     int* x0, int* y0, int* x1, int* y1) {
  *x0 = imin(gx, imax(0, (int)((px - max_radius) / TILE)));
  *y0 = imin(gy, imax(0, (int)((py - max_radius) / TILE)));
  *x1 = imin(gx, imax(0, (int)((px + max_radius + TILE - 1) / TILE)));
  *y1 = imin(gy, imax(0, (int)((py + max_radius + TILE - 1) / TILE)));
}

/* Rotation as the reference builds it (DGR/cuda_rasterizer/forward.cu:126-137): glm::mat3 is filled
 * column-major, so the glm matrix R satisfies R[c][r]; we keep "math" row-major Rm[r][c] = R_glm[c][r]. */
static void quat_to_Rm(const float q[4], float Rm[3][3]) {
  float r = q[0], x = q[1], y = q[2], z = q[3];
  /* glm column 0 = (1-2(yy+zz), 2(xy-rz), 2(xz+ry)) -> math column 0 */
  Rm[0][0] = 1.f - 2.f * (y * y + z * z); Rm[1][0] = 2.f * (x * y - r * z); Rm[2][0] = 2.f * (x * z + r * y);
  Rm[0][1] = 2.f * (x * y + r * z); Rm[1][1] = 1.f - 2.f * (x * x + z * z); Rm[2][1] = 2.f * (y * z - r * x);
  Rm[0][2] = 2.f * (x * z - r * y); Rm[1][2] = 2.f * (y * z + r * x); Rm[2][2] = 1.f - 2.f * (x * x + y * y);
}

/* DGR/cuda_rasterizer/forward.cu:117-151 computeCov3D: M = S*R, Sigma = M^T M, upper triangle.
 * The quaternion is NOT renormalised (:126). */
static void cov3d_from_scale_rot(const float s[3], float mod, const float q[4], float out[6]) {
  float Rm[3][3], M[3][3], Sg[3][3];
  quat_to_Rm(q, Rm);
  float sc[3] = {mod * s[0], mod * s[1], mod * s[2]};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M[i][j] = sc[i] * Rm[i][j];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      Sg[i][j] = M[0][i] * M[0][j] + M[1][i] * M[1][j] + M[2][i] * M[2][j];
  out[0] = Sg[0][0]; out[1] = Sg[0][1]; out[2] = Sg[0][2];
  out[3] = Sg[1][1]; out[4] = Sg[1][2]; out[5] = Sg[2][2];
}

/* T rows used by both cov2D forward and backward (DGR/cuda_rasterizer/forward.cu:93-102,
 * backward.cu:175-190): T = W * NDC2Screen with W = viewmatrix^T-upper-3x3, so
 * Trow[i][k] = vm[4k+i] * s_i, s = (W/2, H/2, 1). In glm notation Trow[i][k] == T[i][k]. */
static void build_T(const float* vm, int W, int H, float T[3][3]) {
  float s[3] = {(float)(W / 2.0), (float)(H / 2.0), 1.0f};
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 3; k++) T[i][k] = vm[4 * k + i] * s[i];
}

static void sym_from6(const float c[6], float V[3][3]) {
  V[0][0] = c[0]; V[0][1] = c[1]; V[0][2] = c[2];
  V[1][0] = c[1]; V[1][1] = c[3]; V[1][2] = c[4];
  V[2][0] = c[2]; V[2][1] = c[4]; V[2][2] = c[5];
}

/* DGR/cuda_rasterizer/forward.cu:74-112 computeCov2D: cov = T^T Vrk^T T (glm), upper-left 2x2.
 * cov(i,j) = T[i] . Vrk . T[j]; left-associated as glm evaluates (T^T Vrk^T) first. */
static void cov2d(const float T[3][3], const float cov3D[6], float* cxx, float* cxy, float* cyy) {
  float V[3][3], A[2][3];
  sym_from6(cov3D, V);
  for (int i = 0; i < 2; i++)
    for (int l = 0; l < 3; l++)
      A[i][l] = T[i][0] * V[0][l] + T[i][1] * V[1][l] + T[i][2] * V[2][l];
  *cxx = A[0][0] * T[0][0] + A[0][1] * T[0][1] + A[0][2] * T[0][2];
  *cxy = A[1][0] * T[0][0] + A[1][1] * T[0][1] + A[1][2] * T[0][2]; /* glm cov[0][1] */
  *cyy = A[1][0] * T[1][0] + A[1][1] * T[1][1] + A[1][2] * T[1][2];
}

/* ---- the arbiter build (oracle/Makefile: librast_oracle_f64.so, -DORACLE_F64) ----
 * `real` is the type the BACKWARD computes in: float in the oracle proper (the reference's fp32, bit for bit as restated), double in
 * the arbiter build, which tests/parity_cases.py uses to decide who is right where the fp32 oracle and the HIP path disagree on an
 * ill-conditioned gradient (covariance backward of strongly anisotropic Gaussians, backward.cu:239-394): an element is accepted iff the
 * HIP value is no further from the double evaluation than twice what valid fp32 evaluations of the reference's algorithm are.
 * The arbiter evaluates the SAME function — the same blend / skip / stop decisions per (pixel, Gaussian) pair, taken from the
 * fp32 forward exactly as the oracle proper takes them — with every differentiable quantity recomputed in double from the inputs:
 * the per-Gaussian forward chain (projection, cov3D, cov2D, conic, antialiasing scale, depth), the final transmittance of each
 * pixel, the per-pixel recursion and the per-Gaussian backward chain. Its forward entry points are the fp32 ones, unchanged. */
#ifdef ORACLE_F64
typedef double real;
#define R_EXP exp
#define R_SQRT sqrt
#define R_FMAX fmax
#define R_FMIN fmin
int eogs_oracle_is_f64(void) { return 1; }
#else
typedef float real;
#define R_EXP expf
#define R_SQRT sqrtf
#define R_FMAX fmaxf
#define R_FMIN fminf
int eogs_oracle_is_f64(void) { return 0; }
#endif
#define RC(x) ((real)(x))

/* (the helpers above once more in `real`: the same operations in the same order — in the fp32 build the same bits) */
static void quat_to_Rm_r(const real q[4], real Rm[3][3]) {
  real r = q[0], x = q[1], y = q[2], z = q[3];
  Rm[0][0] = RC(1) - RC(2) * (y * y + z * z); Rm[1][0] = RC(2) * (x * y - r * z); Rm[2][0] = RC(2) * (x * z + r * y);
  Rm[0][1] = RC(2) * (x * y + r * z); Rm[1][1] = RC(1) - RC(2) * (x * x + z * z); Rm[2][1] = RC(2) * (y * z - r * x);
  Rm[0][2] = RC(2) * (x * z - r * y); Rm[1][2] = RC(2) * (y * z + r * x); Rm[2][2] = RC(1) - RC(2) * (x * x + y * y);
}
static void build_T_r(const float* vm, int W, int H, real T[3][3]) {
  real s[3] = {(real)(W / 2.0), (real)(H / 2.0), RC(1)};
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 3; k++) T[i][k] = (real)vm[4 * k + i] * s[i];
}
static void sym_from6_r(const real c[6], real V[3][3]) {
  V[0][0] = c[0]; V[0][1] = c[1]; V[0][2] = c[2];
  V[1][0] = c[1]; V[1][1] = c[3]; V[1][2] = c[4];
  V[2][0] = c[2]; V[2][1] = c[4]; V[2][2] = c[5];
}
static void cov2d_r(const real T[3][3], const real cov3D[6], real* cxx, real* cxy, real* cyy) {
  real V[3][3], A[2][3];
  sym_from6_r(cov3D, V);
  for (int i = 0; i < 2; i++)
    for (int l = 0; l < 3; l++)
      A[i][l] = T[i][0] * V[0][l] + T[i][1] * V[1][l] + T[i][2] * V[2][l];
  *cxx = A[0][0] * T[0][0] + A[0][1] * T[0][1] + A[0][2] * T[0][2];
  *cxy = A[1][0] * T[0][0] + A[1][1] * T[0][1] + A[1][2] * T[0][2];
  *cyy = A[1][0] * T[1][0] + A[1][1] * T[1][1] + A[1][2] * T[1][2];
}
#ifdef ORACLE_F64
static void cov3d_from_scale_rot_r(const float s[3], float mod, const float q[4], real out[6]) {
  real Rm[3][3], M[3][3], Sg[3][3];
  const real qr[4] = {q[0], q[1], q[2], q[3]};
  quat_to_Rm_r(qr, Rm);
  real sc[3] = {(real)mod * s[0], (real)mod * s[1], (real)mod * s[2]};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M[i][j] = sc[i] * Rm[i][j];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      Sg[i][j] = M[0][i] * M[0][j] + M[1][i] * M[1][j] + M[2][i] * M[2][j];
  out[0] = Sg[0][0]; out[1] = Sg[0][1]; out[2] = Sg[0][2];
  out[3] = Sg[1][1]; out[4] = Sg[1][2]; out[5] = Sg[2][2];
}
#endif

/* ------------------------------------------------------------------------------------------- */
/* Forward phase 1: FORWARD::preprocessCUDA + InclusiveSum                                     */
/* DGR/cuda_rasterizer/forward.cu:154-283, rasterizer_impl.cu:250-284                          */
/* ------------------------------------------------------------------------------------------- */
static int forward_prepare_activated(
    int P, int H, int W,
    const float* means3D, const float* scales, const float* rotations,
    const float* cov3D_precomp, const float* opacities, const float* colors, float scale_modifier,
    const float* viewmatrix, const float* projmatrix, unsigned flags,
    int* radii, void* geom, size_t geom_bytes,
    int64_t* num_rendered, void* stream) {
  (void)projmatrix; (void)stream;
  g_err[0] = 0;
  if (P < 0 || H <= 0 || W <= 0 || !num_rendered) return fail(EOGS_ERR_INVALID_ARG, "forward_prepare: bad sizes");
  *num_rendered = 0;
  if (P == 0) return EOGS_OK;
  if (!colors) return fail(EOGS_ERR_NO_COLORS, "For non-RGB, provide precomputed Gaussian colors!");
  if (!means3D || !opacities || !viewmatrix || !radii || !geom)
    return fail(EOGS_ERR_INVALID_ARG, "forward_prepare: NULL input");
  int have_sr = scales && rotations, have_cov = cov3D_precomp != NULL;
  if (have_sr == have_cov || (!!scales != !!rotations))
    return fail(EOGS_ERR_INVALID_ARG, "forward_prepare: provide exactly one of scale/rotation pair or precomputed 3D covariance");
  if (geom_bytes < geom_layout(NULL, P, NULL)) return fail(EOGS_ERR_WORKSPACE, "forward_prepare: geom workspace too small");
  Geom g;
  geom_layout((char*)geom, P, &g);
  memcpy(g.colors, colors, (size_t)P * C_ * 4);
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  const int aa = (flags & EOGS_FLAG_ANTIALIASING) != 0;
  float T[3][3];
  build_T(viewmatrix, W, H, T);
  int too_high = 0;

  for (int idx = 0; idx < P; idx++) {
    radii[idx] = 0;
    g.radii[idx] = 0;
    g.tiles_touched[idx] = 0;
    /* in_frustum is a no-op that always lets the point through (auxiliary.h:151-176). */
    const float* p = means3D + 3 * (size_t)idx;
    /* transformPoint4x3 (auxiliary.h:70-78) */
    float pv[3];
    for (int i = 0; i < 3; i++)
      pv[i] = viewmatrix[i] * p[0] + viewmatrix[4 + i] * p[1] + viewmatrix[8 + i] * p[2] + viewmatrix[12 + i];

    const float* c3;
    if (have_cov) {
      c3 = cov3D_precomp + 6 * (size_t)idx;
      memcpy(g.cov3D + 6 * (size_t)idx, c3, 6 * 4); /* backward reads one array either way */
    } else {
      cov3d_from_scale_rot(scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, g.cov3D + 6 * (size_t)idx);
      c3 = g.cov3D + 6 * (size_t)idx;
    }
    float cx, cy, cz; /* cov.x, cov.y, cov.z */
    cov2d(T, c3, &cx, &cy, &cz);

    const float h_var = 0.3f;
    const float det_cov = cx * cz - cy * cy;
    cx += h_var;
    cz += h_var;
    const float det_cov_plus_h_cov = cx * cz - cy * cy;
    float h_convolution_scaling = 1.0f;
    if (aa) h_convolution_scaling = sqrtf(fmaxf(0.000025f, det_cov / det_cov_plus_h_cov));
    const float det = det_cov_plus_h_cov;
    if (det == 0.0f) continue;
    float det_inv = 1.f / det;
    float conic[3] = {cz * det_inv, -cy * det_inv, cx * det_inv};

    float mid = 0.5f * (cx + cz);
    float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
    float px = ndc2pix(pv[0], W), py = ndc2pix(pv[1], H);
    int x0, y0, x1, y1;
    get_rect(px, py, (int)my_radius, gx, gy, &x0, &y0, &x1, &y1);
    if ((x1 - x0) * (y1 - y0) == 0) continue;

    /* depth = 200 - altitude, double then narrowed (forward.cu:267); the reference traps if < 0. */
    float depth = (float)(200.0 - pv[2]);
    g.depths[idx] = depth;
    if (depth < 0) too_high = 1;
    radii[idx] = (int)my_radius;
    g.radii[idx] = (int)my_radius;
    g.means2D[2 * (size_t)idx] = px;
    g.means2D[2 * (size_t)idx + 1] = py;
    float* co = g.conic_opacity + 4 * (size_t)idx;
    co[0] = conic[0]; co[1] = conic[1]; co[2] = conic[2];
    co[3] = opacities[idx] * h_convolution_scaling;
    g.tiles_touched[idx] = (uint32_t)((y1 - y0) * (x1 - x0));
  }
  if (too_high) return fail(EOGS_ERR_ALTITUDE, "Point is too high: altitude > 200");

  /* InclusiveSum (rasterizer_impl.cu:280) */
  uint64_t run = 0;
  for (int i = 0; i < P; i++) {
    run += g.tiles_touched[i];
    g.point_offsets[i] = (uint32_t)run;
  }
  if (run >= ((uint64_t)1 << 31)) return fail(EOGS_ERR_OVERFLOW, "num_rendered overflows 31 bits");
  *num_rendered = (int64_t)run;
  return EOGS_OK;
}

/* stable merge sort of (key,value) pairs by key — the contract of cub::DeviceRadixSort::SortPairs
 * (rasterizer_impl.cu:306-311): ascending, stable. Tile ids occupy bits [32, 32+bit), so sorting on
 * the full 64-bit key equals sorting on bits [0, 32+bit). */
static void merge_sort_pairs(uint64_t* k, uint32_t* v, uint64_t* kt, uint32_t* vt, size_t n) {
  for (size_t w = 1; w < n; w *= 2) {
    for (size_t lo = 0; lo < n; lo += 2 * w) {
      size_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
      size_t i = lo, j = mid, o = lo;
      while (i < mid && j < hi) {
        if (k[j] < k[i]) { kt[o] = k[j]; vt[o++] = v[j++]; }
        else { kt[o] = k[i]; vt[o++] = v[i++]; }
      }
      while (i < mid) { kt[o] = k[i]; vt[o++] = v[i++]; }
      while (j < hi) { kt[o] = k[j]; vt[o++] = v[j++]; }
    }
    memcpy(k, kt, n * 8);
    memcpy(v, vt, n * 4);
  }
}


/* ------------------------------------------------------------------------------------------- */
/* Forward phase 2: duplicateWithKeys, sort, identifyTileRanges, FORWARD::renderCUDA           */
/* DGR/cuda_rasterizer/rasterizer_impl.cu:70-138,290-340; forward.cu:288-411                    */
/* ------------------------------------------------------------------------------------------- */
int eogs_rast_forward_render(
    int P, int H, int W, int64_t R,
    const float* bg, unsigned flags,
    void* geom, size_t geom_bytes, void* binning, size_t binning_bytes,
    void* image, size_t image_bytes, void* scratch, size_t scratch_bytes,
    float* out_color, float* out_invdepth, void* stream) {
  (void)stream; (void)scratch; (void)scratch_bytes;
  g_err[0] = 0;
  if (flags & EOGS_FLAG_ALT_ONLY)
    return fail(EOGS_ERR_INVALID_ARG, "forward_render: EOGS_FLAG_ALT_ONLY is not restated by the CPU oracle");
  if (P < 0 || H <= 0 || W <= 0 || R < 0 || !out_color || !bg || !image)
    return fail(EOGS_ERR_INVALID_ARG, "forward_render: bad argument");
  if (image_bytes < image_layout(NULL, H, W, NULL)) return fail(EOGS_ERR_WORKSPACE, "forward_render: image workspace too small");
  if (P > 0 && (!geom || geom_bytes < geom_layout(NULL, P, NULL))) return fail(EOGS_ERR_WORKSPACE, "forward_render: geom workspace too small");
  if (R > 0 && (!binning || binning_bytes < binning_layout(NULL, R, NULL)))
    return fail(EOGS_ERR_WORKSPACE, "forward_render: binning workspace too small");
  Geom g; Binning b; Image im;
  memset(&g, 0, sizeof g); memset(&b, 0, sizeof b);
  if (P > 0) geom_layout((char*)geom, P, &g);
  if (R > 0) binning_layout((char*)binning, R, &b);
  image_layout((char*)image, H, W, &im);
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  const size_t HW = (size_t)H * W;
  const float* colors = g.colors;

  /* duplicateWithKeys (rasterizer_impl.cu:70-111): key = tile << 32 | depth bits, row-major emission. */
  for (int idx = 0; idx < P && R > 0; idx++) {
    if (!(g.radii[idx] > 0)) continue;
    uint32_t off = idx == 0 ? 0 : g.point_offsets[idx - 1];
    int x0, y0, x1, y1;
    get_rect(g.means2D[2 * (size_t)idx], g.means2D[2 * (size_t)idx + 1], g.radii[idx], gx, gy, &x0, &y0, &x1, &y1);
    uint32_t dbits;
    memcpy(&dbits, &g.depths[idx], 4);
    for (int y = y0; y < y1; y++)
      for (int x = x0; x < x1; x++) {
        uint64_t key = (uint64_t)(y * gx + x);
        key <<= 32;
        key |= dbits;
        b.keys[off] = key;
        b.values[off] = (uint32_t)idx;
        off++;
      }
  }
  if (R > 0) merge_sort_pairs(b.keys, b.values, b.keys_tmp, b.values_tmp, (size_t)R);

  /* cudaMemset + identifyTileRanges (rasterizer_impl.cu:313-320,116-138) */
  memset(im.ranges, 0, (size_t)gx * gy * 8);
  for (int64_t i = 0; i < R; i++) {
    uint32_t cur = (uint32_t)(b.keys[i] >> 32);
    if (i == 0) im.ranges[2 * cur] = 0;
    else {
      uint32_t prev = (uint32_t)(b.keys[i - 1] >> 32);
      if (cur != prev) { im.ranges[2 * prev + 1] = (uint32_t)i; im.ranges[2 * cur] = (uint32_t)i; }
    }
    if (i == R - 1) im.ranges[2 * cur + 1] = (uint32_t)R;
  }

  /* FORWARD::renderCUDA (forward.cu:288-411), one pixel at a time; the block-level early exit
   * (:340-342) only skips work for pixels that are all `done`, so it has no effect on results.
   * Rows are dealt to threads (OpenMP, eogs_oracle_set_threads): a pixel reads shared data and writes only its own outputs, so
   * the results are those of the serial loop bit for bit. */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(oracle_threads())
#endif
  for (int py = 0; py < H; py++)
    for (int px = 0; px < W; px++) {
      const uint32_t tile = (uint32_t)((py / TILE) * gx + (px / TILE));
      const uint32_t r0 = im.ranges[2 * tile], r1 = im.ranges[2 * tile + 1];
      const size_t pix_id = (size_t)W * py + px;
      const float pixfx = (float)px, pixfy = (float)py;
      float T = 1.0f;
      uint32_t contributor = 0, last_contributor = 0, nblended = 0;
      float Cc[C_] = {0};
      float expected_invdepth = 0.0f;
      const int ns = nudge_sign(pix_id); /* 0 unless a test moved the thresholds (eogs_oracle_threshold_nudge) */
      for (uint32_t k = r0; k < r1; k++) {
        contributor++;
        const uint32_t id = b.values[k];
        const float dx = g.means2D[2 * (size_t)id] - pixfx, dy = g.means2D[2 * (size_t)id + 1] - pixfy;
        const float* co = g.conic_opacity + 4 * (size_t)id;
        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0.0f) continue;
        const float alpha = fminf(0.99f, co[3] * expf(power));
        if (alpha < alpha_min(ns, co, dx, dy)) continue;
        const float test_T = T * (1 - alpha);
        if (test_T < T_min(ns, nblended)) break; /* done = true: this Gaussian is not blended */
        for (int ch = 0; ch < C_; ch++) Cc[ch] += colors[(size_t)id * C_ + ch] * alpha * T;
        expected_invdepth += (1 / g.depths[id]) * alpha * T;
        T = test_T;
        last_contributor = contributor;
        nblended++;
      }
      im.final_T[pix_id] = T;
      im.n_contrib[pix_id] = last_contributor;
      for (int ch = 0; ch < C_; ch++) out_color[ch * HW + pix_id] = Cc[ch] + T * bg[ch];
      if (out_invdepth) out_invdepth[pix_id] = expected_invdepth;
    }
  return EOGS_OK;
}

/* ------------------------------------------------------------------------------------------- */
/* Backward: BACKWARD::renderCUDA, computeCov2DCUDA, BACKWARD::preprocessCUDA + computeCov3D   */
/* DGR/cuda_rasterizer/backward.cu:457-643, 147-327, 399-454, 331-394                          */
/* ------------------------------------------------------------------------------------------- */
static int backward_activated(
    int P, int H, int W, int64_t R,
    const float* bg, const float* means3D, const int* radii, const float* colors,
    const float* opacities, const float* scales, const float* rotations,
    float scale_modifier, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, unsigned flags,
    const float* out_color, const float* out_invdepth,
    const float* dL_dout_color, const float* dL_dout_invdepth,
    const void* geom, size_t geom_bytes, const void* binning, size_t binning_bytes,
    const void* image, size_t image_bytes,
    float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity,
    float* dL_dmeans3D, float* dL_dcov3D, float* dL_dscales, float* dL_drotations,
    float* dL_dT_sum, float* dL_dvm_mean, void* stream) {
  (void)stream;
  g_err[0] = 0;
  if (P < 0 || H <= 0 || W <= 0 || R < 0) return fail(EOGS_ERR_INVALID_ARG, "backward: bad sizes");
  if (dL_dT_sum) memset(dL_dT_sum, 0, 6 * 4);
  if (dL_dvm_mean) memset(dL_dvm_mean, 0, 12 * 4);
  if (P == 0) return EOGS_OK;
  if (!bg || !means3D || !radii || !colors || !opacities || !viewmatrix || !projmatrix || !dL_dout_color ||
      !geom || !image || !dL_dmeans2D || !dL_dcolors || !dL_dopacity || !dL_dmeans3D || !dL_dcov3D)
    return fail(EOGS_ERR_INVALID_ARG, "backward: NULL argument");
  const int have_sr = scales && rotations;
  if (have_sr == (cov3D_precomp != NULL)) return fail(EOGS_ERR_INVALID_ARG, "backward: scale/rotation xor cov3D_precomp");
  if (have_sr && (!dL_dscales || !dL_drotations)) return fail(EOGS_ERR_INVALID_ARG, "backward: NULL scale/rotation gradient");
  if (geom_bytes < geom_layout(NULL, P, NULL) || image_bytes < image_layout(NULL, H, W, NULL) ||
      (R > 0 && (!binning || binning_bytes < binning_layout(NULL, R, NULL))))
    return fail(EOGS_ERR_WORKSPACE, "backward: workspace too small");
  Geom g; Binning b; Image im;
  memset(&b, 0, sizeof b);
  geom_layout((char*)geom, P, &g);
  if (R > 0) binning_layout((char*)binning, R, &b);
  image_layout((char*)image, H, W, &im);
  const int gx = (W + TILE - 1) / TILE;
  const size_t HW = (size_t)H * W, n = (size_t)P;
  const int aa = (flags & EOGS_FLAG_ANTIALIASING) != 0;

  /* double accumulators for the atomically-summed outputs (see header).
   * The image is cut into ORACLE_BANDS bands of rows, each with accumulators of its own; a band sums its pixels in row-major
   * order and the bands are added in ascending order afterwards. The number of bands is a constant — never the number of
   * threads that happen to work on them — so the result does not depend on the machine: one more fixed order of the sums the
   * reference leaves to its atomics (backward.cu:598-640), and what lets the checker's backward use the host's cores (bands
   * are dealt to OpenMP threads, eogs_oracle_set_threads). With fp32 accumulation switched on (eogs_oracle_accum_float: the
   * sensitivity measurement "one sequence of atomicAdds") there is ONE band: a sequence, not a tree. */
  const int nbands = g_acc_float ? 1 : ORACLE_BANDS;
  const size_t per_band = n * (2 + 3 + 1 + C_ + 1);
  double* acc_all = (double*)calloc(per_band * (size_t)nbands, 8);
  if (!acc_all) return fail(EOGS_ERR_DEVICE, "backward: out of host memory");
  double* acc_mean2D = acc_all;                 /* band 0: after the merge below, the totals */
  double* acc_conic = acc_mean2D + n * 2;       /* x, y, w */
  double* acc_opac = acc_conic + n * 3;
  double* acc_color = acc_opac + n;
  /* (after the colours: dL_dinvdepths, n doubles per band — computed, consumed by nobody: backward.cu:306-307 is commented out) */

  /* The per-Gaussian quantities the pixel loop and the per-Gaussian chain read, in `real`. The oracle proper takes them from the
   * forward's workspace (fp32, what the reference's backward re-reads: backward.cu:480-560); the arbiter build recomputes them in
   * double from the inputs (forward.cu:182-283 restated once more), Gaussians the forward did not list left at zero. */
  real* rG = (real*)calloc(n * 13, sizeof(real));
  if (!rG) { free(acc_all); return fail(EOGS_ERR_DEVICE, "backward: out of host memory"); }
  real* const r_mean2D = rG;          /* [P,2] pixel centre */
  real* const r_conic_o = rG + n * 2; /* [P,4] conic, opacity x antialiasing scale */
  real* const r_invd = rG + n * 6;    /* [P]   1 / depth */
  real* const r_cov3D = rG + n * 7;   /* [P,6] */
  real T[3][3];
  build_T_r(viewmatrix, W, H, T);
  for (size_t i = 0; i < n; i++) {
#ifdef ORACLE_F64
    if (!(radii[i] > 0)) continue;
    const float* pm = means3D + 3 * i;
    double pv[3];
    for (int k = 0; k < 3; k++)
      pv[k] = (double)viewmatrix[k] * pm[0] + (double)viewmatrix[4 + k] * pm[1] + (double)viewmatrix[8 + k] * pm[2] + (double)viewmatrix[12 + k];
    if (have_sr) cov3d_from_scale_rot_r(scales + 3 * i, scale_modifier, rotations + 4 * i, r_cov3D + 6 * i);
    else for (int k = 0; k < 6; k++) r_cov3D[6 * i + k] = cov3D_precomp[6 * i + k];
    double cx, cy, cz;
    cov2d_r(T, r_cov3D + 6 * i, &cx, &cy, &cz);
    const double det_cov = cx * cz - cy * cy;
    cx += (double)0.3f; cz += (double)0.3f; /* (the reference's constants are fp32 literals: the same function) */
    const double det = cx * cz - cy * cy;
    const double hcs = aa ? sqrt(fmax((double)0.000025f, det_cov / det)) : 1.0;
    r_conic_o[4 * i] = cz / det; r_conic_o[4 * i + 1] = -cy / det; r_conic_o[4 * i + 2] = cx / det;
    r_conic_o[4 * i + 3] = (double)opacities[i] * hcs;
    r_mean2D[2 * i] = ((pv[0] + 1.0) * W - 1.0) * 0.5;
    r_mean2D[2 * i + 1] = ((pv[1] + 1.0) * H - 1.0) * 0.5;
    r_invd[i] = 1.0 / (200.0 - pv[2]);
#else
    r_mean2D[2 * i] = g.means2D[2 * i]; r_mean2D[2 * i + 1] = g.means2D[2 * i + 1];
    for (int k = 0; k < 4; k++) r_conic_o[4 * i + k] = g.conic_opacity[4 * i + k];
    r_invd[i] = 1.f / g.depths[i];
    for (int k = 0; k < 6; k++) r_cov3D[6 * i + k] = g.cov3D[6 * i + k];
#endif
  }

  /* ---- BACKWARD::renderCUDA (backward.cu:457-643), back to front per pixel ---- */
  const real ddelx_dx = (real)(0.5 * W), ddely_dy = (real)(0.5 * H);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(oracle_threads() < nbands ? oracle_threads() : nbands)
#endif
  for (int band = 0; band < nbands; band++) {
    double* const acc_mean2D = acc_all + per_band * (size_t)band; /* (shadow the totals: this band's own accumulators) */
    double* const acc_conic = acc_mean2D + n * 2;
    double* const acc_opac = acc_conic + n * 3;
    double* const acc_color = acc_opac + n;
    double* const acc_invd = acc_color + n * C_;
    const int py0 = (int)((long long)H * band / nbands), py1 = (int)((long long)H * (band + 1) / nbands);
  for (int py = py0; py < py1; py++)
    for (int px = 0; px < W; px++) {
      const uint32_t tile = (uint32_t)((py / TILE) * gx + (px / TILE));
      const uint32_t r0 = im.ranges[2 * tile], r1 = im.ranges[2 * tile + 1];
      const size_t pix_id = (size_t)W * py + px;
      const float pixfx = (float)px, pixfy = (float)py;
      const uint32_t last_contributor = im.n_contrib[pix_id];
      real T_final = im.final_T[pix_id];
#ifndef ORACLE_F64
      if (g_pair_noise_ulps > 0.f) { /* the final transmittance of the NOISED alphas (eogs_oracle_pair_noise) */
        T_final = 1.0f;
        for (uint32_t k = r0; k < r1 && k - r0 < last_contributor; k++) {
          const uint32_t id = b.values[k];
          const float fdx = g.means2D[2 * (size_t)id] - pixfx, fdy = g.means2D[2 * (size_t)id + 1] - pixfy;
          const float* fco = g.conic_opacity + 4 * (size_t)id;
          const float fpower = -0.5f * (fco[0] * fdx * fdx + fco[2] * fdy * fdy) - fco[1] * fdx * fdy;
          if (fpower > 0.0f) continue;
          const float fG = expf(fpower);
          if (fminf(0.99f, fco[3] * fG) < alpha_min(nudge_sign(pix_id), fco, fdx, fdy)) continue;
          T_final *= 1.f - fminf(0.99f, fco[3] * noised_G(pix_id, id, fpower));
        }
      }
#endif
#ifdef ORACLE_F64
      { /* the pixel's final transmittance in double: the product over the entries the fp32 forward blended (its decisions) */
        T_final = 1.0;
        for (uint32_t k = r0; k < r1 && k - r0 < last_contributor; k++) {
          const uint32_t id = b.values[k];
          const float fdx = g.means2D[2 * (size_t)id] - pixfx, fdy = g.means2D[2 * (size_t)id + 1] - pixfy;
          const float* fco = g.conic_opacity + 4 * (size_t)id;
          const float fpower = -0.5f * (fco[0] * fdx * fdx + fco[2] * fdy * fdy) - fco[1] * fdx * fdy;
          if (fpower > 0.0f) continue;
          if (fminf(0.99f, fco[3] * expf(fpower)) < alpha_min(nudge_sign(pix_id), fco, fdx, fdy)) continue;
          const real dx = r_mean2D[2 * (size_t)id] - (real)pixfx, dy = r_mean2D[2 * (size_t)id + 1] - (real)pixfy;
          const real* co = r_conic_o + 4 * (size_t)id;
          T_final *= 1.0 - fmin((double)0.99f, co[3] * exp(-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy));
        }
      }
#endif
      real Tcur = T_final;
      uint32_t contributor = r1 - r0;
      real accum_rec[C_] = {0}, dL_dpixel[C_], last_color[C_] = {0};
      real dL_invdepth = 0, accum_invdepth_rec = 0, last_invdepth = 0, last_alpha = 0;
      for (int ch = 0; ch < C_; ch++) dL_dpixel[ch] = dL_dout_color[ch * HW + pix_id];
      if (dL_dout_invdepth) dL_invdepth = dL_dout_invdepth[pix_id];
      real bg_dot_dpixel = 0;
      for (int ch = 0; ch < C_; ch++) bg_dot_dpixel += (real)bg[ch] * dL_dpixel[ch];

      float* alt = NULL; /* diagnostic: dL/dalpha per list entry in the front-to-back formulation (fp32) */
      real proj_a = 0, proj_T = T_final; /* diagnostic 3: the projected recursion's state */
      if (g_suffix_by_subtraction && g_suffix_by_subtraction != 3 && out_color && last_contributor > 0) {
        alt = (float*)calloc(r1 - r0, 4);
        float Dfinal = 0.f, Dacc = 0.f, Tf = 1.0f;
        for (int ch = 0; ch < C_; ch++) Dfinal += (float)dL_dpixel[ch] * out_color[ch * HW + pix_id];
        if (dL_dout_invdepth && out_invdepth) Dfinal += (float)dL_invdepth * out_invdepth[pix_id];
        if (g_suffix_by_subtraction == 2) { /* the running sum's own total: same operations, same order as the loop below */
          float Dt = 0.f, Tt = 1.0f;
          for (uint32_t k = r0; k < r1 && k - r0 < last_contributor; k++) {
            const uint32_t id = b.values[k];
            const float dx = g.means2D[2 * (size_t)id] - pixfx, dy = g.means2D[2 * (size_t)id + 1] - pixfy;
            const float* co = g.conic_opacity + 4 * (size_t)id;
            const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
            if (power > 0.0f) continue;
            const float alpha = fminf(0.99f, co[3] * expf(power));
            if (alpha < alpha_min(nudge_sign(pix_id), co, dx, dy)) continue;
            float gc = 0.f;
            for (int ch = 0; ch < C_; ch++) gc += (float)dL_dpixel[ch] * colors[(size_t)id * C_ + ch];
            if (dL_dout_invdepth) gc += (float)dL_invdepth * (1.f / g.depths[id]);
            Dt += gc * (alpha * Tt);
            Tt *= (1.f - alpha);
          }
          Dfinal = Dt + (float)T_final * (float)bg_dot_dpixel;
        }
        for (uint32_t k = r0; alt && k < r1 && k - r0 < last_contributor; k++) {
          const uint32_t id = b.values[k];
          const float dx = g.means2D[2 * (size_t)id] - pixfx, dy = g.means2D[2 * (size_t)id + 1] - pixfy;
          const float* co = g.conic_opacity + 4 * (size_t)id;
          const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > 0.0f) continue;
          const float alpha = fminf(0.99f, co[3] * expf(power));
          if (alpha < alpha_min(nudge_sign(pix_id), co, dx, dy)) continue;
          float gc = 0.f;
          for (int ch = 0; ch < C_; ch++) gc += (float)dL_dpixel[ch] * colors[(size_t)id * C_ + ch];
          if (dL_dout_invdepth) gc += (float)dL_invdepth * (1.f / g.depths[id]);
          Dacc += gc * (alpha * Tf);
          alt[k - r0] = Tf * gc - (Dfinal - Dacc) / (1.f - alpha);
          Tf *= (1.f - alpha);
        }
      }

      for (uint32_t k = r1; k-- > r0;) {
        contributor--;
        if (contributor >= last_contributor) continue;
        const uint32_t id = b.values[k];
        /* the decisions: the forward's own fp32 evaluation (forward.cu:366-376), in every build */
        const float fdx = g.means2D[2 * (size_t)id] - pixfx, fdy = g.means2D[2 * (size_t)id + 1] - pixfy;
        const float* fco = g.conic_opacity + 4 * (size_t)id;
        const float fpower = -0.5f * (fco[0] * fdx * fdx + fco[2] * fdy * fdy) - fco[1] * fdx * fdy;
        if (fpower > 0.0f) continue;
        const float fG = expf(fpower);
        const float falpha = fminf(0.99f, fco[3] * fG);
        if (falpha < alpha_min(nudge_sign(pix_id), fco, fdx, fdy)) continue; /* the same decision forward took */
#ifdef ORACLE_F64
        const real dx = r_mean2D[2 * (size_t)id] - (real)pixfx, dy = r_mean2D[2 * (size_t)id + 1] - (real)pixfy;
        const real* co = r_conic_o + 4 * (size_t)id;
        const real power = -0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        const real G = exp(power);
        const real alpha = fmin((double)0.99f, co[3] * G);
#else
        const real dx = fdx, dy = fdy;
        const real* co = fco;
        const real G = g_pair_noise_ulps > 0.f ? noised_G(pix_id, id, fpower) : fG;
        const real alpha = g_pair_noise_ulps > 0.f ? fminf(0.99f, fco[3] * G) : falpha;
#endif

        Tcur = Tcur / (RC(1) - alpha);
        const real dchannel_dcolor = alpha * Tcur;
        real dL_dalpha = 0;
        for (int ch = 0; ch < C_; ch++) {
          const real c = colors[(size_t)id * C_ + ch];
          accum_rec[ch] = last_alpha * last_color[ch] + (RC(1) - last_alpha) * accum_rec[ch];
          last_color[ch] = c;
          const real dL_dchannel = dL_dpixel[ch];
          dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
          ACC(acc_color[(size_t)id * C_ + ch], dchannel_dcolor * dL_dchannel);
        }
        if (dL_dout_invdepth) {
          const real invd = r_invd[id];
          accum_invdepth_rec = last_alpha * last_invdepth + (RC(1) - last_alpha) * accum_invdepth_rec;
          last_invdepth = invd;
          dL_dalpha += (invd - accum_invdepth_rec) * dL_invdepth;
          ACC(acc_invd[id], dchannel_dcolor * dL_invdepth);
        }
        dL_dalpha *= Tcur;
        last_alpha = alpha;
        dL_dalpha += (-T_final / (RC(1) - alpha)) * bg_dot_dpixel;
        if (alt) dL_dalpha = alt[k - r0];
        if (g_suffix_by_subtraction == 3) {
          real gc = 0;
          for (int ch = 0; ch < C_; ch++) gc += dL_dpixel[ch] * (real)colors[(size_t)id * C_ + ch];
          if (dL_dout_invdepth) gc += dL_invdepth * r_invd[id];
          const real rinv = RC(1) / (RC(1) - alpha);
          proj_T = proj_T * rinv;
          const real d = gc - proj_a;
          dL_dalpha = d * proj_T + (-T_final * bg_dot_dpixel) * rinv;
          proj_a = proj_a + alpha * d;
        }

        const real dL_dG = co[3] * dL_dalpha;
        const real gdx = G * dx, gdy = G * dy;
        const real dG_ddelx = -gdx * co[0] - gdy * co[1];
        const real dG_ddely = -gdy * co[2] - gdx * co[1];
        ACC(acc_mean2D[2 * (size_t)id], dL_dG * dG_ddelx * ddelx_dx);
        ACC(acc_mean2D[2 * (size_t)id + 1], dL_dG * dG_ddely * ddely_dy);
        ACC(acc_conic[3 * (size_t)id], RC(-0.5) * gdx * dx * dL_dG);
        ACC(acc_conic[3 * (size_t)id + 1], RC(-0.5) * gdx * dy * dL_dG);
        ACC(acc_conic[3 * (size_t)id + 2], RC(-0.5) * gdy * dy * dL_dG);
        ACC(acc_opac[id], G * dL_dalpha);
      }
      free(alt);
    }
  }
  /* the bands' sums, in ascending band order, into band 0 (element-wise: any number of threads gives the same sums) */
  for (int band = 1; band < nbands; band++) {
    const double* src = acc_all + per_band * (size_t)band;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(oracle_threads())
#endif
    for (long long i = 0; i < (long long)per_band; i++) acc_all[i] += src[i];
  }

  /* outputs zero-initialised like rasterize_points.cu:163-174 */
  memset(dL_dmeans2D, 0, n * 3 * 4);
  memset(dL_dmeans3D, 0, n * 3 * 4);
  memset(dL_dcov3D, 0, n * 6 * 4);
  if (dL_dscales) memset(dL_dscales, 0, n * 3 * 4);
  if (dL_drotations) memset(dL_drotations, 0, n * 4 * 4);
  for (size_t i = 0; i < n; i++) {
    dL_dmeans2D[3 * i] = (float)acc_mean2D[2 * i];
    dL_dmeans2D[3 * i + 1] = (float)acc_mean2D[2 * i + 1];
    dL_dopacity[i] = (float)acc_opac[i];
    for (int ch = 0; ch < C_; ch++) dL_dcolors[i * C_ + ch] = (float)acc_color[i * C_ + ch];
  }

  double dT_sum[6] = {0}, vm_mean[12] = {0};

  for (int idx = 0; idx < P; idx++) {
    if (!(radii[idx] > 0)) continue;
    const size_t i = (size_t)idx;
    /* ---- computeCov2DCUDA (backward.cu:147-327) ---- */
    const real* c3 = r_cov3D + 6 * i;
    const real dLc[3] = {(real)acc_conic[3 * i], (real)acc_conic[3 * i + 1], (real)acc_conic[3 * i + 2]};
    real V[3][3];
    sym_from6_r(c3, V);
    real c_xx, c_xy, c_yy;
    cov2d_r(T, c3, &c_xx, &c_xy, &c_yy);
    const real h_var = RC(0.3f);
    real d_inside_root = 0;
    if (aa) {
      const real det_cov = c_xx * c_yy - c_xy * c_xy;
      c_xx += h_var;
      c_yy += h_var;
      const real det_cov_plus_h_cov = c_xx * c_yy - c_xy * c_xy;
      const real hcs = R_SQRT(R_FMAX(RC(0.000025f), det_cov / det_cov_plus_h_cov));
      const real dL_dopacity_v = (real)acc_opac[i];
      const real d_hcs = dL_dopacity_v * (real)opacities[i];
#ifdef ORACLE_F64
      dL_dopacity[i] = (float)(dL_dopacity_v * hcs);
#else
      dL_dopacity[i] = dL_dopacity[i] * hcs; /* (the fp32 sum of the kernel above, rescaled in place: backward.cu:201-237) */
#endif
      d_inside_root = (det_cov / det_cov_plus_h_cov) <= RC(0.000025f) ? RC(0) : d_hcs / (2 * hcs);
    } else {
      c_xx += h_var;
      c_yy += h_var;
    }
    real dL_dc_xx = 0, dL_dc_xy = 0, dL_dc_yy = 0;
    if (aa) {
      const real x = c_xx, y = c_yy, z = c_xy, w = h_var;
      const real sqv = w * w + w * (x + y) + x * y - z * z;
      const real denom_f = d_inside_root / (sqv * sqv);
      dL_dc_xx = w * (w * y + y * y + z * z) * denom_f;
      dL_dc_yy = w * (w * x + x * x + z * z) * denom_f;
      dL_dc_xy = RC(-2) * w * z * (w + x + y) * denom_f;
    }
    const real denom = c_xx * c_yy - c_xy * c_xy;
    const real denom2inv = RC(1) / ((denom * denom) + RC(0.0000001f));
    real dcov[6] = {0, 0, 0, 0, 0, 0};
    if (denom2inv != 0) {
      dL_dc_xx += denom2inv * (-c_yy * c_yy * dLc[0] + 2 * c_xy * c_yy * dLc[1] + (denom - c_xx * c_yy) * dLc[2]);
      dL_dc_yy += denom2inv * (-c_xx * c_xx * dLc[2] + 2 * c_xx * c_xy * dLc[1] + (denom - c_xx * c_yy) * dLc[0]);
      dL_dc_xy += denom2inv * 2 * (c_xy * c_yy * dLc[0] - (denom + 2 * c_xy * c_xy) * dLc[1] + c_xx * c_xy * dLc[2]);
      dcov[0] = (T[0][0] * T[0][0] * dL_dc_xx + T[0][0] * T[1][0] * dL_dc_xy + T[1][0] * T[1][0] * dL_dc_yy);
      dcov[3] = (T[0][1] * T[0][1] * dL_dc_xx + T[0][1] * T[1][1] * dL_dc_xy + T[1][1] * T[1][1] * dL_dc_yy);
      dcov[5] = (T[0][2] * T[0][2] * dL_dc_xx + T[0][2] * T[1][2] * dL_dc_xy + T[1][2] * T[1][2] * dL_dc_yy);
      dcov[1] = 2 * T[0][0] * T[0][1] * dL_dc_xx + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_dc_xy + 2 * T[1][0] * T[1][1] * dL_dc_yy;
      dcov[2] = 2 * T[0][0] * T[0][2] * dL_dc_xx + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_dc_xy + 2 * T[1][0] * T[1][2] * dL_dc_yy;
      dcov[4] = 2 * T[0][2] * T[0][1] * dL_dc_xx + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_dc_xy + 2 * T[1][1] * T[1][2] * dL_dc_yy;
    }
    for (int k = 0; k < 6; k++) dL_dcov3D[6 * i + k] = (float)dcov[k];
    /* dL/dT (2x3), backward.cu:276-287.  Vrk[a][b] is symmetric so glm's [col][row] order is immaterial. */
    real TV[2][3];
    for (int r = 0; r < 2; r++)
      for (int k = 0; k < 3; k++) TV[r][k] = T[r][0] * V[k][0] + T[r][1] * V[k][1] + T[r][2] * V[k][2];
    const real dLdT[6] = {
        2 * TV[0][0] * dL_dc_xx + TV[1][0] * dL_dc_xy,
        2 * TV[0][1] * dL_dc_xx + TV[1][1] * dL_dc_xy,
        2 * TV[0][2] * dL_dc_xx + TV[1][2] * dL_dc_xy,
        2 * TV[1][0] * dL_dc_yy + TV[0][0] * dL_dc_xy,
        2 * TV[1][1] * dL_dc_yy + TV[0][1] * dL_dc_xy,
        2 * TV[1][2] * dL_dc_yy + TV[0][2] * dL_dc_xy};
    /* The reference stores these at dL_dT[idx+k] (backward.cu:320-325), which races between
     * neighbouring threads; the intended layout is 6*idx+k and only its sum over idx is consumed
     * (__init__.py:179-190). We produce that sum. */
    for (int k = 0; k < 6; k++) dT_sum[k] += (double)dLdT[k];

    /* ---- BACKWARD::preprocessCUDA (backward.cu:399-454): dL_dmeans = 0 (cov2D kernel, :313-317) + A^T g ---- */
#ifdef ORACLE_F64
    const real gxn = acc_mean2D[2 * i], gyn = acc_mean2D[2 * i + 1];
#else
    const real gxn = dL_dmeans2D[3 * i], gyn = dL_dmeans2D[3 * i + 1];
#endif
    dL_dmeans3D[3 * i + 0] = (float)((real)projmatrix[0] * gxn + (real)projmatrix[1] * gyn);
    dL_dmeans3D[3 * i + 1] = (float)((real)projmatrix[4] * gxn + (real)projmatrix[5] * gyn);
    dL_dmeans3D[3 * i + 2] = (float)((real)projmatrix[8] * gxn + (real)projmatrix[9] * gyn);

    /* ---- computeCov3D backward (backward.cu:331-394) ---- */
    if (have_sr) {
      const real q[4] = {rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2], rotations[4 * i + 3]};
      const real r = q[0], x = q[1], y = q[2], z = q[3];
      real Rm[3][3];
      quat_to_Rm_r(q, Rm);
      const real sm = scale_modifier;
      const real sc[3] = {sm * (real)scales[3 * i], sm * (real)scales[3 * i + 1], sm * (real)scales[3 * i + 2]};
      real M[3][3];
      for (int a = 0; a < 3; a++)
        for (int c = 0; c < 3; c++) M[a][c] = sc[a] * Rm[a][c];
      real dS[3][3] = {{dcov[0], RC(0.5) * dcov[1], RC(0.5) * dcov[2]},
                       {RC(0.5) * dcov[1], dcov[3], RC(0.5) * dcov[4]},
                       {RC(0.5) * dcov[2], RC(0.5) * dcov[4], dcov[5]}};
      /* dL_dM = 2 * M * dL_dSigma (glm) -> math: dM = 2 * M_math * dS  */
      real dM[3][3];
      for (int a = 0; a < 3; a++)
        for (int c = 0; c < 3; c++)
          dM[a][c] = (RC(2) * M[a][0]) * dS[0][c] + (RC(2) * M[a][1]) * dS[1][c] + (RC(2) * M[a][2]) * dS[2][c];
      /* glm: Rt = transpose(R), dL_dMt = transpose(dL_dM); Rt[k] (glm column k of R^T) = math row k of R_math;
       * dL_dMt[k] = math row k of dM. dL_dscale_k = dot(Rm[k,:], dM[k,:]). */
      float* ds = dL_dscales + 3 * i;
      for (int k = 0; k < 3; k++) ds[k] = (float)(Rm[k][0] * dM[k][0] + Rm[k][1] * dM[k][1] + Rm[k][2] * dM[k][2]);
      /* dL_dMt[k] *= s_k; then dL_dMt[a][b] (glm col a, row b) = s_a * dM[a][b] */
      real D[3][3];
      for (int a = 0; a < 3; a++)
        for (int c = 0; c < 3; c++) D[a][c] = dM[a][c] * sc[a];
      float* dq = dL_drotations + 4 * i;
      dq[0] = (float)(2 * z * (D[0][1] - D[1][0]) + 2 * y * (D[2][0] - D[0][2]) + 2 * x * (D[1][2] - D[2][1]));
      dq[1] = (float)(2 * y * (D[1][0] + D[0][1]) + 2 * z * (D[2][0] + D[0][2]) + 2 * r * (D[1][2] - D[2][1]) - 4 * x * (D[2][2] + D[1][1]));
      dq[2] = (float)(2 * x * (D[1][0] + D[0][1]) + 2 * r * (D[2][0] - D[0][2]) + 2 * z * (D[1][2] + D[2][1]) - 4 * y * (D[2][2] + D[0][0]));
      dq[3] = (float)(2 * r * (D[0][1] - D[1][0]) + 2 * x * (D[2][0] + D[0][2]) + 2 * y * (D[1][2] + D[2][1]) - 4 * z * (D[1][1] + D[0][0]));
    }
  }
  /* wrapper-side reductions (__init__.py:193-201) over ALL Gaussians (invisible ones contribute zeros) */
  for (size_t i = 0; i < n; i++) {
    const float* m = means3D + 3 * i;
#ifdef ORACLE_F64
    const double gm[3] = {acc_mean2D[2 * i], acc_mean2D[2 * i + 1], 0.0};
#else
    const float* gm = dL_dmeans2D + 3 * i;
#endif
    for (int a = 0; a < 3; a++)
      for (int c = 0; c < 3; c++) vm_mean[3 * a + c] += (double)m[a] * (double)gm[c];
    for (int c = 0; c < 3; c++) vm_mean[9 + c] += (double)gm[c];
  }
  if (dL_dT_sum) for (int k = 0; k < 6; k++) dL_dT_sum[k] = (float)dT_sum[k];
  if (dL_dvm_mean) for (int k = 0; k < 12; k++) dL_dvm_mean[k] = (float)vm_mean[k];

  free(rG);
  free(acc_all);
  return EOGS_OK;
}

/* ------------------------------------------------------------------------------------------- */
/* Exported forward_prepare / backward, with the opt-in EOGS_FLAG_RAW_PARAMS front end.          */
/* The raw-parameter mode restates, around the rasterizer above, the PyTorch ops the reference   */
/* runs before each render: gaussian_model.py:41,49,52,109-137 (exp / sigmoid / F.normalize),    */
/* gaussian_renderer/renderer.py:91-96 (SH2RGB, ECEF_to_UVA altitude, ones, cat) and             */
/* utils/sh_utils.py:25,125-126; their backward is the textbook chain rule, in double.           */
/* ------------------------------------------------------------------------------------------- */
#define SH_C0 0.28209479177387814

typedef struct {
  float *scales, *rotations, *opacities, *colors; /* activated copies */
} Activated;

static void activated_free(Activated* a) {
  free(a->scales); free(a->rotations); free(a->opacities); free(a->colors);
}

static int activate_raw(int P, const float* means3D, const float* log_scales, const float* raw_rot,
                        const float* logits, const float* f_dc, const float* alt, Activated* a) {
  const size_t n = (size_t)P;
  a->scales = (float*)malloc(n * 3 * 4);
  a->rotations = (float*)malloc(n * 4 * 4);
  a->opacities = (float*)malloc(n * 4);
  a->colors = f_dc ? (float*)malloc(n * C_ * 4) : NULL;
  if (!a->scales || !a->rotations || !a->opacities || (f_dc && !a->colors)) {
    activated_free(a);
    return 0;
  }
  for (size_t i = 0; i < n; i++) {
    for (int k = 0; k < 3; k++) a->scales[3 * i + k] = expf(log_scales[3 * i + k]);
    const float* r = raw_rot + 4 * i;
    const float nr = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    const float den = fmaxf(nr, 1e-12f);
    for (int k = 0; k < 4; k++) a->rotations[4 * i + k] = r[k] / den;
    a->opacities[i] = 1.f / (1.f + expf(-logits[i]));
    if (f_dc) {
      const float* p = means3D + 3 * i;
      for (int k = 0; k < 3; k++) a->colors[C_ * i + k] = (float)(f_dc[3 * i + k] * SH_C0 + 0.5);
      a->colors[C_ * i + 3] = p[0] * alt[0] + p[1] * alt[1] + p[2] * alt[2] + alt[3];
      a->colors[C_ * i + 4] = 1.f;
    }
  }
  return 1;
}

/* EOGS_FLAG_DEFER_COUNTS (include/eogs_rast.h): the oracle knows its count synchronously; it keeps it for
 * eogs_rast_forward_counts and hands back 0 as the ABI says. It never works from a capacity token: its
 * eogs_rast_capacity_token returns the exact token unchanged. */
static __thread int64_t g_deferred_token = -1;
static __thread int64_t g_last_token = -1; /* eogs_rast_read_counts: the forward that last ran on this thread */
static int forward_prepare_now(
    int P, int H, int W,
    const float* means3D, const float* scales, const float* rotations,
    const float* cov3D_precomp, const float* opacities, const float* colors, float scale_modifier,
    const float* viewmatrix, const float* projmatrix, const float* alt_affine, unsigned flags,
    int* radii, void* geom, size_t geom_bytes, int64_t* num_rendered, void* stream);

int eogs_rast_forward_prepare(
    int P, int H, int W,
    const float* means3D, const float* scales, const float* rotations,
    const float* cov3D_precomp, const float* opacities, const float* colors, float scale_modifier,
    const float* viewmatrix, const float* projmatrix, const float* alt_affine, unsigned flags,
    int* radii, void* geom, size_t geom_bytes, void* scratch, size_t scratch_bytes,
    int64_t* num_rendered, void* stream) {
  (void)scratch; (void)scratch_bytes;
  g_deferred_token = -1;
  /* The restatement follows the reference, which has no one-channel render: a caller that asked for one would hand over
   * [1,H,W] images and this code would write five planes into them. */
  if (flags & EOGS_FLAG_ALT_ONLY)
    return fail(EOGS_ERR_INVALID_ARG, "forward_prepare: EOGS_FLAG_ALT_ONLY is not restated by the CPU oracle (compare channel 3 of a full render)");
  const int rc = forward_prepare_now(P, H, W, means3D, scales, rotations, cov3D_precomp, opacities, colors, scale_modifier,
                                     viewmatrix, projmatrix, alt_affine, flags, radii, geom, geom_bytes, num_rendered, stream);
  g_last_token = rc == EOGS_OK ? *num_rendered : -1;
  if (rc == EOGS_OK && (flags & EOGS_FLAG_DEFER_COUNTS) && P > 0) {
    if (!(flags & EOGS_FLAG_NO_READBACK)) g_deferred_token = *num_rendered;
    *num_rendered = 0;
  }
  return rc;
}

/* (the oracle is synchronous and keeps no counts in `geom`: the token of this thread's last forward_prepare) */
int eogs_rast_read_counts(int P, int H, int W, const void* geom, size_t geom_bytes, int have_scratch, void* stream,
                          int64_t* num_rendered) {
  (void)H; (void)W; (void)geom_bytes; (void)have_scratch; (void)stream;
  g_err[0] = 0;
  if (P <= 0 || !geom || !num_rendered) return fail(EOGS_ERR_INVALID_ARG, "read_counts: bad argument");
  *num_rendered = 0;
  if (g_last_token < 0) return fail(EOGS_ERR_INVALID_ARG, "read_counts: no forward ran on this thread");
  *num_rendered = g_last_token;
  return EOGS_OK;
}

int eogs_rast_forward_counts(int64_t* num_rendered) {
  g_err[0] = 0;
  if (!num_rendered) return fail(EOGS_ERR_INVALID_ARG, "forward_counts: NULL argument");
  *num_rendered = 0;
  if (g_deferred_token < 0) return fail(EOGS_ERR_INVALID_ARG, "forward_counts: no forward_prepare pending on this thread");
  *num_rendered = g_deferred_token;
  g_deferred_token = -1;
  return EOGS_OK;
}

/* count mirror (include/eogs_rast.h): the oracle is synchronous — "arrived" as soon as mirror_counts has been called */
int eogs_rast_mirror_arm(void* host) {
  if (!host) return fail(EOGS_ERR_INVALID_ARG, "mirror_arm: NULL host buffer");
  ((int64_t*)host)[0] = -1;
  return EOGS_OK;
}
int eogs_rast_mirror_counts(int P, const void* geom, size_t geom_bytes, void* host, void* stream) {
  (void)geom_bytes; (void)stream;
  g_err[0] = 0;
  if (P <= 0 || !geom || !host) return fail(EOGS_ERR_INVALID_ARG, "mirror_counts: bad argument");
  ((int64_t*)host)[0] = g_last_token;
  return EOGS_OK;
}
int eogs_rast_mirror_token(int P, int H, int W, const void* host, int have_scratch, int64_t* num_rendered, int* arrived) {
  (void)H; (void)W; (void)have_scratch;
  g_err[0] = 0;
  if (P <= 0 || !host || !num_rendered || !arrived) return fail(EOGS_ERR_INVALID_ARG, "mirror_token: bad argument");
  *arrived = ((const int64_t*)host)[0] >= 0;
  if (*arrived) *num_rendered = ((const int64_t*)host)[0];
  return EOGS_OK;
}

int eogs_rast_capacity_token(int P, int64_t num_rendered, double slack, int have_scratch, int64_t exact, int64_t* capacity,
                             int* fits) {
  (void)have_scratch;
  if (P < 0 || num_rendered < 0 || !(slack >= 0.0) || !capacity) return fail(EOGS_ERR_INVALID_ARG, "capacity_token: bad argument");
  *capacity = num_rendered;
  if (fits) *fits = exact > 0 && exact == num_rendered;
  return EOGS_OK;
}

static int forward_prepare_now(
    int P, int H, int W,
    const float* means3D, const float* scales, const float* rotations,
    const float* cov3D_precomp, const float* opacities, const float* colors, float scale_modifier,
    const float* viewmatrix, const float* projmatrix, const float* alt_affine, unsigned flags,
    int* radii, void* geom, size_t geom_bytes, int64_t* num_rendered, void* stream) {
  if (!(flags & EOGS_FLAG_RAW_PARAMS) || P <= 0)
    return forward_prepare_activated(P, H, W, means3D, scales, rotations, cov3D_precomp, opacities, colors,
                                     scale_modifier, viewmatrix, projmatrix, flags, radii, geom, geom_bytes,
                                     num_rendered, stream);
  g_err[0] = 0;
  if (!colors) return fail(EOGS_ERR_NO_COLORS, "For non-RGB, provide precomputed Gaussian colors!");
  if (!means3D || !scales || !rotations || cov3D_precomp || !opacities || !alt_affine)
    return fail(EOGS_ERR_INVALID_ARG, "forward_prepare: EOGS_FLAG_RAW_PARAMS needs scales, rotations and alt_affine");
  Activated a;
  if (!activate_raw(P, means3D, scales, rotations, opacities, colors, alt_affine, &a))
    return fail(EOGS_ERR_DEVICE, "forward_prepare: out of host memory");
  const int rc = forward_prepare_activated(P, H, W, means3D, a.scales, a.rotations, NULL, a.opacities, a.colors,
                                           scale_modifier, viewmatrix, projmatrix, flags, radii, geom, geom_bytes,
                                           num_rendered, stream);
  activated_free(&a);
  return rc;
}

static int backward_full(
    int P, int H, int W, int64_t R,
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
    float* dL_dT_sum, float* dL_dvm_mean, void* stream) {
  if (!(flags & EOGS_FLAG_RAW_PARAMS) || P <= 0) {
    /* dL_dcov3D may be NULL when scales / rotations are given (the wrapper discards it then): computed into a temporary */
    float* d_cov = dL_dcov3D;
    if (!d_cov && P > 0 && scales && rotations && !(d_cov = (float*)malloc((size_t)P * 6 * 4)))
      return fail(EOGS_ERR_DEVICE, "backward: out of host memory");
    const int rc0 = backward_activated(P, H, W, R, bg, means3D, radii, colors, opacities, scales, rotations, scale_modifier,
                                       cov3D_precomp, viewmatrix, projmatrix, flags, out_color, out_invdepth, dL_dout_color,
                                       dL_dout_invdepth, geom, geom_bytes, binning, binning_bytes, image, image_bytes,
                                       dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, d_cov, dL_dscales,
                                       dL_drotations, dL_dT_sum, dL_dvm_mean, stream);
    if (d_cov != dL_dcov3D) free(d_cov);
    return rc0;
  }
  g_err[0] = 0;
  if (!means3D || !scales || !rotations || cov3D_precomp || !opacities || !alt_affine || !dL_dcolors ||
      !dL_dopacity || !dL_dmeans3D || !dL_dscales || !dL_drotations || !geom)
    return fail(EOGS_ERR_INVALID_ARG, "backward: EOGS_FLAG_RAW_PARAMS needs scales, rotations and alt_affine");
  if (geom_bytes < geom_layout(NULL, P, NULL)) return fail(EOGS_ERR_WORKSPACE, "backward: workspace too small");
  const size_t n = (size_t)P;
  Activated a;
  if (!activate_raw(P, means3D, scales, rotations, opacities, NULL, alt_affine, &a))
    return fail(EOGS_ERR_DEVICE, "backward: out of host memory");
  Geom g;
  geom_layout((char*)geom, P, &g); /* the activated [P,5] features were kept by forward */
  float* d_col5 = (float*)malloc(n * C_ * 4);
  float* d_cov = dL_dcov3D ? dL_dcov3D : (float*)malloc(n * 6 * 4);
  int rc = EOGS_ERR_DEVICE;
  if (d_col5 && d_cov)
    rc = backward_activated(P, H, W, R, bg, means3D, radii, g.colors, a.opacities, a.scales, a.rotations,
                            scale_modifier, NULL, viewmatrix, projmatrix, flags, out_color, out_invdepth,
                            dL_dout_color, dL_dout_invdepth, geom, geom_bytes, binning, binning_bytes, image,
                            image_bytes, dL_dmeans2D, d_col5, dL_dopacity, dL_dmeans3D, d_cov, dL_dscales,
                            dL_drotations, dL_dT_sum, dL_dvm_mean, stream);
  else
    fail(EOGS_ERR_DEVICE, "backward: out of host memory");
  if (rc == EOGS_OK) {
    for (size_t i = 0; i < n; i++) {
      for (int k = 0; k < 3; k++) {
        dL_dcolors[3 * i + k] = (float)(SH_C0 * (double)d_col5[C_ * i + k]);
        dL_dmeans3D[3 * i + k] = (float)((double)dL_dmeans3D[3 * i + k] + (double)alt_affine[k] * d_col5[C_ * i + 3]);
        dL_dscales[3 * i + k] = (float)((double)dL_dscales[3 * i + k] * a.scales[3 * i + k]);
      }
      const double o = a.opacities[i];
      dL_dopacity[i] = (float)((double)dL_dopacity[i] * o * (1.0 - o));
      const float* r = rotations + 4 * i;
      const double nr = sqrt((double)r[0] * r[0] + (double)r[1] * r[1] + (double)r[2] * r[2] + (double)r[3] * r[3]);
      const double den = nr > 1e-12 ? nr : 1e-12;
      double dot = 0;
      for (int k = 0; k < 4; k++) dot += (double)a.rotations[4 * i + k] * dL_drotations[4 * i + k];
      for (int k = 0; k < 4; k++)
        dL_drotations[4 * i + k] = (float)(((double)dL_drotations[4 * i + k] - a.rotations[4 * i + k] * dot) / den);
    }
  }
  free(d_col5);
  if (!dL_dcov3D) free(d_cov);
  activated_free(&a);
  return rc;
}

/* Range form of the backward (include/eogs_rast.h): the checker evaluates the whole backward into temporaries and copies
 * rows [p_begin, p_end) out — K times the work for K ranges, which only ever run at test sizes. */
static int backward_range_impl(
    int P, int H, int W, int64_t R,
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
    float* dL_dT_sum, float* dL_dvm_mean, int p_begin, int p_end, void* stream) {
  if (P < 0 || p_begin < 0 || p_end < p_begin || p_end > P || (p_begin % 256) != 0 || (p_end != P && (p_end % 256) != 0))
    return fail(EOGS_ERR_INVALID_ARG, "backward: the Gaussian range must lie in [0, P] with multiples of 256 as inner bounds");
  if (p_begin == 0 && p_end == P)
    return backward_full(P, H, W, R, bg, means3D, radii, colors, opacities, scales, rotations, scale_modifier,
                              cov3D_precomp, viewmatrix, projmatrix, alt_affine, flags, out_color, out_invdepth,
                              dL_dout_color, dL_dout_invdepth, geom, geom_bytes, binning, binning_bytes, image,
                              image_bytes, dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dscales,
                              dL_drotations, dL_dT_sum, dL_dvm_mean, stream);
  const size_t n = (size_t)P, ncol = (flags & EOGS_FLAG_RAW_PARAMS) ? 3 : C_;
  const size_t w[7] = {3, ncol, 1, 3, 6, 3, 4};
  float* dst[7] = {dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dscales, dL_drotations};
  float* tmp[7] = {0};
  int rc = EOGS_OK;
  for (int k = 0; k < 7; k++)
    if (dst[k] && !(tmp[k] = (float*)malloc(n * w[k] * 4 + 4))) rc = fail(EOGS_ERR_DEVICE, "backward: out of host memory");
  float tsum[6], vsum[12];
  if (rc == EOGS_OK)
    rc = backward_full(P, H, W, R, bg, means3D, radii, colors, opacities, scales, rotations, scale_modifier,
                            cov3D_precomp, viewmatrix, projmatrix, alt_affine, flags, out_color, out_invdepth,
                            dL_dout_color, dL_dout_invdepth, geom, geom_bytes, binning, binning_bytes, image, image_bytes,
                            tmp[0], tmp[1], tmp[2], tmp[3], tmp[4], tmp[5], tmp[6], dL_dT_sum ? tsum : NULL,
                            dL_dvm_mean ? vsum : NULL, stream);
  if (rc == EOGS_OK) {
    for (int k = 0; k < 7; k++)
      if (dst[k]) memcpy(dst[k] + (size_t)p_begin * w[k], tmp[k] + (size_t)p_begin * w[k], (size_t)(p_end - p_begin) * w[k] * 4);
    if (p_end == P) {
      if (dL_dT_sum) memcpy(dL_dT_sum, tsum, sizeof tsum);
      if (dL_dvm_mean) memcpy(dL_dvm_mean, vsum, sizeof vsum);
    }
  }
  for (int k = 0; k < 7; k++) free(tmp[k]);
  return rc;
}

/* The checker has one path: the reference's 16-px tiles. */
int eogs_rast_path_info(int P, int64_t num_rendered, int* list_block_px, int* fwd_kernel, int* bwd_kernel) {
  (void)P; (void)num_rendered;
  if (!list_block_px || !fwd_kernel || !bwd_kernel) return fail(EOGS_ERR_INVALID_ARG, "path_info: bad argument");
  *list_block_px = TILE; *fwd_kernel = -1; *bwd_kernel = -1;
  return EOGS_OK;
}

int eogs_rast_backward_info(int P, int64_t num_rendered, int* gaussian_bwd_wide) {  /* (no kernel builds here: -1) */
  (void)P; (void)num_rendered;
  if (!gaussian_bwd_wide) return fail(EOGS_ERR_INVALID_ARG, "backward_info: bad argument");
  *gaussian_bwd_wide = -1;
  return EOGS_OK;
}

/* checkFrustum (rasterizer_impl.cu:54-66): in_frustum's culling is commented out and the function
 * falls off its end (auxiliary.h:151-176); the intended value is "visible". */
int eogs_rast_mark_visible(int P, const float* means3D, const float* viewmatrix,
                           const float* projmatrix, uint8_t* present, void* stream) {
  (void)means3D; (void)viewmatrix; (void)projmatrix; (void)stream;
  g_err[0] = 0;
  if (P < 0 || (P > 0 && !present)) return fail(EOGS_ERR_INVALID_ARG, "mark_visible: bad argument");
  for (int i = 0; i < P; i++) present[i] = 1;
  return EOGS_OK;
}

/* diagnostics: no-ops in the oracle */
int eogs_rast_profile_enable(int on) { (void)on; return EOGS_OK; }
int eogs_rast_profile_select(unsigned slot_mask) { (void)slot_mask; return EOGS_OK; }
int eogs_rast_profile_reset(void) { return EOGS_OK; }
int eogs_rast_profile_slots(void) { return 0; }
int eogs_rast_profile_get(int slot, double* total_ms, int64_t* launches, const char** name) {
  (void)slot; (void)total_ms; (void)launches; (void)name;
  return fail(EOGS_ERR_INVALID_ARG, "profile_get: the oracle has no profile slots");
}
int eogs_rast_selftest(void* scratch, unsigned* failed, void* stream) {
  (void)scratch; (void)stream;
  if (failed) *failed = 0;
  return EOGS_OK;
}

/* Exported backward entry points: the computation above plus the optional second destination of the colour gradient's
 * leading columns (include/eogs_rast.h dL_dcolors_lead: a data-parallel caller's exchange buffer). */
static void copy_lead(int P, unsigned flags, const float* dL_dcolors, float* lead, int lead_cols, int p0, int p1) {
  const int ncol = (flags & EOGS_FLAG_RAW_PARAMS) ? 3 : C_;
  if (!lead || P <= 0) return;
  for (int i = p0; i < p1; i++)
    for (int k = 0; k < lead_cols; k++) lead[(size_t)i * lead_cols + k] = dL_dcolors[(size_t)i * ncol + k];
}
static int lead_ok(unsigned flags, const float* lead, int lead_cols) {
  const int ncol = (flags & EOGS_FLAG_RAW_PARAMS) ? 3 : C_;
  return !lead || (lead_cols > 0 && lead_cols <= ncol);
}

int eogs_rast_backward(
    int P, int H, int W, int64_t R,
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
    float* dL_dT_sum, float* dL_dvm_mean, float* dL_dcolors_lead, int lead_cols, void* stream) {
  if (!lead_ok(flags, dL_dcolors_lead, lead_cols)) return fail(EOGS_ERR_INVALID_ARG, "backward: bad lead_cols");
  const int rc = backward_full(P, H, W, R, bg, means3D, radii, colors, opacities, scales, rotations, scale_modifier,
                               cov3D_precomp, viewmatrix, projmatrix, alt_affine, flags, out_color, out_invdepth,
                               dL_dout_color, dL_dout_invdepth, geom, geom_bytes, binning, binning_bytes, image, image_bytes,
                               dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dscales, dL_drotations,
                               dL_dT_sum, dL_dvm_mean, stream);
  if (rc == EOGS_OK) copy_lead(P, flags, dL_dcolors, dL_dcolors_lead, lead_cols, 0, P);
  return rc;
}

int eogs_rast_backward_range(
    int P, int H, int W, int64_t R,
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
    float* dL_dT_sum, float* dL_dvm_mean, float* dL_dcolors_lead, int lead_cols, int p_begin, int p_end, void* stream) {
  if (!lead_ok(flags, dL_dcolors_lead, lead_cols)) return fail(EOGS_ERR_INVALID_ARG, "backward: bad lead_cols");
  const int rc = backward_range_impl(P, H, W, R, bg, means3D, radii, colors, opacities, scales, rotations, scale_modifier,
                                     cov3D_precomp, viewmatrix, projmatrix, alt_affine, flags, out_color, out_invdepth,
                                     dL_dout_color, dL_dout_invdepth, geom, geom_bytes, binning, binning_bytes, image,
                                     image_bytes, dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dscales,
                                     dL_drotations, dL_dT_sum, dL_dvm_mean, p_begin, p_end, stream);
  if (rc == EOGS_OK) copy_lead(P, flags, dL_dcolors, dL_dcolors_lead, lead_cols, p_begin, p_end);
  return rc;
}
