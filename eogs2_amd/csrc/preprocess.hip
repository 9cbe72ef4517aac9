// preprocess.hip — per-Gaussian kernels (one lane per Gaussian, wave64).
//
//  preprocess_fwd_kernel : affine projection, Sigma = R S^2 R^T, cov2D = T Sigma T^T + 0.3 I, conic,
//                          radius, tile rect, depth key.  Reference semantics:
//                          DGR/cuda_rasterizer/forward.cu:154-283 (+ auxiliary.h:40-55,70-78).
//                          Additionally: the exact alpha >= 1/255 hit mask over the internal 8x8 tiles of the rect,
//                          the 64-byte render record, the 32-byte binning record, per-workgroup pair counts and
//                          depth-key range (no atomics: reduced by pblock_scan_kernel).
//  gaussian_bwd_kernel   : sums the per-(tile,Gaussian) gradient records written by render_bwd, then
//                          dL/dconic -> dL/dcov2D -> dL/dSigma (+ dL/dT), dL/dmean3D, dL/dscale, dL/dquat.
//                          Reference semantics: DGR/cuda_rasterizer/backward.cu:147-327 (computeCov2DCUDA),
//                          :399-454 (preprocessCUDA), :331-394 (computeCov3D) fused into one kernel, plus the
//                          wrapper-side reductions of DGR/diff_gaussian_rasterization/__init__.py:179-201.
//
// HBM-bound, O(P): inputs arrive as the reference's AoS [P,3]/[P,4]/[P,5] rows; the 12-byte rows are staged
// through LDS so that their global loads are contiguous dwords per lane. Records are read in Gaussian-id order
// (a workgroup's 256 Gaussians own one contiguous region), flags first, then two records per trip.
#include "common.h"
#include <type_traits>

namespace {

struct Mat3 {
  float m[3][3];
};

// Rotation "math" matrix used by the reference: Rm[r][c] = R_glm[c][r] (forward.cu:126-137, glm is column-major).
__device__ inline Mat3 quat_to_Rm(float r, float x, float y, float z) {
  Mat3 R;
  R.m[0][0] = 1.f - 2.f * (y * y + z * z); R.m[1][0] = 2.f * (x * y - r * z); R.m[2][0] = 2.f * (x * z + r * y);
  R.m[0][1] = 2.f * (x * y + r * z); R.m[1][1] = 1.f - 2.f * (x * x + z * z); R.m[2][1] = 2.f * (y * z - r * x);
  R.m[0][2] = 2.f * (x * z - r * y); R.m[1][2] = 2.f * (y * z + r * x); R.m[2][2] = 1.f - 2.f * (x * x + y * y);
  return R;
}

// Sigma (upper triangle) = M^T M, M = diag(mod*s) Rm   (forward.cu:117-151)
__device__ inline void cov3d_from_scale_rot(const float s[3], float mod, const float q[4], float c6[6]) {
  Mat3 R = quat_to_Rm(q[0], q[1], q[2], q[3]);
  float sc[3] = {mod * s[0], mod * s[1], mod * s[2]};
  float M[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) M[i][j] = sc[i] * R.m[i][j];
  float S[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = i; j < 3; j++) S[i][j] = M[0][i] * M[0][j] + M[1][i] * M[1][j] + M[2][i] * M[2][j];
  c6[0] = S[0][0]; c6[1] = S[0][1]; c6[2] = S[0][2]; c6[3] = S[1][1]; c6[4] = S[1][2]; c6[5] = S[2][2];
}

// Trow[i][k] = vm[4k+i] * s_i, s = (W/2, H/2)   (forward.cu:93-102)
__device__ inline void build_T(const float* __restrict__ vm, int W, int H, float T[2][3]) {
  const float sx = (float)(W / 2.0), sy = (float)(H / 2.0);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    T[0][k] = vm[4 * k + 0] * sx;
    T[1][k] = vm[4 * k + 1] * sy;
  }
}

// cov2D = T Sigma T^T (2x2), evaluated as (T Sigma) T^T like glm's left-to-right product (forward.cu:109).
__device__ inline void cov2d(const float T[2][3], const float c6[6], float& cxx, float& cxy, float& cyy) {
  const float V[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
  float A[2][3];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int l = 0; l < 3; l++) A[i][l] = T[i][0] * V[0][l] + T[i][1] * V[1][l] + T[i][2] * V[2][l];
  cxx = A[0][0] * T[0][0] + A[0][1] * T[0][1] + A[0][2] * T[0][2];
  cxy = A[1][0] * T[0][0] + A[1][1] * T[0][1] + A[1][2] * T[0][2];
  cyy = A[1][0] * T[1][0] + A[1][1] * T[1][1] + A[1][2] * T[1][2];
}

__device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Cooperative load of `rows` consecutive 3-float rows (12 B, AoS) into LDS with dword-contiguous lanes, in two halves — the
// loads into registers, the LDS stores later: a kernel requests ALL its rows (and whatever else it can) before anything waits.
// (One function that loaded and stored element by element was what rounds 1-5 used: the compiler waits for each load inside its
// bounds check before it issues the next — five to eight serialised round trips at the head of every preprocess workgroup.)
__device__ inline void stage_rows3_load(const float* __restrict__ src, size_t row0, int rows, float (&r)[3]) {
  const int t = threadIdx.x, n = rows * 3;
  const float* p = src + row0 * 3;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int e = i * BLK + t;
    r[i] = e < n ? p[e] : 0.f;
  }
}
__device__ inline void stage_rows3_store(int rows, const float (&r)[3], float* s_dst) {
  const int t = threadIdx.x, n = rows * 3;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int e = i * BLK + t;
    if (e < n) s_dst[e] = r[i];
  }
}

constexpr float SH_C0 = 0.28209479177387814f;  // utils/sh_utils.py:25

__device__ inline float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// scaling_activation = exp (gaussian_model.py:41), rotation_activation = F.normalize, eps 1e-12 (gaussian_model.py:52)
__device__ inline float raw_activate(float s[3], float q[4]) {
#pragma unroll
  for (int k = 0; k < 3; k++) s[k] = expf(s[k]);
  const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float inv = 1.f / fmaxf(n, 1e-12f);
#pragma unroll
  for (int k = 0; k < 4; k++) q[k] *= inv;
  return inv;
}

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// the same sum on the DPP path (six VALU instructions, no LDS round trips; the total read from lane 63: every lane gets it)
__device__ inline float wave_sum_dpp(float v) {
#define GB_DPP(x, ctrl, rows) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, rows, 0xF, true))
  v += GB_DPP(v, 0x121, 0xF);  // row_ror:1
  v += GB_DPP(v, 0x122, 0xF);  // row_ror:2
  v += GB_DPP(v, 0x124, 0xF);  // row_ror:4
  v += GB_DPP(v, 0x128, 0xF);  // row_ror:8: every lane holds its row's sum
  v += GB_DPP(v, 0x142, 0xA);  // row_bcast15 into rows 1 and 3
  v += GB_DPP(v, 0x143, 0xC);  // row_bcast31 into rows 2 and 3
#undef GB_DPP
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// listed tiles from which a Gaussian's records are summed by its whole wave instead of by its own lane (gaussian_bwd_kernel)
#ifndef GB_COOP
#define GB_COOP 64u
#endif

}  // namespace

// RAW (EOGS_FLAG_RAW_PARAMS): scales/rotations/opacities/colors are the model's raw parameters; the activations and
// the [rgb, altitude, 1] feature assembly of renderer.py:72-96 happen here instead of in ~10 PyTorch kernels.
// Compiled for six waves per SIMD (80 VGPRs, 96 bytes of scratch per lane, on the rarely taken paths) instead of the four that its
// 123 VGPRs allow: the kernel waits 62 % of its wave time on memory, 52 -> 48 us (trained 56 -> 50); five waves changed
// nothing, eight spill in the main path (48 / 57). -DEOGS_PP_WAVES=n overrides.
#ifndef EOGS_PP_WAVES
#define EOGS_PP_WAVES 6
#endif
#define PP_ATTR __attribute__((amdgpu_waves_per_eu(EOGS_PP_WAVES, EOGS_PP_WAVES)))
template <bool RAW>
__global__ __launch_bounds__(BLK) PP_ATTR void preprocess_fwd_kernel(
    int P, int H, int W, int gx, int gy,
    const float* __restrict__ means3D, const float* __restrict__ scales, const float* __restrict__ rotations,
    const float* __restrict__ cov3D_precomp, const float* __restrict__ opacities, const float* __restrict__ colors,
    const float* __restrict__ vm, const float* __restrict__ alt, float scale_modifier, int antialiasing,
    int* __restrict__ radii, float4* __restrict__ packed, uint4* __restrict__ binfo, float4* __restrict__ bext,
    uint32_t* __restrict__ pblock,
    uint32_t* __restrict__ pbkey) {
  __shared__ float s_m[3 * BLK];
  __shared__ float s_s[3 * BLK];
  __shared__ float s_c[RAW ? 3 * BLK : 1];
  __shared__ uint32_t s_cnt[BLK / 64];
  const int t = threadIdx.x;
  const size_t row0 = (size_t)blockIdx.x * BLK;
  const int rows = (int)(((size_t)P - row0) < (size_t)BLK ? ((size_t)P - row0) : (size_t)BLK);
  float4 q_pre = make_float4(1.f, 0.f, 0.f, 0.f);
  float op_pre = 0.f;
  {
    // all the rows' loads first, then their LDS stores (see stage_rows3_load)
    float st_m[3], st_s[3] = {0.f, 0.f, 0.f}, st_c[3] = {0.f, 0.f, 0.f};
    stage_rows3_load(means3D, row0, rows, st_m);
    if (scales) stage_rows3_load(scales, row0, rows, st_s);
    if (RAW) stage_rows3_load(colors, row0, rows, st_c);
    // (... and the Gaussian's rotation and opacity with them, not behind the barrier / inside `area != 0`: two more dependent
    // round trips otherwise)
    if (t < rows) {
      if (!cov3D_precomp) q_pre = reinterpret_cast<const float4*>(rotations)[row0 + t];
      op_pre = opacities[row0 + t];
    }
    stage_rows3_store(rows, st_m, s_m);
    if (scales) stage_rows3_store(rows, st_s, s_s);
    if (RAW) stage_rows3_store(rows, st_c, s_c);
  }
  __syncthreads();

  const size_t idx = row0 + t;
  uint32_t my_tiles = 0, my_entries = 0, key_bits = 0, bkind = BK_RECT, my_err = 0;
  uint32_t op64 = 0;  // round(64 * opacity), for the mean pair opacity that picks the list granularity (api.hip)
  uint4 bi0 = make_uint4(0u, 0u, 0u, 0u);
  bool need_count = false;  // a BK_RECT / BK_SPANS footprint: its entries (and listed tiles) are counted behind the nest
  uint32_t c_kind = BK_RECT;
  SpanParams c_sp = {};
  int c_sx0 = 0, c_sy0 = 0, c_sx1 = 0, c_sy1 = 0;
  if (t < rows) {
    const float p[3] = {s_m[3 * t], s_m[3 * t + 1], s_m[3 * t + 2]};
    // transformPoint4x3 (auxiliary.h:70-78)
    float pv[3];
#pragma unroll
    for (int i = 0; i < 3; i++) pv[i] = vm[i] * p[0] + vm[4 + i] * p[1] + vm[8 + i] * p[2] + vm[12 + i];

    float c6[6];
    if (cov3D_precomp) {
      const float2* c2 = reinterpret_cast<const float2*>(cov3D_precomp + 6 * idx);
      float2 a = c2[0], b = c2[1], c = c2[2];
      c6[0] = a.x; c6[1] = a.y; c6[2] = b.x; c6[3] = b.y; c6[4] = c.x; c6[5] = c.y;
    } else {
      const float4 q = q_pre;
      float s[3] = {s_s[3 * t], s_s[3 * t + 1], s_s[3 * t + 2]};
      float qq[4] = {q.x, q.y, q.z, q.w};
      if (RAW) raw_activate(s, qq);
      cov3d_from_scale_rot(s, scale_modifier, qq, c6);
    }
    float T[2][3];
    build_T(vm, W, H, T);
    float cx, cy, cz;
    cov2d(T, c6, cx, cy, cz);

    const float h_var = 0.3f;
    const float det_cov = cx * cz - cy * cy;
    cx += h_var;
    cz += h_var;
    const float det = cx * cz - cy * cy;
    float hcs = 1.0f;
    if (antialiasing) hcs = sqrtf(fmaxf(0.000025f, det_cov / det));

    int radius = 0;
    if (det != 0.0f) {
      const float det_inv = 1.f / det;
      const float mid = 0.5f * (cx + cz);
      const float root = sqrtf(fmaxf(0.1f, mid * mid - det));
      const float lambda1 = mid + root, lambda2 = mid - root;
      const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
      // ndc2Pix in double (auxiliary.h:40-43)
      const float px = (float)((((double)pv[0] + 1.0) * W - 1.0) * 0.5);
      const float py = (float)((((double)pv[1] + 1.0) * H - 1.0) * 0.5);
      const int r = (int)my_radius;
      // getRect (auxiliary.h:45-55): truncating float->int division
      const int x0 = clampi((int)((px - r) / TILE), 0, gx);
      const int y0 = clampi((int)((py - r) / TILE), 0, gy);
      const int x1 = clampi((int)((px + r + TILE - 1) / TILE), 0, gx);
      const int y1 = clampi((int)((py + r + TILE - 1) / TILE), 0, gy);
      const int area = (x1 - x0) * (y1 - y0);
      if (area != 0) {
        radius = r;
        const float d = (float)(200.0 - (double)pv[2]);  // forward.cu:267
        if (d < 0) my_err = 0x80000000u;  // altitude > 200: reported through the workgroup's partials (no zero-initialised word)
        const float op_in = RAW ? sigmoidf(op_pre) : op_pre;
        const float ca = cz * det_inv, cb = -cy * det_inv, cc = cx * det_inv, op = op_in * hcs;
        op64 = (uint32_t)(fminf(fmaxf(op, 0.f), 1.f) * 64.f + 0.5f);  // (NaN -> 0)
        // Internal SUBX x SUBY tiles: the reference's 16-px tile rect clipped to the image, intersected with the
        // bounding box of the ellipse alpha >= 1/255 (q <= tau  =>  |dx| <= sqrt(tau cov_xx), |dy| <= sqrt(tau cov_yy),
        // cov = conic^-1 = the 2D covariance incl. the 0.3 dilation). Pixels outside that box skip this Gaussian in
        // the reference too (forward.cu:374-376), so dropping their tiles changes nothing; for low opacities the box
        // is much smaller than the 3-sigma rect (tau = 1.87 at opacity 0.01: 1.37 sigma).
        const int gsx = (W + SUBX - 1) / SUBX, gsy = (H + SUBY - 1) / SUBY;
        int sx0 = FX * x0, sy0 = FY * y0;
        int sx1 = FX * x1 < gsx ? FX * x1 : gsx, sy1 = FY * y1 < gsy ? FY * y1 : gsy;
        const float tau = 2.f * __logf(255.f * op);
        const float tau_m = tau + 1e-3f * (1.f + fabsf(tau));
        bool any = true;
        if (op < 1.f / 255.f) {
          any = false;  // alpha = op * G <= op < 1/255 at every pixel (G <= 1): never blended
        } else if (tau_m >= 0.f && tau_m < 3.0e38f) {
          const float ex = __builtin_amdgcn_sqrtf(tau_m * cx) * 1.001f + 1e-2f, ey = __builtin_amdgcn_sqrtf(tau_m * cz) * 1.001f + 1e-2f;
          // pixel centres are the integers: hit pixels lie in [ceil(px - ex), floor(px + ex)]
          const float fx0 = floorf((px - ex) * (1.f / SUBX)), fx1 = floorf((px + ex) * (1.f / SUBX)) + 1.f;
          const float fy0 = floorf((py - ey) * (1.f / SUBY)), fy1 = floorf((py + ey) * (1.f / SUBY)) + 1.f;
          if (fx0 > (float)sx0) sx0 = (int)fminf(fx0, (float)sx1);
          if (fy0 > (float)sy0) sy0 = (int)fminf(fy0, (float)sy1);
          if (fx1 < (float)sx1) sx1 = (int)fmaxf(fx1, (float)sx0);
          if (fy1 < (float)sy1) sy1 = (int)fmaxf(fy1, (float)sy0);
        }  // else (NaN / inf opacity): keep the whole rect, like the reference's blend of a NaN alpha
        const int sw = sx1 - sx0, sh = sy1 - sy0;
        unsigned long long m = 0ull;
        uint32_t kind = BK_RECT;
        SpanParams sp = {};
        if (!any || sw <= 0 || sh <= 0) {
          my_tiles = 0;
        } else if (sw * sh <= MASK_MAX_SUBTILES) {
          kind = BK_MASK;
          // (a rect of at most 64 tiles spans at most ceil(64/BLOCK_BIG)+1 blocks in a row: block bits fit 64 bits)
          const int bx0 = sx0 / BLOCK_BIG, by0 = sy0 / BLOCK_BIG, bw = (sx1 - 1) / BLOCK_BIG - bx0 + 1;
          unsigned long long blocks = 0ull;  // blocks (BLOCK_BIG x BLOCK_BIG tiles) with at least one listed tile
          if (tau_m >= 0.f && tau_m < 3.0e38f && ca * cc - cb * cb > 0.f) {
            // one closed-form column span per tile ROW (common.h row_span: the ellipse cut by the row band is convex) instead
            // of one block test per TILE: the wave runs max-over-lanes iterations, and a footprint has far fewer rows (<= 8)
            // than tiles (<= 64)
            const SpanParams spm = span_params(px, py, ca, cb, cc, tau_m);
            for (int sy = sy0; sy < sy1; sy++) {
              int c0, c1;
              row_span(spm, sy, sx0, sx1, c0, c1);
              if (c1 > c0) {
                const unsigned long long run = (c1 - c0 >= 64 ? ~0ull : ((1ull << (c1 - c0)) - 1ull)) << (c0 - sx0);
                m |= run << ((sy - sy0) * sw);
                const int b0 = c0 / BLOCK_BIG - bx0, b1 = (c1 - 1) / BLOCK_BIG - bx0;  // block columns [b0, b1]
                blocks |= (((1ull << (b1 - b0 + 1)) - 1ull) << b0) << ((sy / BLOCK_BIG - by0) * bw);
              }
            }
          } else {  // degenerate conic / threshold: the per-tile test (keeps NaNs: they fail every comparison)
            const float b_c = cb / cc, b_a = cb / ca;
            for (int sy = sy0; sy < sy1; sy++)
              for (int sx = sx0; sx < sx1; sx++) {
                const float bx = (float)(sx * SUBX), by = (float)(sy * SUBY);
                if (block_hit(px, py, ca, cb, cc, b_c, b_a, tau_m, bx, by, bx + (SUBX - 1), by + (SUBY - 1))) {
                  m |= 1ull << ((sy - sy0) * sw + (sx - sx0));
                  blocks |= 1ull << ((sy / BLOCK_BIG - by0) * bw + (sx / BLOCK_BIG - bx0));
                }
              }
          }
          my_tiles = (uint32_t)__popcll(m);
          my_entries = (uint32_t)__popcll(blocks);
        } else if (tau_m >= 0.f && tau_m < 3.0e38f && ca * cc - cb * cb > 0.f) {
          // larger footprints: per-row column spans in closed form (common.h row_span); expand re-evaluates them
          kind = BK_SPANS;
          sp = span_params(px, py, ca, cb, cc, tau_m);
          my_tiles = 1;  // counted by the macro walk below
        } else {
          my_tiles = (uint32_t)(sw * sh);
        }
        // list entries at block size BLOCK_BIG = blocks with at least one listed internal tile (at block size 1 the
        // entries are the listed tiles themselves); for BK_SPANS the walk also yields the number of listed tiles: counted
        // behind this nest (the wave may count tall footprints together)
        if (my_tiles && kind != BK_MASK) {
          need_count = true;
          c_kind = kind; c_sp = sp; c_sx0 = sx0; c_sy0 = sy0; c_sx1 = sx1; c_sy1 = sy1;
          if (kind == BK_SPANS) {  // (also for the rare footprint whose spans turn out to list no tile: never read then)
            bext[2 * idx] = make_float4(sp.gx, sp.gy, sp.ex, sp.ey);
            bext[2 * idx + 1] = make_float4(sp.boa, sp.boc, sp.ta, sp.da);
          }
        }
        bkind = kind;
        if (my_tiles) {
          // render record: one whole 64-byte line per Gaussian
          const float L2E = 1.4426950408889634f;
          float f[NCH];
          if (RAW) {
            // SH2RGB of the DC band (utils/sh_utils.py:125-126), altitude = ECEF_to_UVA(xyz)[2], constant 1
#pragma unroll
            for (int k = 0; k < 3; k++) f[k] = s_c[3 * t + k] * SH_C0 + 0.5f;
            f[3] = alt[0] * p[0] + alt[1] * p[1] + alt[2] * p[2] + alt[3];
            f[4] = 1.f;
          } else {
#pragma unroll
            for (int k = 0; k < NCH; k++) f[k] = colors[NCH * idx + k];
          }
          // three of the line's four 16-byte quarters: the fourth is never read, and not writing it is 14 % of this kernel
          packed[4 * idx + 0] = make_float4(px, py, ca * (-0.5f * L2E), cb * L2E);
          packed[4 * idx + 1] = make_float4(cc * (-0.5f * L2E), op, f[0], f[1]);
          packed[4 * idx + 2] = make_float4(f[2], f[3], f[4], 1.f / d);
          // internal-tile rect [sx0,sx1) x [sy0,sy1) (16 bits each) + hit mask relative to it (0 = every tile)
          bi0 = make_uint4((uint32_t)sx0 | ((uint32_t)sx1 << 16), (uint32_t)sy0 | ((uint32_t)sy1 << 16), (uint32_t)m,
                           (uint32_t)(m >> 32));
          key_bits = __float_as_uint(d);  // the reference's depth sort key (rasterizer_impl.cu:103-106)
        }
      }
    }
    radii[idx] = radius;
  }
  // ---- entries and listed tiles of the row-span / whole-rect footprints (= what expand's walk_macro_row emits): one closed-form
  //      count per block row (common.h count_macro_row). In its own lane a footprint of R block rows keeps the wave's other lanes
  //      waiting for R counts; counted by the whole wave, one block row per lane, it costs about two (the broadcast of its
  //      parameters, one count, two wave sums). The wave compares the two: the tallest lane against two per tall footprint. With
  //      the log-normal sizes of a trained scene most waves hold one or two footprints of 5-30 block rows
  //      (profiles/r06_surface_front_end.txt); where every footprint is tall they stay in their lanes. ----
  {
    const int lane = t & 63;
    const int my0 = c_sy0 / BLOCK_BIG, my1 = (c_sy1 - 1) / BLOCK_BIG;
    const uint32_t mrows = need_count ? (uint32_t)(my1 - my0 + 1) : 0u;
    const bool tall = mrows >= 4u;
    const uint32_t walk_cost = wave_max_u32_dpp(mrows);
    const uint32_t coop_cost = wave_sum_u32_dpp(tall ? 2u : 0u) + 3u;  // (+ the short ones, still in their lanes)
    const bool together = coop_cost < walk_cost;
    uint32_t ent = 0, fine = 0;
    unsigned long long big = __ballot(tall && together);
    while (big) {
      const int src = __builtin_ctzll(big);
      big &= big - 1ull;
      SpanParams gs;
      gs.gx = __shfl(c_sp.gx, src, 64); gs.gy = __shfl(c_sp.gy, src, 64); gs.ex = __shfl(c_sp.ex, src, 64);
      gs.ey = __shfl(c_sp.ey, src, 64); gs.boa = __shfl(c_sp.boa, src, 64); gs.boc = __shfl(c_sp.boc, src, 64);
      gs.ta = __shfl(c_sp.ta, src, 64); gs.da = __shfl(c_sp.da, src, 64);
      const uint32_t gk = __shfl(c_kind, src, 64);
      const int gx0 = __shfl(c_sx0, src, 64), gy0 = __shfl(c_sy0, src, 64), gx1 = __shfl(c_sx1, src, 64), gy1 = __shfl(c_sy1, src, 64);
      uint32_t ne = 0, nf = 0;
      for (int MY = gy0 / BLOCK_BIG + lane; MY <= (gy1 - 1) / BLOCK_BIG; MY += 64)
        count_macro_row<BLOCK_BIG>(gk, gs, gx0, gy0, gx1, gy1, MY, ne, nf);
      ne = wave_sum_u32_dpp(ne);
      nf = wave_sum_u32_dpp(nf);
      if (lane == src) { ent = ne; fine = nf; }
    }
    if (need_count) {
      if (!(tall && together))
        for (int MY = my0; MY <= my1; MY++) count_macro_row<BLOCK_BIG>(c_kind, c_sp, c_sx0, c_sy0, c_sx1, c_sy1, MY, ent, fine);
      my_tiles = fine;
      my_entries = ent;
    }
  }
  // exclusive prefix of the tile counts inside the workgroup (record slots in Gaussian-id order) and the
  // workgroup total
  const uint32_t inc = wave_incl_scan_u32(my_tiles);
  const int lane = t & 63;
  if (lane == 63) s_cnt[t >> 6] = inc;
  __syncthreads();
  const uint32_t w0 = s_cnt[0], w1 = s_cnt[1], w2 = s_cnt[2], w3 = s_cnt[3];
  const int w = t >> 6;
  const uint32_t pre = (w > 0 ? w0 : 0u) + (w > 1 ? w1 : 0u) + (w > 2 ? w2 : 0u);
  if (t < rows) {
    binfo[idx] = bi0;
    binfo[(size_t)P + idx] = make_uint4(my_tiles, pre + inc - my_tiles, key_bits, bkind | (my_entries << 2));
  }
  // range of the depth keys of listed Gaussians (lets the host drop sort passes whose digit is constant) and the
  // workgroup's pair count: plain stores, reduced by pblock_scan_kernel (same-address atomics from 16k waves cost
  // 0.36 ms here)
  __shared__ uint32_t s_k[5][BLK / 64];
  uint32_t kmax = my_tiles ? key_bits : 0u, knmin = my_tiles ? ~key_bits : 0u, esum = my_tiles ? my_entries : 0u;
  // a lane's share saturates at 2^23 - 1 so that the workgroup's sum (256 lanes) stays below 2^31 whatever the image size:
  // bit 31 of the stored word is the error flag (the sum only feeds the list-granularity heuristic)
  const unsigned long long o64 = (unsigned long long)my_tiles * op64;
  uint32_t osum = o64 > 0x7FFFFFull ? 0x7FFFFFu : (uint32_t)o64;
  kmax = wave_max_u32_dpp(kmax);
  knmin = wave_max_u32_dpp(knmin);
  esum = wave_sum_u32_dpp(esum);
  osum = wave_sum_u32_dpp(osum);
  my_err = wave_or_u32_dpp(my_err);
  if (lane == 0) { s_k[0][w] = kmax; s_k[1][w] = knmin; s_k[2][w] = esum; s_k[3][w] = osum; s_k[4][w] = my_err; }
  __syncthreads();
  if (t == 0) {
    pblock[blockIdx.x] = w0 + w1 + w2 + w3;
    uint32_t a = s_k[0][0], b = s_k[1][0], e = s_k[2][0], ow = s_k[3][0], er = s_k[4][0];
    for (int i = 1; i < BLK / 64; i++) {
      a = s_k[0][i] > a ? s_k[0][i] : a;
      b = s_k[1][i] > b ? s_k[1][i] : b;
      e += s_k[2][i];
      ow += s_k[3][i];
      er |= s_k[4][i];
    }
    pbkey[4 * blockIdx.x] = a;
    pbkey[4 * blockIdx.x + 1] = b;
    pbkey[4 * blockIdx.x + 2] = e;
    pbkey[4 * blockIdx.x + 3] = ow | er;  // bit 31: the workgroup's error flag (EOGS_ERR_ALTITUDE), summed out by the scan
  }
}

void launch_preprocess_fwd(const FwdPrepArgs& a, const GeomWS& g, hipStream_t s) {
  const int gx = (a.W + TILE - 1) / TILE, gy = (a.H + TILE - 1) / TILE;
  const uint32_t nblk = ceil_div_u32((uint64_t)a.P, BLK);
  auto* kern = a.raw ? preprocess_fwd_kernel<true> : preprocess_fwd_kernel<false>;
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(BLK), 0, s, a.P, a.H, a.W, gx, gy, a.means3D, a.scales, a.rotations,
                     a.cov3D_precomp, a.opacities, a.colors, a.viewmatrix, a.alt_affine, a.scale_modifier,
                     (int)a.antialiasing, a.radii, g.packed, g.binfo, g.bext, g.pblock, g.pbkey);
}


// ------------------------------------------------------------------------------------------------------
// Backward, per Gaussian.
// ------------------------------------------------------------------------------------------------------
// WIDE: how many of a Gaussian's records are in flight at once, chosen per launch (launch_gaussian_bwd; every variant adds the same
// records in the same order: the same bits). 0: four (70 VGPRs, seven waves per SIMD). 1: eight per trip of the long-list loop
// while any lane of the wave has more than four to go (122 VGPRs, four waves). 2: as 1, and a Gaussian with up to EIGHT listed
// tiles reads its flags and records in one trip. Measured (profiles/r05_ab_gaussian_bwd_wide.txt), ms at 0 / 1 / 2:
// 1 M at 1024^2 opacity 0.01 (4.05 tiles per Gaussian) 0.0894 / 0.0854 / 0.0922; opacity 0.1 (8.7) 0.1757 / 0.1534 / 0.1456;
// 2048^2 (9.5) 0.1823 / 0.1544 / 0.1455; 2 M at opacity 0.1 (6.6) 0.2134 / 0.1912 / 0.1887; trained opacities (10.8 listed, most
// of them dead: flags first, few records) 0.0787 / 0.0827 / 0.0826 — there the seven waves hide the flags -> records chain better
// than more loads per lane do.
// ALT: the records of an altitude-only render (32 bytes, common.h REC_ALT): only colour 3 has a gradient.
// (the wide builds sit exactly at the 128 VGPRs that four waves per SIMD leave: said to the compiler, which otherwise takes a
// 129th for the cooperative section's sake and loses the wave)
template <bool RAW, bool ALT, int WIDE>
__global__ __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(WIDE ? (ALT ? 5 : 4) : 6, WIDE ? (ALT ? 5 : 4) : 8))) void gaussian_bwd_kernel(
    int P, int H, int W,
    const float* __restrict__ means3D, const float* __restrict__ scales, const float* __restrict__ rotations,
    const float* __restrict__ cov3D_precomp, const float* __restrict__ opacities, const float* __restrict__ vm,
    const float* __restrict__ proj, const float* __restrict__ alt, const int* __restrict__ radii, float scale_modifier, int antialiasing,
    const uint4* __restrict__ binfo, const uint32_t* __restrict__ pblock, const float* __restrict__ records,
    const uint8_t* __restrict__ live,
    float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacity,
    float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dscales,
    float* __restrict__ dL_drotations, bool want_T, bool want_vm, float* __restrict__ vmpart, uint32_t blk0,
    float* __restrict__ dL_dcolors_lead, int lead_cols, const uint32_t* __restrict__ misc, uint32_t cap_slots,
    uint32_t cap_entries, int noflag_ok) {
  __shared__ float s_m[3 * BLK];
  __shared__ float s_s[3 * BLK];
  __shared__ float s_red[BLK / 64][18];
  __shared__ float s_coop[11][BLK];  // record sums of the Gaussians the waves summed cooperatively (GB_COOP)
  // A forward queued on a capacity token (EOGS_FLAG_DEFER_COUNTS) that needed more than its workspaces hold has built no
  // lists (binning.hip block_lists_kernel) and wrote no record: its slots would lie beyond `records` / `live`. The same
  // comparison here: such a backward reads none of them and returns zero gradients (the host repeats the forward).
  constexpr uint32_t GB_DIRECT = WIDE == 2 ? 8u : 4u;  // listed tiles up to which flags and records are read in one trip
  constexpr bool GB_WIDE = WIDE != 0;
  const bool fits = misc[MISC_TOTAL_HI] == 0u && misc[MISC_MACRO_HI] == 0u && misc[MISC_TOTAL_LO] <= cap_slots &&
                    misc[MISC_MACRO_LO] <= cap_entries;
  // Where tiles saturate (trained opacities: a tile's pixels stop after a tenth of its list) most listed pairs are DEAD — behind
  // every pixel's last contributor, never written — and the one-trip read below fetches records only to discard them: there a
  // Gaussian reads its flags first unless it lists a single tile (gaussian_bwd 0.087 -> 0.080 ms at 1 M trained; with live
  // pairs the one-trip read wins: opacity 0.1 0.193 against 0.194 two-trip, and against 0.169 with eight records in one
  // trip, which costs the headline 7 % in registers — profiles/r04_experiments/ab_gb_direct.txt). "Saturating" as the tile
  // schedule defines it (binning.hip): mean list depth x mean pair opacity beyond SCHED_K = 60 entries.
  const float opw = (float)misc[MISC_OPW_LO] + 4294967296.0f * (float)misc[MISC_OPW_HI];
  const float ntiles8 = (float)((W + SUBX - 1) / SUBX) * (float)((H + SUBY - 1) / SUBY);
  const uint32_t dlim = opw > 64.0f * 60.0f * ntiles8 ? 1u : GB_DIRECT;
  // flag-free records (common.h noflag_scene): every listed pair's record was written, the flags were not
  const bool noflag = (noflag_ok & 1) != 0 && ((noflag_ok & 2) != 0 || noflag_scene(misc[MISC_OPW_LO], misc[MISC_OPW_HI], W, H));
  const int t = threadIdx.x;
  const uint32_t blk = blk0 + blockIdx.x;  // workgroup index over ALL Gaussians (the launch may cover a range of them)
  const size_t row0 = (size_t)blk * BLK;
  const int rows = (int)(((size_t)P - row0) < (size_t)BLK ? ((size_t)P - row0) : (size_t)BLK);
  const size_t idx = row0 + t;
  float vmsum[18];
#pragma unroll
  for (int k = 0; k < 18; k++) vmsum[k] = 0.f;
  // (Summing the records in double was measured: 0.109 -> 0.130 ms, and no accuracy gained — the residual error of
  // extreme footprints comes from the per-pixel pass)
  float acc[12];
#pragma unroll
  for (int k = 0; k < 12; k++) acc[k] = 0.f;
  bool visible = false;
  float4 rot_in = make_float4(1.f, 0.f, 0.f, 0.f);
  float op_raw = 0.f;
  // The rows of means3D / scales that the math below reads through LDS are REQUESTED here and parked in LDS only after the record
  // sum: the workgroup barrier that publishes them used to stand at the kernel's head, in front of every other load — one more
  // dependent round trip (rows -> barrier -> radii / binning record -> records) of a kernel that is nothing but round trips.
  float st_m[3], st_s[3] = {0.f, 0.f, 0.f};
  stage_rows3_load(means3D, row0, rows, st_m);
  if (scales) stage_rows3_load(scales, row0, rows, st_s);

  int radius_in = 0;
  uint4 bi1 = make_uint4(0u, 0u, 0u, 0u);
  uint32_t pb = 0;
  if (t < rows) {
    // Round 4: everything that does not depend on another load is requested up front (the kernel waited 45 % of its wave
    // time, profiles/r03_v30: radii -> binfo -> flags -> records -> rotation / opacity was a chain of five dependent round
    // trips per workgroup): radii, the binning record, the workgroup's first slot and the per-Gaussian inputs of the math
    // below are independent loads in flight together; a Gaussian with at most GB_DIRECT listed tiles (the usual case: four)
    // then reads its flags AND its records in ONE trip — a dead pair's record is read and discarded: never-written memory,
    // but only ever selected away — and only longer lists take the two-trip form (flags, then the live records alone).
    radius_in = radii[idx];
    bi1 = binfo[(size_t)P + idx];  // (its own plane: this kernel reads 16 of a Gaussian's 32 bytes, and fetched all 32 while they shared a line)
    pb = pblock[blk];
    if (!cov3D_precomp) rot_in = reinterpret_cast<const float4*>(rotations)[idx];
    if (RAW || antialiasing) op_raw = opacities[idx];
  }
  visible = t < rows && radius_in > 0;
  const uint32_t n_all = (visible && fits) ? bi1.x : 0u;
  // Which Gaussians the wave sums together (next section): those beyond GB_COOP listed tiles — a rule of the Gaussian ALONE. A
  // per-wave estimate (the wave's longest lane against all its candidates beyond GB_COOP / 2 together) measured 8 % / 11 % faster on
  // 1 M / 2 M surface-shaped Gaussians (profiles/r06_ab_gb_coop_threshold.txt) and was withdrawn: the DPP tree adds in another
  // order than the lane, so a Gaussian's bits then depend on which neighbours share its wave — a run that compacts the pruned
  // Gaussians away and one that keeps them retired in place stopped agreeing bit for bit (tests/test_gpu_example.py), and so did
  // the narrow and the wide builds of this kernel while the estimate was in units of the build's own trip width.
  const bool is_big = n_all > GB_COOP;

  // ---- Gaussians that list more than GB_COOP tiles: summed by the whole wave, one after the other ----
  // A lane sums its own Gaussian's records serially, four or eight per memory round trip. That is the right shape while
  // footprints are a few tiles (every lane busy, the records of a wave one contiguous region) and the wrong one for the large
  // splats a trained scene keeps (ground planes, roofs: hundreds to thousands of listed tiles): ONE lane then walks thousands of
  // dependent trips while 63 idle, and the launch lasts as long as its largest Gaussian — 1.1 ms of a 1.6 ms step at 300 k
  // surface-shaped Gaussians / 800^2, 0.6 of 1.2 ms at 1 M (profiles/r06_regime_scan_before_coop.txt; the synthetic scenes of
  // rounds 1-5 had no such tail). Here lane l takes records l, l + 64, ... of the Gaussian (consecutive lanes on consecutive
  // 48-byte records: coalesced) and the 64 partial sums are added by a fixed DPP tree: deterministic, another order of the
  // same additions than the serial sum. The totals wait in LDS (s_coop) until the lane's own math below: this section sits in
  // front of the serial sums, where few registers are live — behind them it cost the wide builds their fourth wave per SIMD.
  {
    const int lane = t & 63;
    unsigned long long big = __builtin_amdgcn_ballot_w64(is_big);
    while (big) {
      const int src = (int)__builtin_ctzll(big);
      big &= big - 1ull;
      const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)n_all, src);
      const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)(pb + bi1.y), src);  // (slots are below 2^31: api.hip)
      constexpr int RQc = (ALT ? REC_ALT : REC) / 4;
      const float4* r4c = reinterpret_cast<const float4*>(records);
      float part[11];
#pragma unroll
      for (int k = 0; k < 11; k++) part[k] = 0.f;
#pragma unroll 2
      for (uint32_t q0 = 0; q0 < n; q0 += 64u) {
        const uint32_t q = q0 + (uint32_t)lane;
        const bool lv = q < n && (noflag || live[s0 + q] != 0);
        const uint32_t qq = q < n ? q : 0u;  // a valid address either way
        const float4 ra = r4c[rec_q((size_t)s0 + qq, 0, cap_slots, RQc)];
        const float4 rb = r4c[rec_q((size_t)s0 + qq, 1, cap_slots, RQc)];
        float3 rc = make_float3(0.f, 0.f, 0.f);
        if (!ALT) rc = reinterpret_cast<const float3*>(r4c + rec_q((size_t)s0 + qq, 2, cap_slots, RQc))[0];
        if (lv) {  // (a dead pair's record is never-written memory: read, selected away)
          part[0] += ra.x; part[1] += ra.y; part[2] += ra.z; part[5] += ra.w;
          part[3] += rb.x; part[4] += rb.y;
          if (ALT) {
            part[9] += rb.z;
          } else {
            part[6] += rb.z; part[7] += rb.w;
            part[8] += rc.x; part[9] += rc.y; part[10] += rc.z;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 11; k++) {  // (ALT: the sums it does not take are zeros, handed over like the others)
        const float tot = wave_sum_dpp(part[k]);
        if (lane == 0) s_coop[k][(t & ~63) + src] = tot;
      }
    }
  }

  {
    if (visible) {
      // fixed-order sum of this Gaussian's (tile,Gaussian) records: deterministic, no atomics
      const uint32_t n = is_big ? 0u : n_all;
      const size_t s0 = (size_t)pb + bi1.y;  // Gaussian-id order: a wave reads one contiguous region
      constexpr int RQ = (ALT ? REC_ALT : REC) / 4;  // quarters per record (common.h rec_q: quarter-major planes of cap_slots)
      const float4* r4 = reinterpret_cast<const float4*>(records);
      uint32_t q_first = 0;
      if (n <= dlim) {
        float4 ra[GB_DIRECT], rb[GB_DIRECT];
        float3 rc[GB_DIRECT];
        bool lv[GB_DIRECT];
        // the (up to four; WIDE = 2: eight) flags of this Gaussian's consecutive slots in ONE unaligned 4- or 8-byte load (bytes past its n-th flag
        // are other Gaussians' or the 256 bytes of slack behind the array, common.h bin_layout: read, never used): the four
        // byte loads cost this kernel 10 % (profiles/r05_ab_noflag.txt: it runs 0.085 instead of 0.0955 ms without them)
        static_assert(GB_DIRECT == 4u || GB_DIRECT == 8u, "one or two dwords of flags");
        unsigned long long fl4 = 0x0101010101010101ull;
        if (!noflag && n) __builtin_memcpy(&fl4, live + s0, GB_DIRECT);
#pragma unroll
        for (uint32_t u = 0; u < GB_DIRECT; u++) {
          const uint32_t qq = u < n ? u : 0u;  // (n == 0: slot s0 itself may lie past the arrays — never dereferenced)
          lv[u] = false;
          if (u < n) {
            lv[u] = ((uint32_t)(fl4 >> (8u * u)) & 0xFFu) != 0u;
            ra[u] = r4[rec_q(s0 + qq, 0, cap_slots, RQ)];
            rb[u] = r4[rec_q(s0 + qq, 1, cap_slots, RQ)];
            if (!ALT) rc[u] = reinterpret_cast<const float3*>(r4 + rec_q(s0 + qq, 2, cap_slots, RQ))[0];
          }
        }
#pragma unroll
        for (uint32_t u = 0; u < GB_DIRECT; u++) {
          if (lv[u]) {
            acc[0] += ra[u].x; acc[1] += ra[u].y; acc[2] += ra[u].z; acc[5] += ra[u].w;
            acc[3] += rb[u].x; acc[4] += rb[u].y;
            if (ALT) {
              acc[9] += rb[u].z;
            } else {
              acc[6] += rb[u].z; acc[7] += rb[u].w;
              acc[8] += rc[u].x; acc[9] += rc[u].y; acc[10] += rc[u].z;
            }
          }
        }
        q_first = n;  // done
      }
      // Two memory round trips instead of 2n dependent ones: first all live flags of this Gaussian (independent byte
      // loads -> a register bitmask), then the live records (independent loads driven by the mask). Dead pairs
      // (behind every pixel's last contributor) were never written and are never read.
      for (uint32_t q0 = q_first; q0 < n; q0 += 32) {
        const uint32_t m_n = n - q0 < 32u ? n - q0 : 32u;
        uint32_t m = 0;
        const uint32_t all = m_n >= 32u ? 0xFFFFFFFFu : (1u << m_n) - 1u;
        if (noflag) {
          m = all;
        } else {  // four flags (bytes that are 0 or 1) per unaligned 4-byte load; what lies past the m_n-th is masked away
#pragma unroll 2
          for (uint32_t q = 0; q < m_n; q += 4) {
            uint32_t f;
            __builtin_memcpy(&f, live + s0 + q0 + q, 4);
            m |= ((f & 1u) | ((f >> 7) & 2u) | ((f >> 14) & 4u) | ((f >> 21) & 8u)) << q;
          }
          m &= all;
        }
        // up to four live records per trip: twelve independent loads in flight, summed in list order; EIGHT per trip while any
        // lane of the wave still has more than four to go (ballot: the lanes inside this loop)
        auto trip = [&](auto widthc) {
          constexpr int NW = decltype(widthc)::value;
          uint32_t q[NW];
          bool have[NW];
#pragma unroll
          for (int u = 0; u < NW; u++) {
            have[u] = m != 0u;
            q[u] = have[u] ? q0 + (uint32_t)__builtin_ctz(m) : q0;
            m &= m - 1u;  // (0 & anything stays 0)
          }
          float4 ra[NW], rb[NW];
          float3 rc[NW];
#pragma unroll
          for (int u = 0; u < NW; u++) {
            if (NW > 4 && u >= 4) {  // the wide trip's second half: only the lanes that have that many
              if (have[u]) {
                ra[u] = r4[rec_q(s0 + q[u], 0, cap_slots, RQ)];
                rb[u] = r4[rec_q(s0 + q[u], 1, cap_slots, RQ)];
                if (!ALT) rc[u] = reinterpret_cast<const float3*>(r4 + rec_q(s0 + q[u], 2, cap_slots, RQ))[0];
              }
              continue;
            }
            const uint32_t qq = have[u] ? q[u] : q[0];  // a valid address either way
            ra[u] = r4[rec_q(s0 + qq, 0, cap_slots, RQ)];
            rb[u] = r4[rec_q(s0 + qq, 1, cap_slots, RQ)];
            if (!ALT) rc[u] = reinterpret_cast<const float3*>(r4 + rec_q(s0 + qq, 2, cap_slots, RQ))[0];
          }
#pragma unroll
          for (int u = 0; u < NW; u++) {
            if (have[u]) {  // record (common.h REC) -> acc: 0,1 mean2D  2,3,4 conic  5 opacity  6..10 colour
              acc[0] += ra[u].x; acc[1] += ra[u].y; acc[2] += ra[u].z; acc[5] += ra[u].w;
              acc[3] += rb[u].x; acc[4] += rb[u].y;
              if (ALT) {
                acc[9] += rb[u].z;
              } else {
                acc[6] += rb[u].z; acc[7] += rb[u].w;
                acc[8] += rc[u].x; acc[9] += rc[u].y; acc[10] += rc[u].z;
              }
            }
          }
        };
        while (m) {
          if (GB_WIDE && __builtin_amdgcn_ballot_w64(__popc(m) > 4) != 0ull) trip(std::integral_constant<int, 8>{});
          else trip(std::integral_constant<int, 4>{});
        }
      }
    }
  }
  stage_rows3_store(rows, st_m, s_m);
  if (scales) stage_rows3_store(rows, st_s, s_s);
  __syncthreads();
  if (is_big) {  // the wave-summed records of a large Gaussian (above)
#pragma unroll
    for (int k = 0; k < 11; k++) acc[k] = s_coop[k][t];
  }
  if (t < rows) {
    // record layout: 0,1 = dL/dmean2D (NDC units)  2,3,4 = dL/dconic (a,b,c)  5 = dL/dopacity  6..10 = dL/dcolor
    const float gxn = acc[0], gyn = acc[1];
    float dop = acc[5];
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dmean3[3] = {0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f};
    float dq[4] = {0.f, 0.f, 0.f, 0.f};
    const float m[3] = {s_m[3 * t], s_m[3 * t + 1], s_m[3 * t + 2]};

    if (visible) {
      float c6[6];
      float q[4] = {1.f, 0.f, 0.f, 0.f};
      float s3[3] = {0.f, 0.f, 0.f};
      float q_inv = 1.f, op_in = 0.f;
      if (RAW || antialiasing) op_in = RAW ? sigmoidf(op_raw) : op_raw;
      if (cov3D_precomp) {
        const float2* c2 = reinterpret_cast<const float2*>(cov3D_precomp + 6 * idx);
        float2 a = c2[0], b = c2[1], c = c2[2];
        c6[0] = a.x; c6[1] = a.y; c6[2] = b.x; c6[3] = b.y; c6[4] = c.x; c6[5] = c.y;
      } else {
        const float4 qq = rot_in;
        q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
        s3[0] = s_s[3 * t]; s3[1] = s_s[3 * t + 1]; s3[2] = s_s[3 * t + 2];
        if (RAW) q_inv = raw_activate(s3, q);
        cov3d_from_scale_rot(s3, scale_modifier, q, c6);  // recomputed instead of stored: saves 48 B/Gaussian of HBM traffic
      }
      float T[2][3];
      build_T(vm, W, H, T);
      float c_xx, c_xy, c_yy;
      cov2d(T, c6, c_xx, c_xy, c_yy);
      const float dLc[3] = {acc[2], acc[3], acc[4]};

      // ---- computeCov2DCUDA (backward.cu:194-272) ----
      const float h_var = 0.3f;
      float d_inside_root = 0.f;
      if (antialiasing) {
        const float det_cov = c_xx * c_yy - c_xy * c_xy;
        c_xx += h_var;
        c_yy += h_var;
        const float det_p = c_xx * c_yy - c_xy * c_xy;
        const float hcs = sqrtf(fmaxf(0.000025f, det_cov / det_p));
        const float d_hcs = dop * op_in;
        dop = dop * hcs;
        d_inside_root = (det_cov / det_p) <= 0.000025f ? 0.f : d_hcs / (2 * hcs);
      } else {
        c_xx += h_var;
        c_yy += h_var;
      }
      float dxx = 0.f, dxy = 0.f, dyy = 0.f;
      if (antialiasing) {
        const float x = c_xx, y = c_yy, z = c_xy, w = h_var;
        const float sqv = w * w + w * (x + y) + x * y - z * z;
        const float denom_f = d_inside_root / (sqv * sqv);
        dxx = w * (w * y + y * y + z * z) * denom_f;
        dyy = w * (w * x + x * x + z * z) * denom_f;
        dxy = -2.f * w * z * (w + x + y) * denom_f;
      }
      const float denom = c_xx * c_yy - c_xy * c_xy;
      const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
      if (denom2inv != 0) {
        dxx += denom2inv * (-c_yy * c_yy * dLc[0] + 2 * c_xy * c_yy * dLc[1] + (denom - c_xx * c_yy) * dLc[2]);
        dyy += denom2inv * (-c_xx * c_xx * dLc[2] + 2 * c_xx * c_xy * dLc[1] + (denom - c_xx * c_yy) * dLc[0]);
        dxy += denom2inv * 2 * (c_xy * c_yy * dLc[0] - (denom + 2 * c_xy * c_xy) * dLc[1] + c_xx * c_xy * dLc[2]);
        dcov[0] = (T[0][0] * T[0][0] * dxx + T[0][0] * T[1][0] * dxy + T[1][0] * T[1][0] * dyy);
        dcov[3] = (T[0][1] * T[0][1] * dxx + T[0][1] * T[1][1] * dxy + T[1][1] * T[1][1] * dyy);
        dcov[5] = (T[0][2] * T[0][2] * dxx + T[0][2] * T[1][2] * dxy + T[1][2] * T[1][2] * dyy);
        dcov[1] = 2 * T[0][0] * T[0][1] * dxx + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dxy + 2 * T[1][0] * T[1][1] * dyy;
        dcov[2] = 2 * T[0][0] * T[0][2] * dxx + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dxy + 2 * T[1][0] * T[1][2] * dyy;
        dcov[4] = 2 * T[0][2] * T[0][1] * dxx + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dxy + 2 * T[1][1] * T[1][2] * dyy;
      }
      // ---- dL/dT (backward.cu:276-287), reduced over Gaussians instead of stored as [P,6] ----
      if (want_T) {
        const float V[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
        float TV[2][3];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
          for (int k = 0; k < 3; k++) TV[r][k] = T[r][0] * V[k][0] + T[r][1] * V[k][1] + T[r][2] * V[k][2];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          vmsum[k] = 2 * TV[0][k] * dxx + TV[1][k] * dxy;
          vmsum[3 + k] = 2 * TV[1][k] * dyy + TV[0][k] * dxy;
        }
      }
      // ---- dL/dmean3D = A[0:2,:]^T g   (backward.cu:439-445; the cov2D kernel contributes zero, :313-317) ----
      dmean3[0] = proj[0] * gxn + proj[1] * gyn;
      dmean3[1] = proj[4] * gxn + proj[5] * gyn;
      dmean3[2] = proj[8] * gxn + proj[9] * gyn;

      // ---- computeCov3D backward (backward.cu:331-394) ----
      if (!cov3D_precomp) {
        const Mat3 R = quat_to_Rm(q[0], q[1], q[2], q[3]);
        const float s[3] = {scale_modifier * s3[0], scale_modifier * s3[1], scale_modifier * s3[2]};
        float M[3][3];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int c = 0; c < 3; c++) M[a][c] = s[a] * R.m[a][c];
        const float dS[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                                {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                                {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
        float dM[3][3];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int c = 0; c < 3; c++)
            dM[a][c] = (2.0f * M[a][0]) * dS[0][c] + (2.0f * M[a][1]) * dS[1][c] + (2.0f * M[a][2]) * dS[2][c];
#pragma unroll
        for (int k = 0; k < 3; k++) dscale[k] = R.m[k][0] * dM[k][0] + R.m[k][1] * dM[k][1] + R.m[k][2] * dM[k][2];
        float D[3][3];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int c = 0; c < 3; c++) D[a][c] = dM[a][c] * s[a];
        const float r = q[0], x = q[1], y = q[2], z = q[3];
        dq[0] = 2 * z * (D[0][1] - D[1][0]) + 2 * y * (D[2][0] - D[0][2]) + 2 * x * (D[1][2] - D[2][1]);
        dq[1] = 2 * y * (D[1][0] + D[0][1]) + 2 * z * (D[2][0] + D[0][2]) + 2 * r * (D[1][2] - D[2][1]) - 4 * x * (D[2][2] + D[1][1]);
        dq[2] = 2 * x * (D[1][0] + D[0][1]) + 2 * r * (D[2][0] - D[0][2]) + 2 * z * (D[1][2] + D[2][1]) - 4 * y * (D[2][2] + D[0][0]);
        dq[3] = 2 * r * (D[0][1] - D[1][0]) + 2 * x * (D[2][0] + D[0][2]) + 2 * y * (D[1][2] + D[2][1]) - 4 * z * (D[1][1] + D[0][0]);
      }
      if (RAW) {
        // chain through the activations: exp, normalize, sigmoid, and the altitude feature's dependence on xyz
#pragma unroll
        for (int k = 0; k < 3; k++) dscale[k] *= s3[k];
        const float dot = q[0] * dq[0] + q[1] * dq[1] + q[2] * dq[2] + q[3] * dq[3];
#pragma unroll
        for (int k = 0; k < 4; k++) dq[k] = (dq[k] - q[k] * dot) * q_inv;
        dop *= op_in * (1.f - op_in);
#pragma unroll
        for (int k = 0; k < 3; k++) dmean3[k] += alt[k] * acc[9];
      }
    }
    // every output row is written (zeros for culled Gaussians): no memset pass over 144 B/Gaussian
    dL_dmeans2D[3 * idx] = gxn; dL_dmeans2D[3 * idx + 1] = gyn; dL_dmeans2D[3 * idx + 2] = 0.f;
    if (RAW) {
#pragma unroll
      for (int ch = 0; ch < 3; ch++) dL_dcolors[3 * idx + ch] = SH_C0 * acc[6 + ch];
    } else {
#pragma unroll
      for (int ch = 0; ch < NCH; ch++) dL_dcolors[NCH * idx + ch] = acc[6 + ch];
    }
    if (dL_dcolors_lead) {  // (uniform) the leading columns once more, contiguous: a data-parallel caller's exchange buffer
#pragma unroll
      for (int ch = 0; ch < NCH; ch++)
        if (ch < lead_cols) dL_dcolors_lead[(size_t)lead_cols * idx + ch] = (RAW ? SH_C0 : 1.f) * acc[6 + ch];
    }
    dL_dopacity[idx] = dop;
#pragma unroll
    for (int k = 0; k < 3; k++) dL_dmeans3D[3 * idx + k] = dmean3[k];
    if (dL_dcov3D) {
#pragma unroll
      for (int k = 0; k < 6; k++) dL_dcov3D[6 * idx + k] = dcov[k];
    }
    if (dL_dscales) {
#pragma unroll
      for (int k = 0; k < 3; k++) dL_dscales[3 * idx + k] = dscale[k];
    }
    if (dL_drotations) reinterpret_cast<float4*>(dL_drotations)[idx] = make_float4(dq[0], dq[1], dq[2], dq[3]);

    if (want_vm) {  // means3D^T @ dL_dmeans2D and sum dL_dmeans2D (__init__.py:193-201); z column is zero
#pragma unroll
      for (int a = 0; a < 3; a++) {
        vmsum[6 + 3 * a + 0] = m[a] * gxn;
        vmsum[6 + 3 * a + 1] = m[a] * gyn;
      }
      vmsum[15] = gxn;
      vmsum[16] = gyn;
    }
  }

  // Camera sums (wrapper math of __init__.py:179-201): one row of 18 partials per workgroup, summed in fixed order by
  // camera_sum_kernel — no atomics, so grad_viewmatrix is bitwise reproducible like every other gradient.
  if (want_T || want_vm) {  // wave-uniform: kernel arguments
#pragma unroll
    for (int k = 0; k < 18; k++) {
      float v = wave_sum(vmsum[k]);
      if ((t & 63) == 0) s_red[t >> 6][k] = v;
    }
    __syncthreads();
    if (t < 18) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < BLK / 64; w++) v += s_red[w][t];
      vmpart[(size_t)blk * 18 + t] = v;
    }
  }
}

// Sums the per-workgroup camera partials [nblk][18] in a fixed order: thread t owns column t % 18 of the rows
// r = t / 18 (mod 14), then 14 row-partials per column are added serially.
__global__ __launch_bounds__(BLK) void camera_sum_kernel(const float* __restrict__ vmpart, uint32_t nblk,
                                                         float* __restrict__ dL_dT_sum, float* __restrict__ dL_dvm_mean) {
  __shared__ float s_p[14][18];
  const int t = threadIdx.x;
  if (t < 14 * 18) {
    const int c = t % 18, r0 = t / 18;
    float v = 0.f;
    for (uint32_t r = r0; r < nblk; r += 14) v += vmpart[(size_t)r * 18 + c];
    s_p[r0][c] = v;
  }
  __syncthreads();
  if (t < 18) {
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 14; r++) v += s_p[r][t];
    if (t < 6) {
      if (dL_dT_sum) dL_dT_sum[t] = v;
    } else if (dL_dvm_mean) {
      // columns 6..14 -> means3D^T @ dL_dmeans2D (3x3, third column stays 0), 15..17 -> sum dL_dmeans2D
      const int k = t - 6;
      if (k < 9) dL_dvm_mean[k] = (k % 3) < 2 ? v : 0.f;
      else dL_dvm_mean[k] = (k - 9) < 2 ? v : 0.f;
    }
  }
}

// Four waves per SIMD with eight records in flight per lane beat seven waves with four while most listed pairs are live; where
// tiles saturate early, few records follow the flags and the waves are what hides the flags -> records chain. The crossover
// (1 M / 2 M Gaussians at 1024^2, gaussian_bwd ms narrow / wide by list depth x mean pair opacity): 80: 0.212 / 0.188,
// 120: 0.118 / 0.101, 195: 0.093 / 0.086, 350: 0.078 / 0.080, trained 1 M: 0.075 / 0.078, trained 2 M: 0.105 / 0.120
// (profiles/r05_ab_gaussian_bwd_wide.txt). A host that does not know the depth keeps the narrow kernel.
int gaussian_bwd_wide(int64_t R, int P) {
  static const int forced = [] {
    const char* e = getenv("EOGS_GB_WIDE");
    return e ? atoi(e) : -1;
  }();
  if (forced >= 0) return forced > 2 ? 2 : forced;
  if (!nr_shallow(R) || P <= 0) return 0;
  return (double)nr_slots(R) >= 6.0 * (double)P ? 2 : 1;
}

void launch_gaussian_bwd(const GaussBwdArgs& a, const GeomWS& g, const BinWS& b, int p_begin, int p_end, hipStream_t s) {
  const uint32_t nblk_all = ceil_div_u32((uint64_t)a.P, BLK);
  const uint32_t blk0 = (uint32_t)p_begin / BLK, nblk = ceil_div_u32((uint64_t)(p_end - p_begin), BLK);
  const bool want_T = a.dL_dT_sum != nullptr, want_vm = a.dL_dvm_mean != nullptr;
  using Kern = decltype(&gaussian_bwd_kernel<false, false, 0>);
  static Kern const table[3][2][2] = {
      {{gaussian_bwd_kernel<false, false, 0>, gaussian_bwd_kernel<false, true, 0>}, {gaussian_bwd_kernel<true, false, 0>, gaussian_bwd_kernel<true, true, 0>}},
      {{gaussian_bwd_kernel<false, false, 1>, gaussian_bwd_kernel<false, true, 1>}, {gaussian_bwd_kernel<true, false, 1>, gaussian_bwd_kernel<true, true, 1>}},
      {{gaussian_bwd_kernel<false, false, 2>, gaussian_bwd_kernel<false, true, 2>}, {gaussian_bwd_kernel<true, false, 2>, gaussian_bwd_kernel<true, true, 2>}}};
  Kern kern = table[a.wide < 0 ? 0 : (a.wide > 2 ? 2 : a.wide)][a.raw ? 1 : 0][a.alt_only ? 1 : 0];
  if (nblk)
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(BLK), 0, s, a.P, a.H, a.W, a.means3D, a.scales, a.rotations,
                       a.cov3D_precomp, a.opacities, a.viewmatrix, a.projmatrix, a.alt_affine, a.radii, a.scale_modifier,
                       (int)a.antialiasing, g.binfo, g.pblock, b.records, b.live, a.dL_dmeans2D, a.dL_dcolors, a.dL_dopacity,
                       a.dL_dmeans3D, a.dL_dcov3D, a.dL_dscales, a.dL_drotations, want_T, want_vm, g.vmpart, blk0,
                       a.dL_dcolors_lead, a.lead_cols, g.misc, b.cap_slots, b.cap_entries, a.noflag_ok);
  if ((want_T || want_vm) && p_end == a.P)
    hipLaunchKernelGGL(camera_sum_kernel, dim3(1), dim3(BLK), 0, s, g.vmpart, nblk_all, a.dL_dT_sum, a.dL_dvm_mean);
}
