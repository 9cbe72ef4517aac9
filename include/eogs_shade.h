/*
 * eogs_shade.h — C-ABI of the per-pixel image chain between the rasterizer's raw render and the scalar losses
 * (last piece). Three small groups, each one forward and one backward kernel where the reference
 * runs 6-15 elementwise PyTorch kernels and autograd replays them:
 *
 *   eogs_shade_*    AffineCamera.render_pipeline     src/gaussiansplatting/scene/cameras/affine_cameras.py:303-348
 *                   with ShadowMap.forward           scene/cameras/affine_cameras.py:33-40
 *   eogs_mloss_*    Suncamera_L.forward              src/gaussiansplatting/loss/shadow.py:37-51
 *                   RandomcamRendering_Loss          src/gaussiansplatting/loss/main_loss.py:83-96,151-164
 *   eogs_tshadow_*  Translucentshadows_L.forward     src/gaussiansplatting/loss/shadow.py:7-17
 *
 * Same conventions as eogs_rast.h: plain DEVICE pointers + sizes, `void* stream` is a hipStream_t, int status
 * (0 ok, <0 error, message via eogs_rast_last_error()), the library never allocates device memory. Images are
 * contiguous fp32 planes [C][H][W]; `uv` is [H][W][2]. Every sum is reduced per workgroup and then in a fixed order
 * (no atomics): results are bitwise reproducible.
 */
#ifndef EOGS_SHADE_H_INCLUDED
#define EOGS_SHADE_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Workspace for the per-workgroup partial sums of any call below on an H x W image (tshadow: n <= H*W). */
int eogs_shade_bytes(int H, int W, size_t* bytes);

/* ---- camera render pipeline -------------------------------------------------------------------------------------
 *   cc[c]     = M[c][0] raw[0] + M[c][1] raw[1] + M[c][2] raw[2] + M[c][3]
 *               (use_cc: Conv2d(3,3,1) weight | bias, affine_cameras.py:311-312; use_exposure: exposure[0], :313-323;
 *                neither: identity)
 *   shadow    = exp(0.4 * min(alt_diff, 0))                       ShadowMap, only when alt_diff != NULL (:329-333)
 *   shaded[c] = shadow * cc[c] + (1 - shadow) * inshadow[c] * cc[c]      (= cc[c] when alt_diff == NULL, :334-336)
 * raw f32[3][H][W], alt_diff f32[H][W] or NULL (= altitude_render - sun_altitude_sample, train_pan.py:318),
 * M f32[3][4] row-major, inshadow f32[3] (ignored when alt_diff == NULL).
 * Outputs: cc, shaded f32[3][H][W]; shadow f32[H][W] (must be NULL exactly when alt_diff is NULL). cc may be NULL. */
int eogs_shade_forward(int H, int W, const float* raw, const float* alt_diff, const float* M, const float* inshadow,
                       float* cc, float* shaded, float* shadow, void* stream);

/* Backward. Upstream gradients g_shaded (required), g_cc and g_shadow (NULL = zero), same shapes as the outputs.
 *   g_raw      f32[3][H][W], fully overwritten
 *   g_alt_diff f32[H][W], fully overwritten (NULL when alt_diff is NULL); clip(max=0) passes the gradient for alt_diff <= 0
 *   g_params   f32[15] = dL/dM[3][4] then dL/dinshadow[3] (zero when alt_diff is NULL) */
int eogs_shade_backward(int H, int W, const float* raw, const float* alt_diff, const float* M, const float* inshadow,
                        const float* g_shaded, const float* g_cc, const float* g_shadow, float* g_raw,
                        float* g_alt_diff, float* g_params, void* ws, size_t ws_bytes, void* stream);

/* ---- masked resample losses ---------------------------------------------------------------------------------------
 *   rgb_diff = rgb_a - rgb_b                                  (raw_render - sun_rgb_sample, shadow.py:38)
 *   mask     = cond(alt_diff) & |u| < 1 & |v| < 1             detached
 *              EOGS_MLOSS_SUN:    alt_diff > -1e-2            (shadow.py:39)
 *              EOGS_MLOSS_RANDOM: |alt_diff| < 0.30           (main_loss.py:153-155)
 *   L_alt    = sum(|alt_diff| mask) / sum(mask),  L_rgb = sum(|rgb_diff| mask) / sum(mask)   (all 3 channels over the
 *              pixel count, shadow.py:42-48)
 * out f32[3] = {L_alt, L_rgb, sum(mask)}; an empty mask gives {0, 0, 0} (main_loss.py:161-163). */
#define EOGS_MLOSS_SUN 0
#define EOGS_MLOSS_RANDOM 1
int eogs_mloss_forward(int H, int W, int mode, const float* alt_diff, const float* rgb_a, const float* rgb_b,
                       const float* uv, float* out, void* ws, size_t ws_bytes, void* stream);

/* Backward: `out` is forward's result (the count is read on the device, no host sync), upstream f32[2] = dL/d{L_alt,
 * L_rgb}. g_alt_diff f32[H][W], g_rgb_a f32[3][H][W] fully overwritten; g_rgb_b (= -g_rgb_a) may be NULL.
 * torch.abs'(0) = 0 is kept (sign). */
int eogs_mloss_backward(int H, int W, int mode, const float* alt_diff, const float* rgb_a, const float* rgb_b,
                        const float* uv, const float* out, const float* upstream, float* g_alt_diff, float* g_rgb_a,
                        float* g_rgb_b, void* stream);

/* ---- translucent-shadow regulariser ------------------------------------------------------------------------------
 *   out[0] = -mean(a log2 b + (1 - a) log2(1 - b)),  b = clip(a, 0.05, 0.95)       over n elements
 * Backward: g_a fully overwritten; the clip passes its gradient for 0.05 <= a <= 0.95. upstream f32[1]. */
int eogs_tshadow_forward(int64_t n, const float* a, float* out, void* ws, size_t ws_bytes, void* stream);
int eogs_tshadow_backward(int64_t n, const float* a, const float* upstream, float* g_a, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EOGS_SHADE_H_INCLUDED */
