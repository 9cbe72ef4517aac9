/*
 * eogs_loss.h — C-ABI of the fused image-space photometric loss (first piece):
 *   L = w_l1 * mean|x - y| + w_ssim * mean(SSIM(x, y)) + bias
 * which covers, with the right weights, the three reference functions
 *   l1_loss(x, y)                      src/gaussiansplatting/utils/loss_utils.py:18-19   (1, 0, 0)
 *   ssim(x, y, 11, size_average)       src/gaussiansplatting/utils/loss_utils.py:45-85   (0, 1, 0)
 *   lphotom(x, y, Ll1, lambda)         src/gaussiansplatting/utils/image_utils.py:27-28  (1-lambda, -lambda, lambda)
 * The reference evaluates SSIM as five depthwise 11x11 conv2d (zero padding 5, window = outer product of a
 * normalised sigma=1.5 Gaussian) plus ~15 elementwise kernels, and autograd replays them; here one forward and one
 * backward kernel do the same arithmetic in fp32.
 *
 * Same conventions as eogs_rast.h: plain DEVICE pointers + sizes, `void* stream` is a hipStream_t, int status
 * (0 ok, <0 error, message via eogs_rast_last_error()), the library never allocates device memory.
 * Images are `planes` contiguous H x W fp32 planes (a [C,H,W] or [N,C,H,W] tensor: planes = N*C; the window is the same
 * for every channel, loss_utils.py:35-42).
 */
#ifndef EOGS_LOSS_H_INCLUDED
#define EOGS_LOSS_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EOGS_LOSS_WINDOW 11 /* loss_utils.py:45 window_size */
#define EOGS_LOSS_L1 1u     /* mode bit: accumulate mean|x-y|            */
#define EOGS_LOSS_SSIM 2u   /* mode bit: accumulate mean SSIM, keep maps */

/* Workspace: per-workgroup partial sums (+ with EOGS_LOSS_SSIM three fp32 maps dSSIM/d{mu1, E[x^2], E[xy]} per pixel,
 * written by forward and re-read by backward). */
int eogs_loss_bytes(int planes, int H, int W, unsigned mode, size_t* bytes);

/* Forward. Writes
 *   out        f32[3]          {w_l1*l1_mean + w_ssim*ssim_mean + bias, l1_mean, ssim_mean}  (means over all planes)
 *   plane_sums f32[planes][2]  {sum|x-y|, sum SSIM} per plane (for size_average=False); may be NULL
 * Sums are reduced in a fixed order: results are bitwise reproducible. Asynchronous on `stream`. */
int eogs_loss_forward(int planes, int H, int W, const float* img, const float* gt, unsigned mode,
                      float w_l1, float w_ssim, float bias, float* out, float* plane_sums,
                      void* ws, size_t ws_bytes, void* stream);

/* Backward with respect to `img` (the rendered image; the reference never differentiates the ground truth).
 *   upstream   device f32[3]: dLoss/d out[0..2] of forward (same w_l1, w_ssim); NULL means {1, 0, 0}
 *   plane_grad device f32[planes][2] or NULL: when given, replaces (upstream, w_l1, w_ssim) by per-plane weights
 *              dLoss/d plane_sums (the size_average=False case, loss_utils.py:84-85)
 *   dL_dimg    f32[planes][H][W], fully overwritten
 * torch.abs'(0) = 0 is kept (sign). */
int eogs_loss_backward(int planes, int H, int W, const float* img, const float* gt, unsigned mode,
                       float w_l1, float w_ssim, const float* upstream, const float* plane_grad,
                       const void* ws, size_t ws_bytes, float* dL_dimg, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EOGS_LOSS_H_INCLUDED */
