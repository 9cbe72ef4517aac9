/*
 * eogs_resample.h — C-ABI of the fused virtual-camera resample (second piece).
 *
 * Replaces steps 2-3 of render_resample_virtual_camera
 * (src/gaussiansplatting/gaussian_renderer/renderer_cc_shadow.py:32-50):
 *     virtual_uv = einsum("...ij,...j->...i", cam2virt, rendered_uva)[..., :2]
 *     sample     = grid_sample(virtual_render[None], virtual_uv[None], align_corners=True)[0]   (bilinear, zeros padding)
 *     sample[3][(virtual_uv.abs() > 1).any(-1)] = -100
 * which PyTorch runs as stack / einsum / grid_sample / mask / index_put forward and their autograd backward
 * (1.2 ms per sun-camera resample at 1024^2 on the MI355X, twice per training iteration: train_pan.py:305-316,375-391).
 * Here: one kernel forward, one backward.
 *
 * Same conventions as eogs_rast.h (DEVICE pointers, `void* stream` = hipStream_t, int status, no allocation).
 *   virtual_render f32[C,Hv,Wv] planar (the virtual camera's render: rgb, altitude, accumulated opacity)
 *   uva            f32[H,W,3]  per output pixel (u, v, altitude) of the true camera   (train_pan.py:281)
 *   cam2virt       f32[9]      row-major 3x3 on the device (affine_cameras.py:360, 428-429)
 *   sample         f32[n_out,H,W]: the first n_out <= C channels of virtual_render resampled (the reference keeps 4)
 *   uv             f32[H,W,2]  the sampling coordinates (returned by the reference, used by its losses)
 *   fill_channel   channel of `sample` overwritten by fill_value where |u| > 1 or |v| > 1 (3 and -100 in the
 *                  reference); -1 disables
 */
#ifndef EOGS_RESAMPLE_H_INCLUDED
#define EOGS_RESAMPLE_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int eogs_resample_forward(int C, int Hv, int Wv, int H, int W, int n_out, const float* virtual_render,
                          const float* uva, const float* cam2virt, int fill_channel, float fill_value,
                          float* sample, float* uv, void* stream);

/* Backward workspace (bounding boxes of the cells touched by each 16x16 output tile). */
int eogs_resample_bytes(int H, int W, size_t* bytes);

/* Backward.
 *   dL_dsample f32[n_out,H,W]; dL_duv f32[H,W,2] or NULL
 *   dL_dvirtual f32[C,Hv,Wv]: fully overwritten. The four-tap scatter is evaluated as a gather per virtual tile (no global
 *     atomics) whose sums run in a fixed order — candidate tiles in tile order, the pixels of a cell in pixel order — so the
 *     result is bitwise reproducible run to run (PyTorch's grid_sampler_2d_backward, which the reference uses, is not).
 *     Exceptions: n_out > 4 or ws == NULL fall back to global fp32 atomic adds, and the round-2 form of the tile kernel
 *     (environment EOGS_RESAMPLE_BWD=1, a tuning aid) sums with LDS float atomics; neither fixes the last bit.
 *   dL_duva f32[H,W,3]: cam2virt[0:2,:]^T (dL_duv + the sampler's gradient with respect to the coordinates), fully overwritten
 * cam2virt itself receives no gradient (the reference derives it from fixed camera matrices). */
int eogs_resample_backward(int C, int Hv, int Wv, int H, int W, int n_out, const float* virtual_render,
                           const float* uva, const float* cam2virt, int fill_channel,
                           const float* dL_dsample, const float* dL_duv,
                           float* dL_dvirtual, float* dL_duva, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EOGS_RESAMPLE_H_INCLUDED */
