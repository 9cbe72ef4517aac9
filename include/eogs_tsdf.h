/*
 * eogs_tsdf.h — C-ABI of the TSDF integration step of the DSM post-processing (second piece):
 *   TSDFVolume.integrate      src/gaussiansplatting/tsdf.py:459-498  (+ update_tsdf :500-520)
 *   RangeImageEOGS.sample_sdf src/gaussiansplatting/tsdf.py:325-368  (+ _world_to_view / _view_to_world :233-241)
 * One kernel per range image where the reference materialises ~25 voxel-sized temporaries (12 B/voxel coordinates,
 * grid_sample output, masks, three index gathers and two index scatters).
 *
 * Same conventions as eogs_rast.h: plain DEVICE pointers + sizes, `void* stream` is a hipStream_t, int status
 * (0 ok, <0 error, message via eogs_rast_last_error()), the library never allocates device memory.
 */
#ifndef EOGS_TSDF_H_INCLUDED
#define EOGS_TSDF_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Integrates one altitude image into the volume, in place.
 *   nx, ny, nz        voxels per dimension; volumes are f32[nx][ny][nz] (tsdf.py:451-456)
 *   ax, ay, az        f32[nx], f32[ny], f32[nz]: world coordinates of the voxel centres along each axis (the
 *                     reference's torch.linspace values, tsdf.py:402-407, so coordinates are bit-identical)
 *   affine            f32[24] = coef[3][3], intercept[3], inv(coef)[3][3], inv(coef) @ intercept [3]
 *                     (view = coef p + intercept, :234; world = inv(coef) view - inv(coef) intercept, :238-240)
 *   model_scale       points are divided by it before projection, distances multiplied by it (:341,366)
 *   trunc_margin      truncation distance (:385)
 *   H, W              image size; altitude f32[H][W], weight f32[H][W] (= get_weights(): clamp(angle, 0, 1), :322-323)
 * Per voxel: p = (x,y,z)/scale; (u,v,a) = view(p); bilinear sample (align_corners, zero padding) of altitude and weight
 * at (u,v); valid = |u| <= 1 & |v| <= 1; sdf = |world(u,v,alt_s) - p| sign(a - alt_s) scale; where valid & sdf >= -trunc:
 *   w_new = w_old + weight_s;  tsdf_new = (w_old tsdf_old + weight_s min(1, sdf/trunc)) / w_new      (:510-518)
 * (0/0 gives NaN exactly as in the reference when both weights are zero). */
int eogs_tsdf_integrate(int nx, int ny, int nz, const float* ax, const float* ay, const float* az, const float* affine,
                        float model_scale, float trunc_margin, int H, int W, const float* altitude, const float* weight,
                        float* tsdf_vol, float* weight_vol, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EOGS_TSDF_H_INCLUDED */
