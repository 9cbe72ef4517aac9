/*
 * eogs_knn.h — C-ABI of the initialisation-time nearest-neighbour statistic.
 *
 * Replaces `distCUDA2(points)` of the reference's simple-knn extension
 * (src/gaussiansplatting/submodules/simple-knn/spatial.cu:15-26 -> SimpleKNN::knn, simple_knn.cu:187-222), called once
 * at scene/gaussian_model.py:179-182 to initialise the scales: for every point the mean of the squared distances to its
 * three nearest neighbours (self excluded by index; exact — the box pruning of simple_knn.cu:147-185 is conservative).
 * Same conventions as eogs_rast.h (DEVICE pointers, `void* stream` = hipStream_t, int status, no allocation).
 */
#ifndef EOGS_KNN_H_INCLUDED
#define EOGS_KNN_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int eogs_knn_bytes(int P, size_t* bytes);
/* points f32[P,3]; mean_dist2 f32[P] (FLT_MAX-based like the reference when P < 4: with fewer than three other points the
 * missing neighbours count as 1e37). Asynchronous on `stream`. */
int eogs_knn_mean_dist2(int P, const float* points, float* mean_dist2, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EOGS_KNN_H_INCLUDED */
