/*
 * eogs_optim.h — C-ABI of the optimizer / density-control kernels on the Gaussian parameter tensors
 * (SURVEY.md §8 row f3). Same conventions as eogs_rast.h: DEVICE pointers + sizes, `void* stream` is a hipStream_t,
 * int status (0 ok, <0 error, message via eogs_rast_last_error()), no device allocation inside the library.
 *
 *  - eogs_adam_step: one launch for all parameter groups of the reference's optimizer
 *        torch.optim.Adam(l, lr=0.0, eps=1e-15) with six single-tensor groups xyz / f_dc / f_rest / opacity /
 *        scaling / rotation, each with its own lr   (src/gaussiansplatting/scene/gaussian_model.py:228-262,
 *        stepped once per iteration at train_pan.py:663-670).
 *        Arithmetic = torch.optim.Adam (amsgrad off, no weight decay, maximize off), fp32:
 *            m <- b1 m + (1-b1) g;  v <- b2 v + (1-b2) g^2
 *            p <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 *  - eogs_sum_into: the gradient accumulation of an iteration's renders, every parameter tensor in one launch.
 *  - eogs_pack_columns: gathers column ranges of several row-major [P, k] gradient tensors side by side into ONE
 *        row-major [P, K] bucket (or scatters it back) in one launch: the pack step of the view-sharded data-parallel
 *        all-reduce (xyz 3 + f_dc 3 + opacity 1 + scaling 3 + rotation 4 = 14 floats = 56 B/Gaussian).
 *  - eogs_compact_*: stable stream compaction of the rows of many tensors by one keep-mask, replacing the
 *        `tensor[mask]` chain of prune_points / _prune_optimizer (gaussian_model.py:466-505: 6 parameters, their 12 Adam
 *        moments and 3 statistics = 21 boolean-mask gathers, each with its own nonzero + host sync).
 */
#ifndef EOGS_OPTIM_H_INCLUDED
#define EOGS_OPTIM_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EOGS_ADAM_MAX_TENSORS 16

/* One parameter tensor of a step: all four arrays hold `numel` fp32 values (exp_avg / exp_avg_sq are torch's state names). */
typedef struct {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t numel;
  float lr;
} eogs_adam_tensor;

/* `tensors` is a HOST array of n <= EOGS_ADAM_MAX_TENSORS descriptors (copied into the kernel arguments).
 * `step` is the 1-based step count t of THIS update (torch increments state['step'] before using it).
 * beta1 / beta2 / eps are doubles because torch derives 1 - beta and the bias corrections from Python floats and only
 * then rounds to fp32 (1 - 0.999f differs from 0.001f by 1.3e-5 relative).
 * Tensors with numel == 0 are skipped. Asynchronous on `stream`. */
int eogs_adam_step(int n, const eogs_adam_tensor* tensors, double beta1, double beta2, double eps, int64_t step,
                   void* stream);

/* Gradient accumulation over the renders of one iteration, all parameter tensors in ONE launch:
 *     dst[i] = ((dst[i] + src[0][i]) + src[1][i]) + ...      (fp32, this order: what autograd's `p.grad += g` does render by render,
 *     train_pan.py:278-469: the reference accumulates the view's, the sun camera's and the random camera's backward into p.grad).
 * `tensors` is a HOST array of n <= EOGS_SUM_MAX_TENSORS descriptors with nsrc <= EOGS_SUM_MAX_SOURCES sources each (all non-NULL,
 * none overlapping its dst). Tensors with numel == 0 are skipped. Asynchronous on `stream`. */
#define EOGS_SUM_MAX_TENSORS 8
#define EOGS_SUM_MAX_SOURCES 4
typedef struct {
  float* dst;
  const float* src[EOGS_SUM_MAX_SOURCES];
  int64_t numel;
} eogs_sum_tensor;
int eogs_sum_into(int n, const eogs_sum_tensor* tensors, int nsrc, void* stream);

/* One tensor of a pack: `data` holds rows x width fp32 values; columns [col0, col0 + ncols) take part. */
#define EOGS_PACK_MAX_TENSORS 8
typedef struct {
  float* data;
  int width;
  int col0;
  int ncols;
} eogs_pack_tensor;

/* unpack == 0: packed[r][o_t + c] = tensors[t].data[r * width_t + col0_t + c]; unpack != 0: the reverse copy.
 * o_t = sum of ncols of the tensors before t; packed_cols must equal the sum of all ncols (<= 16).
 * `tensors` is a HOST array of n <= EOGS_PACK_MAX_TENSORS descriptors. Asynchronous on `stream`. */
int eogs_pack_columns(int64_t rows, int n, const eogs_pack_tensor* tensors, float* packed, int packed_cols, int unpack,
                      void* stream);

/* Row compaction. Workspace: per-workgroup keep counts / offsets. */
int eogs_compact_bytes(int64_t n_rows, size_t* bytes);
/* Scans keep[n_rows] (device, 1 byte per row, non-zero = keep) and returns the number of kept rows in *n_keep
 * (host; synchronises `stream` once, like the reference's boolean indexing does per tensor). */
int eogs_compact_plan(int64_t n_rows, const uint8_t* keep, void* ws, size_t ws_bytes, int64_t* n_keep, void* stream);
/* Copies the kept rows of n_tensors row-major tensors, in order, to dst[i] (n_keep rows each): HOST arrays of DEVICE
 * pointers and of row sizes in bytes (multiples of 4, <= 256). src[i] and dst[i] must not overlap. Asynchronous. */
int eogs_compact_apply(int64_t n_rows, const uint8_t* keep, int n_tensors, const void* const* src, void* const* dst,
                       const int* row_bytes, const void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EOGS_OPTIM_H_INCLUDED */
