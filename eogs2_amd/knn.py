"""`distCUDA2` over the C-ABI of include/eogs_knn.h.

Mirrors the reference's simple-knn binding (src/gaussiansplatting/submodules/simple-knn/spatial.cu:15-26): takes a
float [P,3] tensor on the GPU, returns float32 [P] = mean squared distance to the three nearest neighbours. The top-level
package `simple_knn` re-exports it as `simple_knn._C.distCUDA2`, the name scene/gaussian_model.py:20 imports.
No CPU fallback.
"""
import ctypes

import torch

from . import _lib
from .rasterizer import _Ctx, _ptr


def distCUDA2(points):
    abi = _lib.get()
    if points.ndim != 2 or points.shape[1] != 3:
        raise RuntimeError("distCUDA2: points must have dimensions (num_points, 3)")
    dev, P = points.device, points.shape[0]
    pts = points.detach().to(torch.float32).contiguous()
    means = torch.full((P,), 0.0, dtype=torch.float32, device=dev)  # spatial.cu:21
    with _Ctx(abi, dev) as cx:
        n = ctypes.c_size_t()
        abi.check(abi.knn_bytes(P, ctypes.byref(n)))
        ws = torch.empty((n.value,), dtype=torch.uint8, device=dev)
        abi.check(abi.knn_mean_dist2(P, _ptr(pts), _ptr(means), _ptr(ws), ws.numel(), cx.stream))
    return means


__all__ = ["distCUDA2"]
