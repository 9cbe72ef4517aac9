"""TSDF volume integration over the C-ABI of include/eogs_tsdf.h (second piece).

`TSDFVolume(vol_bounds, vox_size, trunc_margin_fact)` has the reference's constructor arithmetic and attributes
(src/gaussiansplatting/tsdf.py:374-456: `num_voxels_per_dimension`, `axes`, `_tsdf_vol` = ones, `_weight_vol` = zeros)
and `integrate(rangeimage)` (:459-498) runs ONE HIP kernel per range image instead of the reference's ~25 voxel-sized
PyTorch temporaries. `rangeimage` is duck-typed like `RangeImageEOGS` (:186-368): `affine_model = (coef[3,3],
intercept[3])`, `model_scale`, `altitude_img [1,1,H,W]`, `get_weights() [1,1,H,W]` (the normal estimation that produces
the weights stays PyTorch: it runs once per image on H x W pixels). No CPU / eager fallback.
"""
import numpy as np
import torch

from . import _lib
from .rasterizer import _Ctx, _ptr


def volume_axes(vol_bounds, vox_size, device):
    """tsdf.py:387-407: voxel counts and per-axis centre coordinates, the reference's own statements."""
    vb = torch.tensor(np.asarray(vol_bounds), dtype=torch.float32, device=device)
    assert vb.shape == (3, 2), "vol_bounds should be of shape (3,2)"
    n = (vb[:, 1] - vb[:, 0]) // vox_size + 1
    n = n.ceil().long()
    starts = vb[:, 0]
    ends = vb[:, 0] + n * vox_size
    dims = tuple(n.cpu().numpy().tolist())
    axes = [torch.linspace(starts[i], ends[i], dims[i], device=device) for i in range(3)]
    return dims, axes


class TSDFVolume:
    def __init__(self, vol_bounds, vox_size, trunc_margin_fact, device="cuda:0"):
        self.device = torch.device(device)
        self.vox_size = vox_size
        self._trunc_margin = trunc_margin_fact * self.vox_size
        self.num_voxels_per_dimension, self.axes = volume_axes(vol_bounds, vox_size, self.device)
        self._tsdf_vol = torch.ones(size=self.num_voxels_per_dimension, device=self.device, dtype=torch.float32)
        self._weight_vol = torch.zeros(size=self.num_voxels_per_dimension, device=self.device, dtype=torch.float32)

    def integrate(self, rangeimage):
        integrate(self._tsdf_vol, self._weight_vol, self.axes, rangeimage.affine_model[0], rangeimage.affine_model[1],
                  float(rangeimage.model_scale), float(self._trunc_margin), rangeimage.altitude_img,
                  rangeimage.get_weights())


def integrate(tsdf_vol, weight_vol, axes, coef, intercept, model_scale, trunc_margin, altitude_img, weight_img):
    """In place on `tsdf_vol` / `weight_vol` (f32[nx,ny,nz], contiguous)."""
    abi = _lib.get()
    dev = tsdf_vol.device
    for t in (tsdf_vol, weight_vol):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.ndim != 3:
            raise RuntimeError("tsdf integrate: volumes must be contiguous float32 [nx, ny, nz]")
    if tsdf_vol.shape != weight_vol.shape:
        raise RuntimeError("tsdf integrate: volume shapes differ")
    nx, ny, nz = tsdf_vol.shape
    f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
    ax, ay, az = (f(a) for a in axes)
    if (ax.numel(), ay.numel(), az.numel()) != (nx, ny, nz):
        raise RuntimeError("tsdf integrate: axes do not match the volume")
    A, b = f(coef).reshape(3, 3), f(intercept).reshape(3)
    Ainv = torch.linalg.inv(A)  # tsdf.py:238-239
    affine = torch.cat([A.reshape(-1), b, Ainv.reshape(-1), Ainv @ b])
    alt, wgt = f(altitude_img), f(weight_img)
    H, W = alt.shape[-2:]
    if alt.numel() != H * W or wgt.numel() != H * W:
        raise RuntimeError("tsdf integrate: altitude and weight images must be single-channel H x W")
    with _Ctx(abi, dev) as cx:
        abi.check(abi.tsdf_integrate(nx, ny, nz, _ptr(ax), _ptr(ay), _ptr(az), _ptr(affine), model_scale, trunc_margin,
                                     H, W, _ptr(alt), _ptr(wgt), _ptr(tsdf_vol), _ptr(weight_vol), cx.stream))


__all__ = ["TSDFVolume", "integrate", "volume_axes"]
