"""Drop-in module name used by the reference's only caller
(src/gaussiansplatting/gaussian_renderer/renderer.py:3-6):

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

Put this repository's root on PYTHONPATH ahead of the CUDA extension and
`train_pan.py` runs unchanged on PyTorch-ROCm.
"""
from eogs2_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
)
