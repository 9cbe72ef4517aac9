"""eogs2_amd — MI355X-native differentiable Gaussian-splatting rasterizer for EOGS++.

Drop-in for the hot path of gardiens/EOGS2: the `diff_gaussian_rasterization`
Python API (see `rasterizer.py`) over a C-ABI HIP library (`csrc/`, `include/eogs_rast.h`).
"""
from .rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
    RastError,
    NUM_CHANNELS,
)

__version__ = "0.1.0"
