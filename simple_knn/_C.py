"""`from simple_knn._C import distCUDA2` (src/gaussiansplatting/scene/gaussian_model.py:20) on the MI355X library."""
from eogs2_amd.knn import distCUDA2  # noqa: F401

__all__ = ["distCUDA2"]
