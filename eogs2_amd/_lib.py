"""Locates and loads the HIP rasterizer library (libeogs_rast_hip.so, built in-tree).

There is no CPU fallback: if the library is missing or cannot be loaded the
product path raises. `get()` is the single accessor the host code uses.
"""
import os
import threading

from ._abi import RastABI

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libeogs_rast_hip.so")

_lock = threading.Lock()
_abi = None


def get() -> RastABI:
    global _abi
    if _abi is None:
        with _lock:
            if _abi is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"HIP rasterizer library not built: {LIB_PATH} is missing. "
                        "Run `python -m eogs2_amd.build` (needs hipcc). There is no CPU fallback."
                    )
                _abi = RastABI(LIB_PATH)
                if _abi.backend != "hip-gfx950":
                    raise RuntimeError(f"{LIB_PATH}: unexpected backend {_abi.backend!r}")
    return _abi
