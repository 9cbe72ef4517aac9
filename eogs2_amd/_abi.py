"""ctypes prototypes for the C-ABI declared in include/eogs_rast.h.

One `RastABI` instance wraps one loaded shared library. The product path only
ever wraps the HIP library (see `_lib.py`); tests may wrap the CPU oracle, which
exports the same symbols over host pointers.
"""
import ctypes as C

ABI_VERSION = 8

EOGS_OK = 0
ERR_NAMES = {
    -1: "EOGS_ERR_INVALID_ARG",
    -2: "EOGS_ERR_DEVICE",
    -3: "EOGS_ERR_WORKSPACE",
    -4: "EOGS_ERR_ALTITUDE",
    -5: "EOGS_ERR_OVERFLOW",
    -6: "EOGS_ERR_NO_COLORS",
}
FLAG_ANTIALIASING = 1
FLAG_DEBUG = 2
FLAG_RAW_PARAMS = 4
FLAG_DEFER_COUNTS = 8
FLAG_NO_READBACK = 16
FLAG_ALT_ONLY = 32  # include/eogs_rast.h EOGS_FLAG_ALT_ONLY: an altitude-only render
MIRROR_BYTES = 64
LOSS_L1 = 1
LOSS_SSIM = 2
MLOSS_SUN = 0
MLOSS_RANDOM = 1

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_u = C.c_uint
_z = C.c_size_t
_i64 = C.c_int64

# name -> (restype, argtypes); the order is exactly include/eogs_rast.h
SIGNATURES = {
    "eogs_rast_last_error": (C.c_char_p, []),
    "eogs_rast_abi_version": (_i, []),
    "eogs_rast_backend": (C.c_char_p, []),
    "eogs_rast_geom_bytes": (_i, [_i, C.POINTER(_z)]),
    "eogs_rast_image_bytes": (_i, [_i, _i, C.POINTER(_z)]),
    "eogs_rast_binning_bytes": (_i, [_i, _i, _i, _i64, C.POINTER(_z)]),
    "eogs_rast_scratch_bytes": (_i, [_i, _i, _i, C.POINTER(_z)]),
    "eogs_rast_forward_prepare": (
        _i,
        [_i, _i, _i, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _u, _p, _p, _z, _p, _z, C.POINTER(_i64), _p],
    ),
    "eogs_rast_forward_counts": (_i, [C.POINTER(_i64)]),
    "eogs_rast_read_counts": (_i, [_i, _i, _i, _p, _z, _i, _p, C.POINTER(_i64)]),
    "eogs_rast_mirror_arm": (_i, [_p]),
    "eogs_rast_mirror_counts": (_i, [_i, _p, _z, _p, _p]),
    "eogs_rast_mirror_token": (_i, [_i, _i, _i, _p, _i, C.POINTER(_i64), C.POINTER(_i)]),
    "eogs_rast_capacity_token": (_i, [_i, _i64, C.c_double, _i, _i64, C.POINTER(_i64), C.POINTER(_i)]),
    "eogs_rast_forward_render": (
        _i,
        [_i, _i, _i, _i64, _p, _u, _p, _z, _p, _z, _p, _z, _p, _z, _p, _p, _p],
    ),
    "eogs_rast_backward": (
        _i,
        [_i, _i, _i, _i64]
        + [_p] * 7  # bg, means3D, radii, colors, opacities, scales, rotations
        + [_f, _p, _p, _p, _p, _u]  # scale_modifier, cov3D_precomp, viewmatrix, projmatrix, alt_affine, flags
        + [_p] * 4  # out_color, out_invdepth, dL_dout_color, dL_dout_invdepth
        + [_p, _z, _p, _z, _p, _z]  # geom, binning, image workspaces
        + [_p] * 9  # 7 gradients + dL_dT_sum + dL_dvm_mean
        + [_p, _i]  # dL_dcolors_lead, lead_cols
        + [_p],  # stream
    ),
    "eogs_rast_backward_range": (
        _i,
        [_i, _i, _i, _i64]
        + [_p] * 7
        + [_f, _p, _p, _p, _p, _u]
        + [_p] * 4
        + [_p, _z, _p, _z, _p, _z]
        + [_p] * 9
        + [_p, _i]  # dL_dcolors_lead, lead_cols
        + [_i, _i]  # p_begin, p_end
        + [_p],
    ),
    "eogs_rast_mark_visible": (_i, [_i, _p, _p, _p, _p, _p]),
    "eogs_rast_path_info": (_i, [_i, _i64, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "eogs_rast_backward_info": (_i, [_i, _i64, C.POINTER(_i)]),
    "eogs_rast_profile_enable": (_i, [_i]),
    "eogs_rast_profile_select": (_i, [_u]),
    "eogs_rast_profile_reset": (_i, []),
    "eogs_rast_profile_slots": (_i, []),
    "eogs_rast_profile_get": (_i, [_i, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_char_p)]),
    "eogs_rast_selftest": (_i, [_p, C.POINTER(_u), _p]),
    # include/eogs_loss.h
    "eogs_loss_bytes": (_i, [_i, _i, _i, _u, C.POINTER(_z)]),
    "eogs_loss_forward": (_i, [_i, _i, _i, _p, _p, _u, _f, _f, _f, _p, _p, _p, _z, _p]),
    "eogs_loss_backward": (_i, [_i, _i, _i, _p, _p, _u, _f, _f, _p, _p, _p, _z, _p, _p]),
    # include/eogs_optim.h
    "eogs_adam_step": (_i, [_i, _p, C.c_double, C.c_double, C.c_double, _i64, _p]),
    "eogs_sum_into": (_i, [_i, _p, _i, _p]),
    "eogs_pack_columns": (_i, [_i64, _i, _p, _p, _i, _i, _p]),
    "eogs_compact_bytes": (_i, [_i64, C.POINTER(_z)]),
    "eogs_compact_plan": (_i, [_i64, _p, _p, _z, C.POINTER(_i64), _p]),
    "eogs_compact_apply": (_i, [_i64, _p, _i, _p, _p, _p, _p, _z, _p]),
    # include/eogs_resample.h
    "eogs_resample_forward": (_i, [_i, _i, _i, _i, _i, _i, _p, _p, _p, _i, _f, _p, _p, _p]),
    "eogs_resample_bytes": (_i, [_i, _i, C.POINTER(_z)]),
    "eogs_resample_backward": (_i, [_i, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _z, _p]),
    # include/eogs_knn.h
    "eogs_knn_bytes": (_i, [_i, C.POINTER(_z)]),
    "eogs_knn_mean_dist2": (_i, [_i, _p, _p, _p, _z, _p]),
    # include/eogs_shade.h
    "eogs_shade_bytes": (_i, [_i, _i, C.POINTER(_z)]),
    "eogs_shade_forward": (_i, [_i, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "eogs_shade_backward": (_i, [_i, _i] + [_p] * 10 + [_p, _z, _p]),
    "eogs_mloss_forward": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _z, _p]),
    "eogs_mloss_backward": (_i, [_i, _i, _i] + [_p] * 9 + [_p]),
    "eogs_tshadow_forward": (_i, [_i64, _p, _p, _p, _z, _p]),
    "eogs_tshadow_backward": (_i, [_i64, _p, _p, _p, _p]),
    # include/eogs_tsdf.h
    "eogs_tsdf_integrate": (_i, [_i, _i, _i, _p, _p, _p, _p, _f, _f, _i, _i, _p, _p, _p, _p, _p]),
}
# symbols only the HIP library exports (the CPU oracle of the loss is oracle/loss_oracle.py, not a C-ABI twin)
HIP_ONLY = ("eogs_sum_into", "eogs_pack_columns", "eogs_loss_bytes", "eogs_loss_forward", "eogs_loss_backward", "eogs_adam_step", "eogs_compact_bytes",
            "eogs_compact_plan", "eogs_compact_apply", "eogs_resample_forward", "eogs_resample_bytes", "eogs_resample_backward", "eogs_knn_bytes",
            "eogs_knn_mean_dist2", "eogs_shade_bytes", "eogs_shade_forward", "eogs_shade_backward", "eogs_mloss_forward",
            "eogs_mloss_backward", "eogs_tshadow_forward", "eogs_tshadow_backward", "eogs_tsdf_integrate")


class PackTensor(C.Structure):
    """eogs_pack_tensor (include/eogs_optim.h)"""

    _fields_ = [("data", _p), ("width", _i), ("col0", _i), ("ncols", _i)]


class SumTensor(C.Structure):
    """eogs_sum_tensor (include/eogs_optim.h)"""

    _fields_ = [("dst", _p), ("src", _p * 4), ("numel", _i64)]


class AdamTensor(C.Structure):
    """eogs_adam_tensor (include/eogs_optim.h)"""

    _fields_ = [("param", _p), ("grad", _p), ("exp_avg", _p), ("exp_avg_sq", _p), ("numel", _i64), ("lr", _f)]


class RastError(RuntimeError):
    """A C-ABI call returned a negative status."""

    def __init__(self, code, message):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {message}")
        self.code = code


class RastABI:
    def __init__(self, path):
        self.path = str(path)
        self.cdll = C.CDLL(self.path)
        self.cdll.eogs_rast_backend.restype = C.c_char_p
        oracle_lib = self.cdll.eogs_rast_backend().decode() == "cpu-oracle"
        for name, (res, args) in SIGNATURES.items():
            if oracle_lib and name in HIP_ONLY:
                continue
            fn = getattr(self.cdll, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        v = self.cdll.eogs_rast_abi_version()
        if v != ABI_VERSION:
            raise RuntimeError(f"{path}: ABI version {v}, expected {ABI_VERSION}")
        self.backend = self.cdll.eogs_rast_backend().decode()
        # which torch device type this library's pointers live on
        self.device_type = "cpu" if self.backend == "cpu-oracle" else "cuda"

    def check(self, code):
        if code != EOGS_OK:
            raise RastError(code, self.cdll.eogs_rast_last_error().decode())

    def __getattr__(self, name):
        short = name.startswith(("loss_", "adam_", "sum_", "pack_", "compact_", "resample_", "knn_", "shade_", "mloss_", "tshadow_", "tsdf_"))
        return getattr(self.cdll, ("eogs_" if short else "eogs_rast_") + name)

    def path_info(self, P, num_rendered):
        """(list block in pixels, forward kernel variant, backward kernel variant) of a forward (include/eogs_rast.h)."""
        b, f, w = _i(), _i(), _i()
        self.check(self.cdll.eogs_rast_path_info(int(P), int(num_rendered), C.byref(b), C.byref(f), C.byref(w)))
        return b.value, f.value, w.value

    def backward_info(self, P, num_rendered):
        """Which build of the per-Gaussian backward kernel a backward of this token would launch now (include/eogs_rast.h)."""
        w = _i()
        self.check(self.cdll.eogs_rast_backward_info(int(P), int(num_rendered), C.byref(w)))
        return w.value

    def profile_slot_names(self):
        names = []
        for i in range(self.cdll.eogs_rast_profile_slots()):
            ms, n, nm = C.c_double(), _i64(), C.c_char_p()
            self.check(self.cdll.eogs_rast_profile_get(i, C.byref(ms), C.byref(n), C.byref(nm)))
            names.append(nm.value.decode())
        return names

    def profile(self):
        """{group name: (total device ms, launches)} accumulated since the last profile_reset()."""
        out = {}
        for i in range(self.cdll.eogs_rast_profile_slots()):
            ms, n, nm = C.c_double(), _i64(), C.c_char_p()
            self.check(self.cdll.eogs_rast_profile_get(i, C.byref(ms), C.byref(n), C.byref(nm)))
            out[nm.value.decode()] = (ms.value, n.value)
        return out
