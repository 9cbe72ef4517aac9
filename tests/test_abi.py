"""CPU: the HIP C-ABI library loads and exports every symbol include/eogs_rast.h declares; the host wrapper
mirrors the reference's argument checks. No compute call is made (there is no GPU here)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(headers=("eogs_rast.h", "eogs_loss.h", "eogs_optim.h", "eogs_resample.h", "eogs_knn.h", "eogs_shade.h", "eogs_tsdf.h")):
    out = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        out |= set(re.findall(r"\b(eogs_(?:rast|loss|adam|sum|pack|compact|resample|knn|shade|mloss|tshadow|tsdf)_[a-z_0-9]+)\s*\(", src))
    return sorted(out)


def test_header_and_binding_agree():
    from eogs2_amd._abi import SIGNATURES

    assert header_symbols() == sorted(SIGNATURES)


@pytest.fixture(scope="module")
def hip_lib():
    from eogs2_amd import build

    build.build(verbose=False)
    from eogs2_amd import _lib

    return _lib.get()


def test_hip_library_exports_everything(hip_lib):
    assert hip_lib.backend == "hip-gfx950"
    assert hip_lib.cdll.eogs_rast_abi_version() == 8
    for name in header_symbols():
        assert hasattr(hip_lib.cdll, name), name


def test_hip_size_queries_and_arg_checks(hip_lib):
    n = ctypes.c_size_t()
    hip_lib.check(hip_lib.geom_bytes(1 << 20, ctypes.byref(n)))
    assert 40e6 < n.value < 200e6
    hip_lib.check(hip_lib.image_bytes(1024, 1024, ctypes.byref(n)))
    assert n.value >= 8 * 1024 * 1024
    hip_lib.check(hip_lib.binning_bytes(1 << 20, 1024, 1024, 10_000_000, ctypes.byref(n)))
    assert n.value >= 10_000_000 * 16
    assert hip_lib.geom_bytes(-1, ctypes.byref(n)) == -1
    assert b"geom_bytes" in hip_lib.cdll.eogs_rast_last_error()
    R = ctypes.c_int64(7)
    # NULL inputs are rejected before anything touches a device
    assert hip_lib.forward_prepare(10, 32, 32, None, None, None, None, None, None, 1.0, None, None, None, 0, None, None, 0,
                                   None, 0, ctypes.byref(R), None) == -1
    assert R.value == 0
    # P == 0 is a no-op
    assert hip_lib.forward_prepare(0, 32, 32, None, None, None, None, None, None, 1.0, None, None, None, 0, None, None, 0,
                                   None, 0, ctypes.byref(R), None) == 0
    hip_lib.check(hip_lib.scratch_bytes(1 << 20, 1024, 1024, ctypes.byref(n)))
    assert 100e6 < n.value < 300e6  # transient, shared by every forward on a stream


def test_capacity_token_and_counts_without_a_pending_forward(hip_lib):
    """Host-only pieces of the deferred count readback (EOGS_FLAG_DEFER_COUNTS): the capacity token keeps the list
    granularity, holds `slack` more slots and entries, claims "sorted in scratch" only if everything that fits it fits
    the scratch (6 entries per Gaussian), and `fits` compares both counts."""
    def pack(slots, entries, block, sorted_):
        return (block << 62) | (sorted_ << 61) | (entries << 32) | slots

    cap, fits = ctypes.c_int64(), ctypes.c_int()
    P = 100_000
    last = pack(1_000_000, 200_000, 0, 1)
    hip_lib.check(hip_lib.capacity_token(P, last, 0.25, 1, pack(1_200_000, 250_000, 0, 1), ctypes.byref(cap), ctypes.byref(fits)))
    c = cap.value
    assert (c & 0x7FFFFFFF) == 1_250_000 + 4096 and ((c >> 32) & 0x07FFFFFF) == 250_000 + 1024
    assert (c >> 61) & 1 == 1 and (c >> 62) & 1 == 0 and fits.value == 1
    hip_lib.check(hip_lib.capacity_token(P, last, 0.25, 1, pack(1_254_097, 250_000, 0, 1), ctypes.byref(cap), ctypes.byref(fits)))
    assert fits.value == 0
    hip_lib.check(hip_lib.capacity_token(P, last, 0.25, 1, pack(1_000, 251_025, 0, 1), ctypes.byref(cap), ctypes.byref(fits)))
    assert fits.value == 0
    # 500 k entries x 1.25 exceed the scratch of 100 k Gaussians (600 k entries): the forward sorts in its binning workspace
    hip_lib.check(hip_lib.capacity_token(P, pack(9_000_000, 500_000, 1, 1), 0.25, 1, 0, ctypes.byref(cap), None))
    assert (cap.value >> 61) & 1 == 0 and (cap.value >> 62) & 1 == 1
    hip_lib.check(hip_lib.capacity_token(P, last, 0.25, 0, 0, ctypes.byref(cap), None))
    assert (cap.value >> 61) & 1 == 0  # no scratch, no sort in scratch
    assert hip_lib.capacity_token(P, last, -1.0, 1, 0, ctypes.byref(cap), None) == -1
    R = ctypes.c_int64(5)
    assert hip_lib.forward_counts(ctypes.byref(R)) == -1 and R.value == 0
    assert b"pending" in hip_lib.cdll.eogs_rast_last_error()


def test_oracle_defers_nothing(oracle_backend):
    """The checker library keeps the ABI's shape: with EOGS_FLAG_DEFER_COUNTS forward_prepare hands back 0 and
    forward_counts the token, once; its capacity token is the exact token."""
    import numpy as np
    from eogs2_amd._abi import FLAG_DEFER_COUNTS
    from eogs2_amd.synthetic import make_scene

    a = oracle_backend
    P, H, W = 200, 48, 48
    sc = {k: np.ascontiguousarray(v.numpy()) for k, v in make_scene(P, H, W, seed=2).items()}
    ptr = lambda x: ctypes.c_void_p(x.ctypes.data)  # noqa: E731
    n = ctypes.c_size_t()
    a.check(a.geom_bytes(P, ctypes.byref(n)))
    geom, radii = np.zeros(n.value, np.uint8), np.zeros(P, np.int32)
    tokens = []
    for flags in (0, FLAG_DEFER_COUNTS):
        R = ctypes.c_int64(-1)
        a.check(a.forward_prepare(P, H, W, ptr(sc["means3D"]), ptr(sc["scales"]), ptr(sc["rotations"]), None, ptr(sc["opacities"]),
                                  ptr(sc["colors"]), 1.0, ptr(sc["viewmatrix"]), ptr(sc["viewmatrix"]), None, flags, ptr(radii),
                                  ptr(geom), geom.size, None, 0, ctypes.byref(R), None))
        tokens.append(R.value)
    assert tokens[0] > 0 and tokens[1] == 0
    R = ctypes.c_int64(-1)
    a.check(a.forward_counts(ctypes.byref(R)))
    assert R.value == tokens[0]
    assert a.forward_counts(ctypes.byref(R)) == -1  # consumed
    cap, fits = ctypes.c_int64(), ctypes.c_int()
    a.check(a.capacity_token(P, tokens[0], 0.25, 0, tokens[0], ctypes.byref(cap), ctypes.byref(fits)))
    assert cap.value == tokens[0] and fits.value == 1


def test_oracle_library_exports_everything():
    import oracle

    a = oracle.abi()
    assert a.backend == "cpu-oracle"
    for name in header_symbols(("eogs_rast.h",)):  # the loss oracle is oracle/loss_oracle.py (numpy), not a C-ABI twin
        assert hasattr(a.cdll, name), name


def test_product_never_imports_oracle():
    for d in ("eogs2_amd", "diff_gaussian_rasterization"):
        for fn in os.listdir(os.path.join(ROOT, d)):
            if fn.endswith(".py"):
                src = open(os.path.join(ROOT, d, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{d}/{fn} imports the oracle"


def test_no_cpu_fallback_without_gpu():
    """CPU tensors + the HIP library -> loud failure (no silent CPU path)."""
    from eogs2_amd import GaussianRasterizer
    from eogs2_amd.synthetic import make_scene, settings_for

    sc = make_scene(10, 32, 32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        GaussianRasterizer(settings_for(sc, 32, 32))(sc["means3D"], torch.zeros(10, 3), sc["opacities"],
                                                      colors_precomp=sc["colors"], scales=sc["scales"],
                                                      rotations=sc["rotations"])


def test_argument_validation_matches_reference(oracle_backend):
    """Same exceptions as DGR/diff_gaussian_rasterization/__init__.py:263-275 and rasterize_points.cu:58-60."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from eogs2_amd.synthetic import make_scene, settings_for

    sc = make_scene(10, 32, 32)
    r = GaussianRasterizer(settings_for(sc, 32, 32))
    m2 = torch.zeros(10, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(sc["means3D"], m2, sc["opacities"], scales=sc["scales"], rotations=sc["rotations"])
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(sc["means3D"], m2, sc["opacities"], colors_precomp=sc["colors"], scales=sc["scales"])
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        r(sc["means3D"], m2, sc["opacities"], colors_precomp=sc["colors"], scales=sc["scales"],
          rotations=sc["rotations"], cov3D_precomp=torch.zeros(10, 6))
    with pytest.raises(RuntimeError, match="provide precomputed Gaussian colors"):
        r(sc["means3D"], m2, sc["opacities"], shs=torch.zeros(10, 1, 3), scales=sc["scales"], rotations=sc["rotations"])
    with pytest.raises(RuntimeError, match=r"means3D must have dimensions \(num_points, 3\)"):
        r(sc["means3D"][:, :2], m2, sc["opacities"], colors_precomp=sc["colors"], scales=sc["scales"],
          rotations=sc["rotations"])
    assert r.markVisible(sc["means3D"]).all()
