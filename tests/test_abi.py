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
        out |= set(re.findall(r"\b(eogs_(?:rast|loss|adam|pack|compact|resample|knn|shade|mloss|tshadow|tsdf)_[a-z_0-9]+)\s*\(", src))
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
    assert hip_lib.cdll.eogs_rast_abi_version() == 5
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
