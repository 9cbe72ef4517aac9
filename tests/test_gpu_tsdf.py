"""GPU (MI355X): TSDF integration (include/eogs_tsdf.h, eogs2_amd/tsdf.py) through the C-ABI against the restatement of
the reference's statements (oracle/tsdf_oracle.py) in float64 on the CPU and in fp32 on the GPU (the op sequence the
reference executes), plus size-independent properties on a 4M-voxel volume."""
import types

import numpy as np
import pytest
import torch

from oracle import tsdf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from eogs2_amd import _lib

    assert _lib.get().backend == "hip-gfx950"
    return torch.device("cuda:0")


def scene(H, W, seed, shear=0.15):
    """A near-nadir affine camera (coef ~ the reference's Nadir model, to_affine.py:244-249, plus shear) looking at a
    smooth altitude field; the volume is larger than the footprint so part of it falls outside the view."""
    g = torch.Generator().manual_seed(seed)
    coef = torch.tensor([[0.0, 0.9, 0.0], [0.9, 0.0, 0.0], [0.0, 0.0, 1.0]])
    coef[:2, 2] = shear * torch.randn(2, generator=g)
    intercept = torch.tensor([0.02, -0.03, 0.1])
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    alt = (0.15 * torch.sin(3 * xx) * torch.cos(2 * yy) + 0.05 * torch.rand((H, W), generator=g))[None, None]
    wgt = torch.rand((1, 1, H, W), generator=g).clamp(0.0, 1.0)
    wgt[..., : H // 4, : W // 4] = 0.0  # zero-weight region: 0/0 -> NaN exactly as in the reference
    return coef, intercept, alt, wgt


def assert_same(a, b, tol, what):
    a, b = a.double().cpu(), b.double().cpu()
    nan_a, nan_b = torch.isnan(a), torch.isnan(b)
    assert torch.equal(nan_a, nan_b), f"{what}: NaN pattern differs"
    d = (a - b)[~nan_a].abs()
    bad = d > tol
    # a voxel within rounding of the truncation boundary or of |u| = 1 may legitimately fall on the other side
    assert bad.sum() <= 2e-5 * a.numel() + 1, f"{what}: {int(bad.sum())} of {a.numel()} voxels differ by more than {tol}"


@pytest.mark.parametrize("H,W,bounds,vox", [
    (48, 64, [[-1.4, 1.4], [-1.3, 1.3], [-0.3, 0.4]], 0.06),
    (33, 17, [[-0.5, 0.5], [-0.5, 0.5], [-0.2, 0.3]], 0.031),
    (64, 64, [[-4.0, -3.5], [-0.2, 0.2], [0.0, 0.1]], 0.05),  # entirely outside the view: nothing changes
])
def test_integrate_matches_restatement(dev, H, W, bounds, vox):
    from eogs2_amd.tsdf import TSDFVolume

    scale, fact = 1.7, 3.0
    vol = TSDFVolume(np.array(bounds), vox, fact, device=dev)
    t_ref = torch.ones(vol.num_voxels_per_dimension, dtype=torch.float64)
    w_ref = torch.zeros(vol.num_voxels_per_dimension, dtype=torch.float64)
    t32 = torch.ones(vol.num_voxels_per_dimension, device=dev)
    w32 = torch.zeros(vol.num_voxels_per_dimension, device=dev)
    axes64 = [a.double().cpu() for a in vol.axes]
    for seed in (1, 2, 3):  # three views accumulated
        coef, intercept, alt, wgt = scene(H, W, seed)
        ri = types.SimpleNamespace(affine_model=(coef.to(dev), intercept.to(dev)), model_scale=scale, altitude_img=alt.to(dev),
                                   get_weights=lambda wgt=wgt: wgt.to(dev))
        vol.integrate(ri)
        t_ref, w_ref = O.integrate(t_ref, w_ref, axes64, coef.double(), intercept.double(), scale, fact * vox, alt.double(), wgt.double())
        t32, w32 = O.integrate(t32, w32, vol.axes, coef.to(dev), intercept.to(dev), scale, fact * vox, alt.to(dev), wgt.to(dev))
    assert_same(vol._tsdf_vol, t_ref, 2e-4, "tsdf vs float64 restatement")
    assert_same(vol._weight_vol, w_ref, 2e-4, "weights vs float64 restatement")
    assert_same(vol._tsdf_vol, t32, 2e-5, "tsdf vs fp32 op sequence")
    assert_same(vol._weight_vol, w32, 2e-5, "weights vs fp32 op sequence")
    if bounds[0][1] < -3.0:
        assert torch.equal(vol._tsdf_vol, torch.ones_like(vol._tsdf_vol)) and float(vol._weight_vol.abs().max()) == 0.0


def test_full_size_properties(dev):
    """256 x 256 x 64 voxels, 1024^2 image: determinism; integrating the same image twice leaves the TSDF where it was
    and doubles the weights; untouched voxels keep their initial values."""
    from eogs2_amd.tsdf import TSDFVolume

    coef, intercept, alt, wgt = scene(1024, 1024, 5)
    wgt = wgt.clamp(min=0.05)
    ri = types.SimpleNamespace(affine_model=(coef.to(dev), intercept.to(dev)), model_scale=1.0, altitude_img=alt.to(dev),
                               get_weights=lambda: wgt.to(dev))
    bounds = np.array([[-1.2, 1.2], [-1.2, 1.2], [-0.3, 0.35]])
    a = TSDFVolume(bounds, 2.4 / 255, 4.0, device=dev)
    b = TSDFVolume(bounds, 2.4 / 255, 4.0, device=dev)
    assert a._tsdf_vol.numel() >= 4_000_000
    a.integrate(ri)
    b.integrate(ri)
    assert torch.equal(a._tsdf_vol, b._tsdf_vol) and torch.equal(a._weight_vol, b._weight_vol)
    touched = a._weight_vol > 0
    assert 0.05 < float(touched.float().mean()) < 0.95
    assert torch.equal(a._tsdf_vol[~touched], torch.ones_like(a._tsdf_vol[~touched]))
    b.integrate(ri)
    assert torch.allclose(b._weight_vol, 2 * a._weight_vol, rtol=1e-6, atol=0)
    assert torch.allclose(b._tsdf_vol, a._tsdf_vol, rtol=0, atol=2e-6)
    assert float(a._tsdf_vol.max()) <= 1.0 and float(a._tsdf_vol[touched].min()) >= -1.0 - 1e-6


# ---- vectors from the REFERENCE's own tsdf.py (tests/golden/make_golden_tsdf.py; tests/test_tsdf_oracle.py pins the restatement) ----
import glob  # noqa: E402
import os  # noqa: E402

from util import GOLDEN_DIR  # noqa: E402

TSDF_FIXTURES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "tsdf_*.npz")))


@pytest.mark.parametrize("path", TSDF_FIXTURES, ids=lambda p: os.path.basename(p)[5:-4])
def test_integrate_matches_reference_vectors(dev, path):
    """eogs2_amd.tsdf.TSDFVolume (constructor + one HIP kernel per view) against the volumes the reference's
    TSDFVolume.integrate produced after every accumulated view, NaN voxels of the zero-weight region included."""
    from eogs2_amd.tsdf import TSDFVolume

    z = np.load(path)
    c = {k: z[k] for k in z.files}
    vol = TSDFVolume(c["vol_bounds"], float(c["vox_size"]), float(c["trunc_margin_fact"]), device=dev)
    assert tuple(vol.num_voxels_per_dimension) == tuple(int(x) for x in c["num_voxels"])
    for i in range(3):
        assert torch.equal(vol.axes[i].cpu(), torch.as_tensor(c[f"axis{i}"]))
    t = lambda k: torch.as_tensor(c[k]).to(dev)
    for v in range(int(c["n_views"])):
        ri = types.SimpleNamespace(affine_model=(t(f"v{v}_coef"), t(f"v{v}_intercept")), model_scale=float(c["model_scale"]),
                                   altitude_img=t(f"v{v}_altitude"), get_weights=lambda v=v: t(f"v{v}_weights"))
        vol.integrate(ri)
        assert_same(vol._tsdf_vol, torch.as_tensor(c[f"v{v}_tsdf_vol"]), 2e-5, f"view {v}: tsdf vs the reference's volume")
        assert_same(vol._weight_vol, torch.as_tensor(c[f"v{v}_weight_vol"]), 2e-5, f"view {v}: weights vs the reference's volume")
    if "outside" in path:
        assert torch.equal(vol._tsdf_vol, torch.ones_like(vol._tsdf_vol)) and float(vol._weight_vol.abs().max()) == 0.0
