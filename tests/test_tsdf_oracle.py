"""CPU: the TSDF restatement (oracle/tsdf_oracle.py) against vectors produced by the REFERENCE's own tsdf.py
(tests/golden/make_golden_tsdf.py: RangeImageEOGS.sample_sdf and TSDFVolume.__init__ / integrate executed unmodified on
the CPU): sample_sdf at every voxel centre, both volumes after every accumulated view including the 0 / 0 = NaN voxels of a
zero-weight region, and the constructor arithmetic the product's `volume_axes` repeats (voxel counts, axes)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import tsdf_oracle as O
from util import GOLDEN_DIR

FIXTURES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "tsdf_*.npz")))


def load(path):
    z = np.load(path)
    return {k: z[k] for k in z.files}


def test_fixtures_present():
    assert len(FIXTURES) >= 3


def same(a, b, tol, what):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    na, nb = torch.isnan(a), torch.isnan(b)
    assert torch.equal(na, nb), f"{what}: NaN pattern differs"
    d = (a - b)[~na].abs()
    assert d.numel() == 0 or float(d.max()) <= tol, f"{what}: {float(d.max()):.3e} > {tol}"


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[5:-4])
def test_restatement_matches_reference_vectors(path):
    c = load(path)
    t = lambda k, dt=torch.float32: torch.as_tensor(c[k]).to(dt)
    scale, margin = float(c["model_scale"]), float(c["trunc_margin"])
    axes = [t(f"axis{i}") for i in range(3)]
    world = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(-1, 3)
    dims = tuple(int(x) for x in c["num_voxels"])
    for dt, tol in ((torch.float32, 0.0), (torch.float64, 2e-4)):
        tv = torch.ones(dims, dtype=dt)
        wv = torch.zeros(dims, dtype=dt)
        for v in range(int(c["n_views"])):
            args = (t(f"v{v}_coef", dt), t(f"v{v}_intercept", dt), scale)
            alt, wgt = t(f"v{v}_altitude", dt), t(f"v{v}_weights", dt)
            sdf, mask, ws = O.sample_sdf(world.to(dt), *args, alt, wgt)
            if dt == torch.float32:  # the same torch ops in the same dtype: bit for bit
                assert torch.equal(mask, torch.as_tensor(c[f"v{v}_mask"]))
                assert torch.equal(sdf, t(f"v{v}_sdf")) and torch.equal(ws, t(f"v{v}_sampled_weights"))
            else:  # float64: a voxel within rounding of |u| = 1 may fall on the other side
                assert int((mask != torch.as_tensor(c[f"v{v}_mask"])).sum()) <= 2
                same(sdf, c[f"v{v}_sdf"], 2e-5 * max(1.0, float(np.abs(c[f"v{v}_sdf"]).max())), f"view {v} sdf")
            tv, wv = O.integrate(tv, wv, [a.to(dt) for a in axes], *args, margin, alt, wgt)
            if dt == torch.float32:
                same(tv, c[f"v{v}_tsdf_vol"], 0.0, f"view {v} tsdf (fp32)")
                same(wv, c[f"v{v}_weight_vol"], 0.0, f"view {v} weights (fp32)")
            else:
                bad = (tv - t(f"v{v}_tsdf_vol", dt)).abs() > tol  # NaN compares False
                assert int(bad.sum()) <= 2, f"view {v}: {int(bad.sum())} voxels differ in float64"
                assert torch.equal(torch.isnan(tv), torch.isnan(t(f"v{v}_tsdf_vol")))


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[5:-4])
def test_volume_constructor_arithmetic(path):
    """eogs2_amd.tsdf.volume_axes (pure torch; the part of the product's TSDFVolume that repeats tsdf.py:387-407)."""
    from eogs2_amd.tsdf import volume_axes

    c = load(path)
    dims, axes = volume_axes(c["vol_bounds"], float(c["vox_size"]), torch.device("cpu"))
    assert tuple(dims) == tuple(int(x) for x in c["num_voxels"])
    for i in range(3):
        assert torch.equal(axes[i], torch.as_tensor(c[f"axis{i}"]))
