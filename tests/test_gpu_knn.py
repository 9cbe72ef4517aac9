"""GPU (MI355X): `distCUDA2` (include/eogs_knn.h) against an exact k-d tree (scipy.spatial.cKDTree) — the quantity the
reference computes is the exact mean squared distance to the three nearest neighbours (simple_knn.cu:147-185)."""
import numpy as np
import pytest
import torch
from scipy.spatial import cKDTree

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from eogs2_amd import _lib

    assert _lib.get().backend == "hip-gfx950"
    return torch.device("cuda:0")


def _exact(pts):
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    return (d[:, 1:] ** 2).mean(1)


@pytest.mark.parametrize("P,kind", [(4, "uniform"), (1000, "uniform"), (100_003, "uniform"), (60_000, "scene"),
                                    (30_000, "clustered"), (5000, "planar")])
def test_dist2_matches_exact_knn(dev, P, kind):
    from simple_knn._C import distCUDA2  # the import path of the reference

    g = np.random.default_rng(P)
    if kind == "uniform":
        pts = g.random((P, 3))
    elif kind == "scene":   # the normalised EOGS scene box: thin in z
        pts = g.random((P, 3)) * [1.8, 1.8, 0.2] - [0.9, 0.9, 0.05]
    elif kind == "clustered":
        pts = g.normal(size=(P, 3)) * 0.01 + g.integers(0, 5, (P, 1)) * 3.0
    else:                    # one flat axis: the Morton normalisation of that axis is degenerate
        pts = np.concatenate([g.random((P, 2)), np.zeros((P, 1))], 1)
    pts = pts.astype(np.float32)
    got = distCUDA2(torch.from_numpy(pts).to(dev)).cpu().numpy()
    ref = _exact(pts)
    assert got.shape == (P,) and got.dtype == np.float32
    assert np.allclose(got, ref, rtol=2e-5, atol=1e-12), float(np.abs(got - ref).max())


def test_duplicates_and_tiny_inputs(dev):
    from eogs2_amd.knn import distCUDA2

    pts = torch.tensor([[0.0, 0, 0], [0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 0]], device=dev)
    got = distCUDA2(pts).cpu().numpy()
    assert np.allclose(got, [1 / 3, 1 / 3, 1.0, 4.0, 1 / 3])  # coincident points are neighbours at distance 0
    assert distCUDA2(torch.zeros(0, 3, device=dev)).shape == (0,)
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(5, 2, device=dev))
