"""CPU: the virtual-camera resample oracle (oracle/resample_oracle.py, float64) against vectors produced by the REFERENCE's
own render_resample_virtual_camera (tests/golden/make_golden_resample.py: renderer_cc_shadow.py:5-50 executed with a seeded
virtual render): sampled RGB and altitude incl. the -100 fill, the sampling coordinates, and the autograd gradients with
respect to the virtual render and the true camera's altitude."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import resample_oracle
from util import GOLDEN_DIR, assert_close

FIXTURES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "resample_*.npz")))


def load(path):
    z = np.load(path)
    return {k: torch.as_tensor(z[k]) for k in z.files}


def run(fn, c, dev=torch.device("cpu")):
    """The case through `fn(virtual_render, cam2virt, uva) -> (sample[>=4,H,W], uv)`; returns what the fixture stores."""
    vr = c["virtual_render"].to(dev).clone().requires_grad_(True)
    alt = c["altitude"].to(dev).clone().requires_grad_(True)
    uva = torch.stack((c["U"].to(dev), c["V"].to(dev), alt), dim=-1)
    s, uv = fn(vr, c["cam2virt"].to(dev), uva)
    ((s[:3] * c["w_rgb"].to(dev)).sum() + (s[3] * c["w_alt"].to(dev)).sum() + (uv * c["w_uv"].to(dev)).sum()).backward()
    return dict(rgb_sample=s[:3].detach().cpu(), altitude_sample=s[3].detach().cpu(), virtual_uv=uv.detach().cpu(),
                g_virtual_render=vr.grad.cpu(), g_altitude=alt.grad.cpu())


def compare(got, c, rtol):
    filled = c["altitude_sample"] == -100
    assert torch.equal(got["altitude_sample"] == -100, filled), "the out-of-view fill covers different pixels"
    for k in ("rgb_sample", "altitude_sample", "virtual_uv", "g_virtual_render", "g_altitude"):
        assert_close(got[k].float(), c[k], k, rtol=rtol, allow_flips=False)


def test_fixtures_present():
    assert len(FIXTURES) >= 4


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[9:-4])
def test_oracle_matches_reference_vectors(path):
    c = load(path)
    # the reference runs in fp32, the oracle in float64: bilinear weights of fp32 pixel coordinates up to ~128 agree to ~1e-5
    compare(run(resample_oracle.resample, c), c, rtol=5e-5)
