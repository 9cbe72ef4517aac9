"""GPU (MI355X): the quad sub-list render kernels (render_fwd_quad_kernel, render_bwd_quad_kernel) against
the one-list-per-tile kernels on the same inputs. The quad masks only drop (pixel, entry) evaluations that cannot blend, so
both directions must agree to rounding: a few ulp in the forward (the compiler contracts `power` differently in the two
unrolled copies of the entry loop, and an entry sits at an even position in one kernel and an odd one in the other), the
summation order of the per-quad partial sums in the backward. Far tighter than the oracle tolerance: a dropped
(pixel, entry) would show as ~alpha*T*|c| >= 4e-3 |c|. The switches are read once per process, hence two child processes
(run one after the other)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from util import assert_close

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _render(tmp_path, tag, env_extra, args):
    out = os.path.join(str(tmp_path), f"{tag}.npz")
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(HERE, "quad_child.py"), out, *map(str, args)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(out)
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("P,H,W,opacity,invdepth", [(120_000, 344, 392, "init", 0), (60_000, 256, 250, "0.04", 1)])
def test_quad_kernels_agree_with_tile_kernels(tmp_path, P, H, W, opacity, invdepth):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    args = (P, H, W, opacity, invdepth)
    # per-tile lists in both runs; quad kernels off / on
    plain = _render(tmp_path, "plain", {"EOGS_BLOCK_SWITCH": "1000", "EOGS_QUAD_SWITCH": "0", "EOGS_QUAD_BWD_SWITCH": "0"}, args)
    quad = _render(tmp_path, "quad", {"EOGS_BLOCK_SWITCH": "1000", "EOGS_QUAD_SWITCH": "1000", "EOGS_QUAD_BWD_SWITCH": "1000"}, args)
    assert np.array_equal(plain["out_radii"], quad["out_radii"])
    for k in ("out_color", "out_invdepth"):
        assert_close(torch.from_numpy(quad[k]), torch.from_numpy(plain[k]), k, rtol=2e-6, flip_rtol=1e-2)
    assert float(np.abs(plain["out_color"]).max()) > 0.1
    for k in plain:
        if k.startswith("g_"):
            assert_close(torch.from_numpy(quad[k]), torch.from_numpy(plain[k]), k, rtol=2e-5)
