"""GPU (MI355X): the quad sub-list render kernels (render_fwd_quad_kernel, render_bwd_quad_kernel) against
the one-list-per-tile kernels on the same inputs. The quad masks only drop (pixel, entry) evaluations that cannot blend, so
the forward must agree BITWISE (same per-pixel arithmetic in the same order; `power_of` fixes the rounding of the exponent
in every kernel) and the backward to the summation order of the per-quad partial sums. A dropped (pixel, entry) would show
as ~alpha*T*|c| >= 4e-3 |c|. The switches are read once per process, hence two child processes
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
    for k in ("out_radii", "out_color", "out_invdepth"):
        d = plain[k] != quad[k]
        assert not d.any(), (f"{k}: the quad forward is not bitwise identical: {int(d.sum())} of {d.size} elements, max |diff| "
                             f"{float(np.abs(plain[k].astype(np.float64) - quad[k].astype(np.float64)).max()):.3e}")
    assert float(np.abs(plain["out_color"]).max()) > 0.1
    for k in plain:
        if k.startswith("g_"):
            assert_close(torch.from_numpy(quad[k]), torch.from_numpy(plain[k]), k, rtol=2e-5)
