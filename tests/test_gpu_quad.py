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


@pytest.mark.parametrize("P,H,W,opacity,invdepth", [(150_000, 340, 390, "init", 0), (80_000, 256, 250, "0.04", 1),
                                                    (60_000, 200, 264, "trained", 0), (100_000, 304, 296, "surface", 0)])
def test_round5_fast_paths_change_no_bit(tmp_path, P, H, W, opacity, invdepth):
    """Round 5: the forward's plain chunks leave out operations that are no-ops where they are left out,
    the backward's flag-free records add exact zeros, and the quad masks the backward takes over from the forward are the
    ones it would compute itself: with all of it switched off, on (the default picks per scene and per chunk) or forced, every
    output and every gradient is the same BITS (-0.0 and +0.0 count as equal: a zero record added to
    a zero sum). Two of the images are off the 8-px grid (340 x 390: both edges; 256 x 250: one): edge tiles keep the general loop beside
    interior tiles' plain chunks; the 'trained' scene saturates, so its forced flag-free backward walks dead entries; the 'surface'
    scene has the size spread in which waves of the per-Gaussian backward sum some Gaussians together (GB_COOP): which ones must not
    depend on the build (round 6: an estimate in units of the build's own trip width made eager and replayed steps differ)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    args = (P, H, W, opacity, invdepth)
    quad = {"EOGS_BLOCK_SWITCH": "1000", "EOGS_QUAD_SWITCH": "1000", "EOGS_QUAD_BWD_SWITCH": "1000"}
    off = _render(tmp_path, "off", dict(quad, EOGS_PLAIN_TRIPS="0", EOGS_NOFLAG="0", EOGS_FWD_MASKS="0", EOGS_GB_WIDE="0"), args)
    auto = _render(tmp_path, "auto", dict(quad), args)
    # (EOGS_GB_WIDE: the per-Gaussian backward with four / eight records in flight per lane, csrc/preprocess.hip gaussian_bwd_wide:
    # the same sums in the same order)
    forced = _render(tmp_path, "forced", dict(quad, EOGS_NOFLAG="2", EOGS_GB_WIDE="2"), args)
    wide1 = _render(tmp_path, "wide1", dict(quad, EOGS_GB_WIDE="1"), args)
    assert float(np.abs(off["out_color"]).max()) > 0.1 and float(np.abs(off["g_means3D"]).max()) > 0
    for name, got in (("default", auto), ("forced flag-free, eight records per trip", forced), ("eight records per long-list trip", wide1)):
        for k in off:
            d = off[k] != got[k]  # (numpy: -0.0 == 0.0; a NaN would differ from itself and fail here as it should)
            assert not d.any(), f"{name}: {k} differs in {int(d.sum())} of {d.size} elements from the run with the fast paths off"


def test_backward_kernel_hint_travels_with_the_token():
    """The per-Gaussian backward has a narrow and two wide builds (csrc/preprocess.hip gaussian_bwd_wide); which one a backward
    launches depends on the forward's list depth x mean pair opacity, which the host learns with the counts and packs into the
    token (bit 60; include/eogs_rast.h, forward_prepare). Shallow scene -> a wide build; saturating scene -> the narrow
    one; a capacity token counted on a forward inherits that forward's hint; the hint is the token's alone — no table
    beside it that the order or number of earlier forwards could change (rounds 5's 16-entry table)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import ctypes

    from eogs2_amd import GaussianRasterizer, _lib
    from eogs2_amd.rasterizer import last_exact_token
    from eogs2_amd.synthetic import make_camera, make_scene, settings_for

    if os.environ.get("EOGS_GB_WIDE") is not None:
        pytest.skip("EOGS_GB_WIDE forces one build")
    abi = _lib.get()
    dev = torch.device("cuda:0")
    P, H, W = 200_000, 256, 256  # (trained opacities: ~2000 listed pairs per tile x a mean opacity of ~0.5, far beyond the switch at 256)

    def token_of(opacity):
        sc = make_scene(P, H, W, seed=3, opacity=opacity, device=dev)
        sc["viewmatrix"] = make_camera(H, W, seed=3, device=dev)
        rast = GaussianRasterizer(settings_for(sc, H, W))
        with torch.no_grad():
            rast(sc["means3D"], torch.zeros(P, 3, device=dev), sc["opacities"], colors_precomp=sc["colors"], scales=sc["scales"],
                 rotations=sc["rotations"])
        torch.cuda.synchronize()
        return int(last_exact_token(dev))

    shallow, deep = token_of("init"), token_of("trained")
    assert shallow > 0 and deep > 0
    assert abi.backward_info(P, shallow) in (1, 2)
    assert abi.backward_info(P, deep) == 0
    cap = ctypes.c_int64()
    abi.check(abi.capacity_token(P, shallow, 0.25, 1, shallow, ctypes.byref(cap), None))
    assert cap.value != shallow and abi.backward_info(P, cap.value) == abi.backward_info(P, shallow)
    abi.check(abi.capacity_token(P, deep, 0.25, 1, deep, ctypes.byref(cap), None))
    assert abi.backward_info(P, cap.value) == 0
    for _ in range(40):  # (forty other forwards in between: the hint travels in the token, nothing to evict)
        token_of("trained")
    assert abi.backward_info(P, shallow) in (1, 2)
    assert abi.backward_info(P, shallow & ~(1 << 60)) == 0  # the same counts without the hint bit: the narrow build
