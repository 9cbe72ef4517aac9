"""CPU: pins the C oracle (oracle/rast_oracle.c) against the independent dense PyTorch/autograd renderer
(oracle/torch_dense.py): forward images, radii, and every gradient the reference returns.

Known, documented deviations of the REFERENCE from the true gradient (the oracle restates the reference):
  * antialiasing: the d(h_convolution_scaling)/d(cov) formula is evaluated at the +0.3 entries
    (DGR/cuda_rasterizer/backward.cu:226-236) -> scales/rotations differ from autograd at the 1e-3 level;
  * grad_viewmatrix[:3,:2] covariance term is scaled per ROW by NCD2Screen (DGR/.../__init__.py:183-190);
    we therefore pin sum(dL_dT) itself through a T_override leaf.
"""
import ctypes
import math

import pytest
import torch

from eogs2_amd.synthetic import make_scene, settings_for
from oracle.torch_dense import render_dense


def _oracle_run(sc, H, W, aa, depth_grad, oracle_backend):
    from eogs2_amd import GaussianRasterizer

    P = sc["means3D"].shape[0]
    leaves = {k: sc[k].clone().requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "colors"]}
    m2d = torch.zeros(P, 3, requires_grad=True)
    vm = sc["viewmatrix"].clone().requires_grad_(True)
    rs = settings_for(sc, H, W, antialiasing=aa)._replace(viewmatrix=vm, projmatrix=vm.detach())
    color, radii, invd = GaussianRasterizer(rs)(
        leaves["means3D"], m2d, leaves["opacities"], colors_precomp=leaves["colors"], scales=leaves["scales"],
        rotations=leaves["rotations"])
    loss = (color * sc["dL_dcolor"]).sum()
    if depth_grad is not None:
        loss = loss + (invd * depth_grad).sum()
    loss.backward()
    g = {k: v.grad for k, v in leaves.items()}
    g["means2D"], g["viewmatrix"] = m2d.grad, vm.grad
    return color.detach(), radii, invd.detach(), g


def _dense_run(sc, H, W, aa, depth_grad):
    P = sc["means3D"].shape[0]
    leaves = {k: sc[k].clone().requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "colors"]}
    m2d = torch.zeros(P, 3, requires_grad=True)
    vm = sc["viewmatrix"].clone().requires_grad_(True)
    s = torch.tensor([W / 2.0, H / 2.0])
    T_leaf = (sc["viewmatrix"][:3, :2].t() * s[:, None]).clone().requires_grad_(True)
    color, radii, invd = render_dense(
        leaves["means3D"], leaves["opacities"], leaves["colors"], sc["bg"], vm, H, W, scales=leaves["scales"],
        rotations=leaves["rotations"], antialiasing=aa, means2D=m2d, T_override=T_leaf, block=32)
    loss = (color * sc["dL_dcolor"]).sum()
    if depth_grad is not None:
        loss = loss + (invd * depth_grad).sum()
    loss.backward()
    g = {k: v.grad for k, v in leaves.items()}
    g["means2D"], g["viewmatrix"], g["T"] = m2d.grad, vm.grad, T_leaf.grad
    return color.detach(), radii, invd.detach(), g


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


CASES = [
    # P, H, W, seed, opacity, scale_mult, aa, depth_grad
    (400, 64, 64, 0, "init", 2.5, False, False),
    (300, 48, 80, 1, "trained", 3.0, False, True),
    (300, 40, 56, 2, "trained", 3.0, True, True),
    (1500, 32, 32, 3, 0.6, 10.0, False, False),  # > 256 Gaussians per tile, early termination
]


@pytest.mark.parametrize("P,H,W,seed,opacity,scale_mult,aa,dgrad", CASES)
def test_oracle_matches_dense_autograd(P, H, W, seed, opacity, scale_mult, aa, dgrad, oracle_backend):
    sc = make_scene(P, H, W, seed=seed, opacity=opacity, scale_mult=scale_mult)
    gd = torch.randn(1, H, W, generator=torch.Generator().manual_seed(5)) / (H * W) * 100 if dgrad else None
    c1, r1, i1, g1 = _oracle_run(sc, H, W, aa, gd, oracle_backend)
    c2, r2, i2, g2 = _dense_run(sc, H, W, aa, gd)
    assert torch.equal(r1, r2)
    assert _rel(c1, c2) < 2e-5
    assert _rel(i1, i2) < 2e-5
    for k in ["means3D", "means2D", "opacities", "colors"]:
        assert _rel(g1[k], g2[k]) < 1e-4, k
    tol_cov = 5e-3 if aa else 1e-4  # see module docstring
    assert _rel(g1["scales"], g2["scales"]) < tol_cov
    assert _rel(g1["rotations"], g2["rotations"]) < tol_cov
    # grad_viewmatrix: reference assembly = mean path (autograd through vm with T held as a separate leaf)
    # + NCD2Screen-row-scaled dL_dT sum
    ncd = torch.tensor([W / 2.0, H / 2.0, 1.0])
    expect = g2["viewmatrix"].clone()
    expect[:3, :2] += ncd[:, None] * g2["T"].t()
    tol_vm = 5e-3 if aa else 2e-4
    assert _rel(g1["viewmatrix"], expect) < tol_vm


@pytest.mark.parametrize("mod", [0.5, 1.7])
def test_scale_modifier_and_the_missing_factor_in_dL_dscale(mod, oracle_backend):
    """raster_settings.scale_modifier: the covariance is built from mod * scale (forward.cu:117-151), and the reference returns
    dL/d(mod * scale) AS dL/dscale — no factor mod (backward.cu:331-383: `dL_dscale->x = dot(Rt[0], dL_dMt[0])`, M built with
    the modified scales). Against the dense autograd renderer fed mod * scale: same image, every other gradient equal, and the
    oracle's dL/dscale equal to autograd's dL/dscale DIVIDED by mod."""
    from eogs2_amd import GaussianRasterizer

    P, H, W = 300, 48, 64
    sc = make_scene(P, H, W, seed=21, opacity="trained", scale_mult=2.5)
    leaves = {k: sc[k].clone().requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "colors"]}
    rs = settings_for(sc, H, W)._replace(scale_modifier=mod)
    color, radii, invd = GaussianRasterizer(rs)(leaves["means3D"], torch.zeros(P, 3), leaves["opacities"], colors_precomp=leaves["colors"],
                                                scales=leaves["scales"], rotations=leaves["rotations"])
    (color * sc["dL_dcolor"]).sum().backward()
    g1 = {k: v.grad for k, v in leaves.items()}
    dl = {k: sc[k].clone().requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "colors"]}
    c2, r2, _ = render_dense(dl["means3D"], dl["opacities"], dl["colors"], sc["bg"], sc["viewmatrix"], H, W, scales=dl["scales"] * mod,
                             rotations=dl["rotations"], antialiasing=False, block=32)
    (c2 * sc["dL_dcolor"]).sum().backward()
    assert torch.equal(radii, r2) and _rel(color.detach(), c2.detach()) < 2e-5
    for k in ["means3D", "opacities", "colors", "rotations"]:
        assert _rel(g1[k], dl[k].grad) < 1e-4, k
    assert _rel(g1["scales"], dl["scales"].grad / mod) < 1e-4  # autograd's chain rule has the factor the reference leaves out
    assert _rel(g1["scales"], dl["scales"].grad) > 0.2


def test_oracle_altitude_trap(oracle_backend):
    from eogs2_amd import GaussianRasterizer, RastError

    H = W = 32
    sc = make_scene(50, H, W, seed=0)
    sc["means3D"][7, 2] = 1.0  # altitude 350 > 200
    rs = settings_for(sc, H, W)
    with pytest.raises(RastError, match="too high"):
        GaussianRasterizer(rs)(sc["means3D"], torch.zeros(50, 3), sc["opacities"], colors_precomp=sc["colors"],
                               scales=sc["scales"], rotations=sc["rotations"])


def test_config1_plumbing_1k_128(oracle_backend):
    """BASELINE.json configs[0] / SURVEY.md 8d config 1: 1 k synthetic Gaussians, 128 x 128, seed 0, CPU only — the
    pure-PyTorch alpha-blend forward equals the restatement reached through the drop-in API
    (`diff_gaussian_rasterization.GaussianRasterizer`, here routed to the checker library) to 1e-5 absolute."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    P, H, W = 1000, 128, 128
    sc = make_scene(P, H, W, seed=0)
    rs = settings_for(sc, H, W)
    assert isinstance(rs, GaussianRasterizationSettings)
    with torch.no_grad():
        color, radii, invd = GaussianRasterizer(rs)(sc["means3D"], torch.zeros(P, 3), sc["opacities"],
                                                    colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
        c2, r2, i2 = render_dense(sc["means3D"], sc["opacities"], sc["colors"], sc["bg"], sc["viewmatrix"], H, W,
                                  scales=sc["scales"], rotations=sc["rotations"])
    assert torch.equal(radii, r2) and int((radii > 0).sum()) == P
    assert float((color - c2).abs().max()) <= 1e-5
    assert float((invd - i2).abs().max()) <= 1e-5
    assert float((color[:3] - sc["bg"][:3, None, None]).abs().max()) > 1e-3  # something was blended


def test_threshold_nudge_moves_exactly_the_near_threshold_decision(oracle_backend):
    """eogs_oracle_threshold_nudge (the causal flip attribution of tests/parity_cases.py): a Gaussian centred on pixel
    (8, 8) with alpha = (1/255)(1 + 4 ulp) there is blended by the reference's arithmetic and with the thresholds moved
    down, skipped with them moved up by their (16 + 8 M)-ulp margin (M: the magnitude of the exponent's terms); no other pixel changes, a per-pixel map moves
    only its pixel, and sign 0 is the reference's arithmetic bit for bit."""
    import sys
    import os

    import numpy as np

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from parity_cases import nudged_run, oracle_run

    H = W = 16
    op = np.float32(1.0 / 255.0) * np.float32(1.0 + 4 * 2.0 ** -23)
    vm = np.eye(4, dtype=np.float32)
    case = dict(means3D=np.array([[1 / 16, 1 / 16, 10.0]], np.float32), scales=np.full((1, 3), 0.05, np.float32),
                rotations=np.array([[1, 0, 0, 0]], np.float32), opacities=np.array([[op]], np.float32),
                colors=np.array([[0.9, 0.1, 0.5, 10.0, 1.0]], np.float32), bg=np.zeros(5, np.float32), viewmatrix=vm,
                dL_dcolor=np.ones((5, H, W), np.float32), H=H, W=W, antialiasing=False)
    base, down, up = oracle_run(case), nudged_run(case, uniform=-1), nudged_run(case, uniform=1)
    assert base["out_color"][0, 8, 8] > 0 and float(base["g_opacities"].sum()) != 0  # blended at its centre pixel
    for k in base:
        assert np.array_equal(base[k], down[k]), k
        assert np.array_equal(base[k], nudged_run(case, uniform=0)[k]), k
    assert up["out_color"][0, 8, 8] == 0 and float(np.abs(up["g_opacities"]).sum()) == 0  # skipped: forward AND backward
    diff = np.abs(up["out_color"] - base["out_color"]).max(0)
    assert int((diff > 0).sum()) == 1 and diff[8, 8] > 0
    m = np.zeros((H, W), np.int8)
    m[3, 3] = 1  # a pixel without a near pair: nothing moves
    assert all(np.array_equal(base[k], v) for k, v in nudged_run(case, sign_map=m).items())
    m[8, 8] = 1
    assert all(np.array_equal(up[k], v) for k, v in nudged_run(case, sign_map=m).items())
    # the same two uniform runs taken in worker processes (prefetch_nudges: what the 1 M-Gaussian comparison does to halve its
    # wall time) are the in-process ones, and an unused prefetch is dropped by compare()
    from parity_cases import _PREFETCH, _take_prefetched, prefetch_nudges

    prefetch_nudges(case)
    pre = _take_prefetched(case)
    assert id(case) not in _PREFETCH
    for k in base:
        assert np.array_equal(pre[-1][k], down[k]) and np.array_equal(pre[1][k], up[k]), k


def test_threshold_margin_follows_the_magnitude_of_the_exponents_terms(oracle_backend):
    """The alpha-threshold margin of the moved-threshold runs is (16 + 8 M) ulp with M = |a| dx^2 / 2 + |c| dy^2 / 2 + |b dx dy|, not
    (16 + 8 |power|): sweep seed 9241 holds a thin rotated Gaussian whose exponent at pixel (195, 69) is -1720.0 - 1766.7 + 3485.8 =
    -0.94 — rounding of terms of that size moves alpha by ~M ulp — and whose alpha sits 1563 ulp below 1/255 (|power|-margin: 24 ulp).
    The oracle skips it; with the threshold moved DOWN it blends it (all five channels of that pixel move: what the HIP path, which
    associates the exponent differently, renders there); seed 9376's pixel (23, 6) is the mirror case, 3510 ulp above the threshold."""
    import os
    import sys

    import numpy as np

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from parity_cases import nudged_run, oracle_run, sweep_case

    for seed, (x, y), moving, hip in ((9241, (195, 69), -1, (0.8845484, 0.2205877, 0.6583073, -13.6817, 0.111772)),
                                      (9376, (23, 6), +1, (0.4699688, None, None, 22.25215, 0.9587148))):
        case, _ = sweep_case(seed)
        base, moved, other = oracle_run(case), nudged_run(case, uniform=moving), nudged_run(case, uniform=-moving)
        assert np.array_equal(base["out_color"][:, y, x], other["out_color"][:, y, x])  # the other direction leaves the pixel alone
        assert not np.array_equal(base["out_color"][:, y, x], moved["out_color"][:, y, x])
        for ch, v in enumerate(hip):  # the values the HIP path renders at that pixel (profiles/r06_sweeps.txt, tools/pixel_probe.py)
            if v is not None:
                assert abs(float(moved["out_color"][ch, y, x]) - v) <= 1e-5 * max(1.0, abs(v)), (seed, ch, moved["out_color"][ch, y, x], v)


@pytest.mark.parametrize("P,H,W,seed,opacity,scale_mult", [(1500, 32, 32, 3, 0.6, 10.0), (2000, 40, 48, 7, 0.95, 15.0)])
def test_back_to_front_recursion_is_the_accurate_form_for_image_sized_gaussians(P, H, W, seed, opacity, scale_mult, oracle_backend):
    """Why every backward kernel walks back to front (csrc/render.hip): against the independent dense autograd
    renderer the reference's recursion (backward.cu:586-620) is an order of magnitude closer than the front-to-back form the
    kernels of rounds 1-5 used (sum behind a Gaussian = rendered total minus running prefix; eogs_oracle_suffix_by_subtraction(1)
    evaluates it inside the oracle) — and so is the form the HIP kernels evaluate it in, the recursion projected on the pixel's
    upstream gradient (one number per pixel instead of five; eogs_oracle_suffix_by_subtraction(3)). Ordinary footprints show
    no difference."""
    sc = make_scene(P, H, W, seed=seed, opacity=opacity, scale_mult=scale_mult)
    _, _, _, gd = _dense_run(sc, H, W, False, None)
    err = {}
    for mode in (0, 1, 3):
        oracle_backend.cdll.eogs_oracle_suffix_by_subtraction(mode)
        try:
            _, _, _, g = _oracle_run(sc, H, W, False, None, oracle_backend)
        finally:
            oracle_backend.cdll.eogs_oracle_suffix_by_subtraction(0)
        err[mode] = {k: _rel(g[k], gd[k]) for k in ("opacities", "means2D", "means3D", "scales")}
    for k in err[0]:
        assert err[0][k] < 2e-5, (k, err)          # the reference's form: at the level of fp32 rounding
        assert err[3][k] < 2e-5, (k, err)          # ... and so is its projection on the upstream gradient (the HIP kernels' form)
        assert err[1][k] > 2.0 * err[0][k], (k, err)  # the front-to-back form: visibly worse on every gradient
    assert err[1]["opacities"] > 2e-5, err
    # an ordinary scene (footprints of a few tiles): both forms agree with the dense renderer alike
    sc = make_scene(400, 64, 64, seed=0, opacity="trained", scale_mult=1.0)
    _, _, _, gd = _dense_run(sc, 64, 64, False, None)
    for mode in (0, 1):
        oracle_backend.cdll.eogs_oracle_suffix_by_subtraction(mode)
        try:
            _, _, _, g = _oracle_run(sc, 64, 64, False, None, oracle_backend)
        finally:
            oracle_backend.cdll.eogs_oracle_suffix_by_subtraction(0)
        assert all(_rel(g[k], gd[k]) < 2e-5 for k in ("opacities", "means2D", "means3D")), mode
