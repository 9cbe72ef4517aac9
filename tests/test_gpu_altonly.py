"""GPU (MI355X): the altitude-only render (EOGS_FLAG_ALT_ONLY, include/eogs_rast.h; eogs2_amd.fused.rasterize_raw(...,
altitude_only=True)) — the reference's sun-camera render is consumed through its altitude channel alone with the shipped
configuration (train_pan.py:305-324, gs_config/train.yaml:123), so forward and backward blend one channel instead of five.
Checked against (1) the full five-channel HIP render: its channel 3, and its backward with zero upstream gradient on the other
channels; (2) the CPU oracle's restatement of the raw-parameter front end, same upstream gradient; (3) through
render_resample_virtual_camera, the deferred-count path and a replayed HIP graph."""
import types

import pytest
import torch

from util import assert_close, raw_params_from_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from eogs2_amd import _lib

    assert _lib.get().backend == "hip-gfx950"
    return torch.device("cuda:0")


def _run(raw, alt_affine, scene, H, W, aa, altitude_only, g_alt):
    """fwd + bwd through rasterize_raw with the upstream gradient `g_alt` [H, W] on the altitude channel only."""
    from eogs2_amd.fused import rasterize_raw
    from eogs2_amd.synthetic import settings_for

    leaves = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    vm = scene["viewmatrix"].clone().requires_grad_(True)
    rs = settings_for(dict(scene, viewmatrix=vm), H, W, antialiasing=aa)._replace(projmatrix=vm.detach())
    P = raw["xyz"].shape[0]
    m2 = torch.zeros(P, 3, device=vm.device, requires_grad=True)
    color, radii, _ = rasterize_raw(leaves["xyz"], m2, leaves["f_dc"], leaves["opacity_logit"], leaves["log_scaling"],
                                    leaves["raw_rotation"], alt_affine, rs, altitude_only=altitude_only)
    assert tuple(color.shape) == ((1, H, W) if altitude_only else (5, H, W))
    alt_img = color[0] if altitude_only else color[3]
    (alt_img * g_alt).sum().backward()
    out = dict(altitude=alt_img.detach(), out_radii=radii, g_means2D=m2.grad, g_viewmatrix=vm.grad)
    out.update({"g_" + k: v.grad for k, v in leaves.items()})
    out["_token"] = int(color.grad_fn.num_rendered)
    return out


CASES = [  # P, H, W, seed, opacity, scale_mult, antialiasing
    (30000, 200, 264, 61, "trained", 1.5, False),
    (8000, 131, 97, 62, "init", 2.5, True),       # partial tiles on both edges
    (3000, 64, 64, 63, 0.7, 8.0, False),          # long lists, early termination
    (400, 200, 168, 64, "trained", 14.0, False),  # image-sized footprints: the full render takes the back-to-front backward
    (1 << 20, 2048, 2048, 65, "init", 1.0, False),  # the sun camera of the bench's iteration at its own size (row-span listing)
]


@pytest.mark.parametrize("P,H,W,seed,opacity,scale_mult,aa", CASES)
def test_altitude_only_equals_channel_3_of_the_full_render(dev, P, H, W, seed, opacity, scale_mult, aa):
    from eogs2_amd import _lib
    from eogs2_amd.synthetic import make_scene

    scene = make_scene(P, H, W, seed=seed, opacity=opacity, scale_mult=scale_mult, device=dev)
    raw, alt = raw_params_from_scene(scene, seed=seed)
    g_alt = torch.randn(H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(seed)) / (H * W)
    a = _run(raw, alt, scene, H, W, aa, True, g_alt)
    b = _run(raw, alt, scene, H, W, aa, False, g_alt)
    assert (a["_token"] >> 59) & 1 == 1 and (b["_token"] >> 59) & 1 == 0
    assert _lib.get().path_info(P, a["_token"]) == (8, 2, 6)  # per-tile lists, the quad kernels' one-channel variants
    assert torch.equal(a["out_radii"], b["out_radii"])
    assert torch.equal(a["altitude"], b["altitude"]), "forward: the same products in the same order"
    assert float(a["g_f_dc"].abs().max()) == 0.0 and float(b["g_f_dc"].abs().max()) == 0.0
    # The same recursion on both sides (every backward kernel walks back to front): equal to the rounding of two orders of
    # summation — the full render of large footprints reads block lists and sums a tile's survivors in rounds of eight, the
    # altitude-only one always walks per-tile lists with quad sub-lists — at every footprint, image-sized Gaussians included
    # (rounds 4-5 accepted 1e-3 there: the full render then ran another FORMULATION of dL/dalpha than the altitude-only one,
    # which had no such fallback).
    for k in a:
        if k.startswith("g_") and k != "g_viewmatrix":
            # (2e-5 everywhere but on the rotations of image-sized footprints, whose covariance backward amplifies the two
            # summation orders' rounding: those are held to the parity tolerance itself)
            assert_close(a[k], b[k], f"alt-only vs full:{k}", rtol=1e-4 if (k == "g_raw_rotation" and scale_mult >= 8) else 2e-5, allow_flips=False)
    scale = float((scene["means3D"].abs().t() @ b["g_means2D"].abs()).max())
    assert float((a["g_viewmatrix"] - b["g_viewmatrix"]).abs().max()) <= 1e-5 * scale


@pytest.mark.parametrize("P,H,W,seed,opacity,scale_mult,aa", CASES + [(3000, 96, 80, 32, "trained", 2.0, True)])
def test_altitude_only_matches_oracle(dev, monkeypatch, P, H, W, seed, opacity, scale_mult, aa):
    """Every case above against the ORACLE's restatement of the raw-parameter front end (C, double-precision chain): its full
    render with the same upstream gradient on channel 3 and zero on the other channels — image-sized footprints and the sun
    camera's own size (1 M Gaussians at 2048^2) included (round 5 compared only the last, small case with the oracle and the
    others with the HIP path's own full render)."""
    import time

    import oracle
    from util import run_raw

    from eogs2_amd import _lib
    from eogs2_amd.synthetic import make_scene

    scene = make_scene(P, H, W, seed=seed, opacity=opacity, scale_mult=scale_mult)
    raw, alt = raw_params_from_scene(scene, seed=seed)
    g_alt = torch.randn(H, W, generator=torch.Generator().manual_seed(5)) / (H * W)
    to = lambda d: {k: v.to(dev) for k, v in d.items()}
    got = _run(to(raw), alt.to(dev), to(scene), H, W, aa, True, g_alt.to(dev))
    dL = torch.zeros(5, H, W)
    dL[3] = g_alt
    oabi = oracle.abi()
    monkeypatch.setattr(_lib, "get", lambda: oabi)
    t0 = time.perf_counter()
    ref = run_raw(raw, alt, dict(scene, dL_dcolor=dL), H, W, aa, fused=True)
    monkeypatch.undo()
    print(f"oracle, full render of {P} Gaussians at {H}x{W}: {time.perf_counter() - t0:.1f} s")
    # (raw-parameter mode: the scale is exp(log-scale), evaluated by libm on one side and the GPU's expf on the other — an ulp
    # apart now and then, and radius = ceil(3 sqrt(lambda)) then steps by one for a Gaussian in a few hundred thousand)
    nrad = int((got["out_radii"].cpu() != ref["out_radii"]).sum())
    assert nrad <= P // 100_000 and int((got["out_radii"].cpu() - ref["out_radii"]).abs().max()) <= 1, nrad
    # (flip_floor: pixels / Gaussians a blend decision within an ulp of its threshold may move — v_exp_f32 against libm's expf —
    # each bounded by FLIP_RTOL; the non-raw suites attribute such elements causally, tests/parity_cases.py)
    floor = 4 + P // 30000
    # ([1, H, W]: an image — a bare [H, W] tensor would be read as H per-Gaussian rows, one moved pixel spoiling a whole row)
    # (flip_rtol 2e-2: a moved pair changes its pixel by alpha T c <= 0.01 c at these opacities, and c — a Gaussian's altitude —
    # can exceed the blended image's maximum, the scale; two moved pairs may share a pixel)
    assert_close(got["altitude"][None], ref["out_color"][3:4], "altitude vs oracle channel 3", flip_floor=floor, flip_rtol=2e-2)
    # (gradients: a moved pair changes the gradient rows of the Gaussians at its pixel by that pixel's term — with a white-noise
    # upstream gradient up to a few per cent of a column's largest entry; counted per Gaussian, bounded at 5e-2)
    for k in ("g_xyz", "g_opacity_logit", "g_log_scaling", "g_raw_rotation", "g_means2D"):
        assert_close(got[k], ref[k], f"alt-only vs oracle:{k}", flip_floor=floor, flip_rtol=5e-2)
    assert float(ref["g_f_dc"].abs().max()) == 0.0 and float(got["g_f_dc"].abs().max()) == 0.0


def test_nothing_visible_and_errors(dev):
    from eogs2_amd.fused import rasterize_raw
    from eogs2_amd.synthetic import make_scene, settings_for

    H, W, P = 64, 96, 500
    scene = make_scene(P, H, W, seed=3, opacity="trained", device=dev)
    scene["means3D"] = scene["means3D"] + torch.tensor([50.0, 50.0, 0.0], device=dev)  # everything outside the image
    raw, alt = raw_params_from_scene(scene)
    rs = settings_for(scene, H, W)
    leaves = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    color, radii, invd = rasterize_raw(leaves["xyz"], m2, leaves["f_dc"], leaves["opacity_logit"], leaves["log_scaling"],
                                       leaves["raw_rotation"], alt, rs, altitude_only=True)
    assert invd is None  # (no inverse-depth image exists for such a render: nothing uninitialised to read)
    assert torch.equal(color, scene["bg"][3].expand(1, H, W))  # the background's altitude everywhere
    color.sum().backward()
    assert all(float(v.grad.abs().max()) == 0.0 for v in leaves.values())


def test_resample_entry_point_altitude_only(dev):
    """render_resample_virtual_camera(..., altitude_only=True): altitude sample, coordinates and every gradient equal the
    full call's; rgb_sample is None."""
    from test_fused_cpu import _Cam, _Model

    from eogs2_amd.resample import render_resample_virtual_camera
    from eogs2_amd.synthetic import make_scene

    H, W, P = 96, 128, 6000
    scene = make_scene(P, 2 * H, 2 * W, seed=7, opacity="trained", device=dev, scale_mult=2.0)
    raw, _ = raw_params_from_scene(scene)
    pipe = types.SimpleNamespace(debug=False, antialiasing=False, compute_cov3D_python=False, require_radii=False)
    U, V = torch.meshgrid(torch.linspace(-1, 1, W, device=dev), torch.linspace(-1, 1, H, device=dev), indexing="xy")
    M = torch.eye(3, device=dev)
    M[0, 0], M[1, 1], M[0, 2] = 1.1, 0.95, 0.002  # part of the grid leaves the virtual view
    w_alt = torch.randn(H, W, device=dev)
    res = {}
    for alt_only in (True, False):
        cam, pc = _Cam(scene["viewmatrix"], 2 * H, 2 * W), _Model(raw)
        cam.last_row = cam.last_row.detach().to(dev)
        cam.camera_center = cam.camera_center.to(dev)
        true_alt = (0.05 * torch.randn(H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(1))).requires_grad_(True)
        uva = torch.stack((U, V, true_alt), dim=-1)
        rgb, a, uv = render_resample_virtual_camera(cam, M, uva, pc, pipe, scene["bg"], altitude_only=alt_only)
        assert (rgb is None) == alt_only
        (a * w_alt).sum().backward()
        res[alt_only] = dict(alt=a.detach(), uv=uv.detach(), g_true_alt=true_alt.grad, **{k: v.grad for k, v in pc.params().items()})
    assert int((res[True]["alt"] == -100).sum()) > 0
    for k in res[True]:
        assert_close(res[True][k], res[False][k], f"resample alt-only vs full:{k}", rtol=2e-6, allow_flips=False)


def test_altitude_only_deferred_counts_and_graph(dev):
    """The token's altitude bit survives the paths that build tokens without flags: the deferred count readback (capacity
    tokens) and a replayed HIP graph (mirrored counts); results bit-identical to the plain eager call."""
    from eogs2_amd import rasterizer
    from eogs2_amd.graph import GraphedStep
    from eogs2_amd.synthetic import make_scene

    H, W, P = 128, 160, 20000
    scene = make_scene(P, H, W, seed=9, opacity="trained", device=dev)
    raw, alt = raw_params_from_scene(scene)
    g_alt = torch.randn(H, W, device=dev) / (H * W)
    want = _run(raw, alt, scene, H, W, False, True, g_alt)
    old = rasterizer.set_speculation(True, forget=True)
    try:
        _run(raw, alt, scene, H, W, False, True, g_alt)
        rasterizer.speculation_stats(reset=True)
        again = _run(raw, alt, scene, H, W, False, True, g_alt)
        assert rasterizer.speculation_stats()["hit"] == 1
    finally:
        rasterizer.set_speculation(old)
    for k in want:
        if not k.startswith("_"):
            assert torch.equal(want[k], again[k]), k

    def fn():
        out = _run(raw, alt, scene, H, W, False, True, g_alt)
        return tuple(out[k] for k in sorted(out) if not k.startswith("_"))

    g = GraphedStep(fn, warmup=1)
    got = [t.clone() for t in g()]
    for t, k in zip(got, [k for k in sorted(want) if not k.startswith("_")]):
        assert torch.equal(t, want[k]), k


@pytest.mark.parametrize("P,H,W,seed,opacity,scale_mult,aa", [CASES[0], CASES[1], (120_000, 344, 392, 66, "init", 1.0, False)])
def test_render_without_the_inverse_depth_image_changes_no_bit(dev, P, H, W, seed, opacity, scale_mult, aa):
    """`rasterize_raw(..., invdepth=False)` (what eogs2_amd.render.render passes: the reference's render() drops the rasterizer's
    inverse-depth image, renderer.py:101,126) hands the C-ABI out_invdepth = NULL; the quad forward then leaves that multiply-add
    out of every evaluated pair. The colour image, the radii and every gradient are the same BITS as with the image rendered."""
    from eogs2_amd.fused import rasterize_raw
    from eogs2_amd.synthetic import make_scene, settings_for

    scene = make_scene(P, H, W, seed=seed, opacity=opacity, scale_mult=scale_mult, device=dev)
    raw, alt = raw_params_from_scene(scene, seed=seed)
    g = torch.randn(5, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(seed)) / (H * W)

    def run(invdepth):
        leaves = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
        vm = scene["viewmatrix"].clone().requires_grad_(True)
        rs = settings_for(dict(scene, viewmatrix=vm), H, W, antialiasing=aa)._replace(projmatrix=vm.detach())
        m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
        color, radii, invd = rasterize_raw(leaves["xyz"], m2, leaves["f_dc"], leaves["opacity_logit"], leaves["log_scaling"],
                                           leaves["raw_rotation"], alt, rs, invdepth=invdepth)
        assert (invd is None) == (not invdepth)
        (color * g).sum().backward()
        out = dict(color=color.detach(), radii=radii, g_means2D=m2.grad, g_viewmatrix=vm.grad)
        out.update({"g_" + k: v.grad for k, v in leaves.items()})
        return out

    a, b = run(True), run(False)
    assert float(a["color"].abs().max()) > 0.1 and float(a["g_xyz"].abs().max()) > 0
    for k in a:
        assert torch.equal(a[k], b[k]), k
