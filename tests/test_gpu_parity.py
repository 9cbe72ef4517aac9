"""GPU (MI355X): the HIP path, called through the C-ABI via the drop-in Python API, against
  (1) the committed golden vectors (reference wrapper over the oracle),
  (2) the CPU oracle on fresh seeded inputs,
  (3) size-independent properties at the BASELINE size (1M Gaussians / 1024^2).
Tolerance: north_star's 1e-4 rel, see tests/util.py.
"""
import math

import numpy as np
import pytest
import torch

from util import GOLDEN, assert_close, load_golden, run_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from eogs2_amd import _lib

    assert _lib.get().backend == "hip-gfx950"  # the native library is what runs
    return torch.device("cuda:0")


def test_wave64_primitives_selftest(dev):
    import ctypes

    from eogs2_amd import _lib

    abi = _lib.get()
    scratch = torch.zeros(4, dtype=torch.int32, device=dev)
    failed = ctypes.c_uint(99)
    abi.check(abi.selftest(ctypes.c_void_p(scratch.data_ptr()), ctypes.byref(failed),
                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert failed.value == 0, f"wave64 primitive self-test failed: mask {failed.value:#x}"


def _compare(out, ref, name, means3D=None):
    assert np.array_equal(out["out_radii"].cpu().numpy(), np.asarray(ref["out_radii"])), f"{name}: radii differ"
    for k, v in out.items():
        if k == "out_radii":
            continue
        r = torch.as_tensor(np.asarray(ref[k]))
        if k == "g_viewmatrix":
            # a cancelling sum over all Gaussians of terms that are each within 1e-4: the meaningful scale is the
            # sum of magnitudes |means3D|^T @ |dL_dmeans2D| (and sum |dL_dmeans2D| for the last row), not |sum|
            g2 = torch.as_tensor(np.asarray(ref["g_means2D"])).abs().double()
            m = torch.as_tensor(np.asarray(means3D)).abs().double()
            scale = max(float((m.t() @ g2).max()), float(g2.sum(0).max()), float(r.abs().max()))
            err = float((v.cpu().double() - r.double()).abs().max()) / scale
            assert err <= 1e-4, f"{name}:{k}: {err:.3e} of the magnitude sum"
            continue
        from util import GRAD_RTOL, RTOL

        assert_close(v, r, f"{name}:{k}", rtol=GRAD_RTOL.get(name, RTOL) if k.startswith("g_") else RTOL)


@pytest.mark.parametrize("name", GOLDEN)
def test_hip_matches_golden(name, dev):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    case = load_golden(name)
    out = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    _compare(out, case, name, case["means3D"])


SEEDED = [
    # P, H, W, seed, opacity, scale_mult, aa, depth_grad
    (5000, 160, 208, 10, "init", 2.0, False, False),
    (5000, 160, 208, 11, "trained", 2.0, True, True),
    (3000, 64, 64, 12, 0.7, 8.0, False, False),      # long lists (>256/tile), early termination
    (20000, 256, 256, 13, "trained", 1.0, False, False),
    (777, 33, 47, 14, "trained", 4.0, False, True),    # ragged image, ragged P
]


@pytest.mark.parametrize("P,H,W,seed,opacity,scale_mult,aa,dgrad", SEEDED)
def test_hip_matches_oracle_seeded(P, H, W, seed, opacity, scale_mult, aa, dgrad, dev, monkeypatch):
    import oracle
    from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
    from eogs2_amd.synthetic import make_scene

    sc = make_scene(P, H, W, seed=seed, opacity=opacity, scale_mult=scale_mult)
    case = {k: v.numpy() for k, v in sc.items()}
    case.update(H=H, W=W, antialiasing=aa)
    if dgrad:
        case["dL_dinvdepth"] = (torch.randn(1, H, W, generator=torch.Generator().manual_seed(seed)) / (H * W) * 100).numpy()
    got = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    hip = _lib.get()
    monkeypatch.setattr(_lib, "get", lambda: oracle.abi())  # checker: same wrapper over the CPU oracle
    ref = run_case(case, torch.device("cpu"), GaussianRasterizer, GaussianRasterizationSettings)
    monkeypatch.setattr(_lib, "get", lambda: hip)
    _compare(got, {k: v.cpu().numpy() for k, v in ref.items()}, f"seed{seed}", case["means3D"])


def test_list_order_matches_reference_sort(dev):
    """The (tile, depth, index) order of the binned list equals a stable 64-bit-key sort (bit-exact integer work)."""
    import ctypes

    from eogs2_amd import _lib
    from eogs2_amd.synthetic import make_scene

    abi = _lib.get()
    P, H, W = 30000, 320, 272
    sc = make_scene(P, H, W, seed=3, opacity="trained", scale_mult=1.5, device=dev)
    # duplicate depths on purpose: ties must resolve by Gaussian index
    sc["means3D"][::7, 2] = sc["means3D"][1::7, 2][: sc["means3D"][::7].shape[0]]
    n = ctypes.c_size_t()
    abi.check(abi.geom_bytes(P, ctypes.byref(n))); geom = torch.empty(n.value, dtype=torch.uint8, device=dev)
    abi.check(abi.image_bytes(H, W, ctypes.byref(n))); img = torch.empty(n.value, dtype=torch.uint8, device=dev)
    radii = torch.empty(P, dtype=torch.int32, device=dev)
    R = ctypes.c_int64()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    abi.check(abi.forward_prepare(P, H, W, p(sc["means3D"]), p(sc["scales"]), p(sc["rotations"]), None, p(sc["opacities"]),
                                  1.0, p(sc["viewmatrix"]), p(sc["viewmatrix"]), 0, p(radii), p(geom), geom.numel(),
                                  ctypes.byref(R), st))
    abi.check(abi.binning_bytes(P, H, W, R.value, ctypes.byref(n))); binning = torch.empty(n.value, dtype=torch.uint8, device=dev)
    color = torch.empty(5, H, W, device=dev)
    abi.check(abi.forward_render(P, H, W, R.value, p(sc["colors"]), p(sc["bg"]), 0, p(geom), geom.numel(), p(binning),
                                 binning.numel(), p(img), img.numel(), p(color), None, st))
    torch.cuda.synchronize()
    # reference order recomputed with torch from the oracle-equivalent per-Gaussian data
    from oracle.torch_dense import render_dense

    _, r2, _, aux = render_dense(sc["means3D"].cpu(), sc["opacities"].cpu(), sc["colors"].cpu(), sc["bg"].cpu(),
                                 sc["viewmatrix"].cpu(), H, W, scales=sc["scales"].cpu(), rotations=sc["rotations"].cpu(),
                                 want_aux=True, block=H + 16 - H % 16 if H % 16 else H)
    assert torch.equal(radii.cpu(), r2)
    assert R.value == aux["num_rendered"]
    # parse our binning workspace: point_list location is internal, so check through the public effect instead:
    # every pixel's blend order is by construction the list order; identical images on a scene with many depth
    # ties and overlapping Gaussians is the observable. (Bit-exact list comparison lives in the oracle test.)
    col2 = render_dense(sc["means3D"].cpu(), sc["opacities"].cpu(), sc["colors"].cpu(), sc["bg"].cpu(),
                        sc["viewmatrix"].cpu(), H, W, scales=sc["scales"].cpu(), rotations=sc["rotations"].cpu(), block=64)[0]
    assert_close(color, col2, "tie-order image")


def test_backward_is_deterministic(dev):
    """Atomic-free backward: two runs give bitwise identical gradients."""
    from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer
    from eogs2_amd.synthetic import make_scene

    P, H, W = 20000, 256, 256
    sc = make_scene(P, H, W, seed=21, opacity="trained", scale_mult=1.5)
    case = {k: v.numpy() for k, v in sc.items()}
    case.update(H=H, W=W, antialiasing=False)
    a = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    b = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    for k in a:
        if k == "g_viewmatrix":
            continue  # 18 floats reduced with atomics
        assert torch.equal(a[k], b[k]), k


def test_full_size_properties(dev):
    """BASELINE size (1M Gaussians / 1024^2): properties that need no CPU reference.
    - linearity of backward in dL/dcolor; gradient of an all-zero dL is exactly zero
    - background: out(bg + d) - out(bg) = T_final * d, with 0 <= T_final <= 1
    - invisible Gaussians (radii == 0) receive exactly zero gradient; num visible matches radii
    - colour-channel 4 (constant 1, bg 0) renders accumulated opacity 1 - T_final
    """
    from eogs2_amd import GaussianRasterizer
    from eogs2_amd.synthetic import make_scene, settings_for

    P, H, W = 1_000_000, 1024, 1024
    sc = make_scene(P, H, W, seed=0, opacity="init", device=dev)
    rs = settings_for(sc, H, W)

    def fwd_bwd(bg, dL):
        leaves = {k: sc[k].clone().requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "colors"]}
        color, radii, invd = GaussianRasterizer(rs._replace(bg=bg))(
            leaves["means3D"], torch.zeros(P, 3, device=dev), leaves["opacities"], colors_precomp=leaves["colors"],
            scales=leaves["scales"], rotations=leaves["rotations"])
        (color * dL).sum().backward()
        return color.detach(), radii, {k: v.grad for k, v in leaves.items()}

    dL = sc["dL_dcolor"]
    c0, radii, g1 = fwd_bwd(sc["bg"], dL)
    assert torch.isfinite(c0).all()
    for v in g1.values():
        assert torch.isfinite(v).all()
    # linearity
    _, _, g2 = fwd_bwd(sc["bg"], 2.0 * dL)
    for k in g1:
        assert_close(g2[k], 2.0 * g1[k], f"linearity:{k}", allow_flips=False)
    _, _, g0 = fwd_bwd(sc["bg"], torch.zeros_like(dL))
    for k in g0:
        assert float(g0[k].abs().max()) == 0.0, k
    # background
    d = torch.tensor([0.25, -0.5, 1.0, 3.0, 2.0], device=dev)
    c1, _, _ = fwd_bwd(sc["bg"] + d, dL)
    Tf = (c1 - c0)[4] / d[4]
    assert float(Tf.min()) >= -1e-6 and float(Tf.max()) <= 1 + 1e-6
    assert_close((c1 - c0), Tf[None] * d[:, None, None], "bg-linearity", rtol=1e-4, allow_flips=False)
    # accumulated opacity channel: bg[4] = 0 -> c0[4] = 1 - T_final
    assert_close(c0[4], 1.0 - Tf, "opacity-channel", rtol=1e-4, allow_flips=False)
    # invisible Gaussians
    inv = radii == 0
    for k, v in g1.items():
        assert float(v[inv].abs().sum()) == 0.0, k
