"""GPU (MI355X): the fused image chain (include/eogs_shade.h, eogs2_amd/shade.py) through the C-ABI against
(1) the vectors produced by the reference's own modules (tests/golden/shade_*.npz), (2) the float64 oracle
(oracle/shade_oracle.py) on odd sizes, (3) size-independent properties at 1024^2, and the reference-named entry points."""
import types

import numpy as np
import pytest
import torch

from oracle import shade_oracle as O
from shade_cases import close, expected_matrix_grad, fixtures, load, run_shade
from test_shade_oracle import check_mloss, run_mloss

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from eogs2_amd import _lib

    assert _lib.get().backend == "hip-gfx950"
    return torch.device("cuda:0")


@pytest.mark.parametrize("path", fixtures("cc_") + fixtures("exposure_") + fixtures("identity_"), ids=lambda p: p.split("shade_")[-1][:-4])
def test_render_pipeline_golden(dev, path):
    from eogs2_amd.shade import shade

    fx = load(path)
    got = run_shade(shade, fx, torch.float32, dev)
    for k in ("cc", "shaded", "g_raw"):
        close(got[k], fx[k], TOL, k)
    if "alt_diff" in fx:
        for k in ("shadow", "g_alt_diff", "g_inshadow"):
            close(got[k], fx[k], TOL, k)
    gM = expected_matrix_grad(fx)
    if gM is not None:
        close(got["g_M"], gM, TOL, "g_M")


@pytest.mark.parametrize("path", fixtures("mloss_"), ids=lambda p: p.split("shade_")[-1][:-4])
def test_masked_losses_golden(dev, path):
    from eogs2_amd.shade import randomcam_l, suncamera_l

    fx = load(path)
    check_mloss(run_mloss(suncamera_l, randomcam_l, fx, torch.float32, dev), fx, TOL)


@pytest.mark.parametrize("path", fixtures("tshadow_"), ids=lambda p: p.split("shade_")[-1][:-4])
def test_translucent_shadows_golden(dev, path):
    from eogs2_amd.shade import translucentshadows_l

    fx = load(path)
    a = torch.tensor(fx["a"], device=dev).requires_grad_(True)
    L = translucentshadows_l(a)
    (float(fx["upstream"]) * L).backward()
    assert abs(float(L) - float(fx["L"])) <= TOL * abs(float(fx["L"]))
    close(a.grad.cpu().numpy(), fx["g_a"], TOL, "g_a")


def _synthetic(H, W, seed, shadow=True):
    g = torch.Generator().manual_seed(seed)
    fx = dict(kind=np.array("exposure"), raw=torch.rand((3, H, W), generator=g).numpy(),
              exposure=(torch.eye(3, 4) + 0.3 * torch.randn((3, 4), generator=g))[None].numpy(),
              inshadow=(0.05 + 0.5 * torch.rand(3, generator=g)).numpy(),
              g_shaded=torch.randn((3, H, W), generator=g).numpy(), g_cc=torch.randn((3, H, W), generator=g).numpy())
    if shadow:
        fx.update(alt_diff=(3 * torch.randn((H, W), generator=g)).numpy(), g_shadow=torch.randn((H, W), generator=g).numpy())
    return fx


@pytest.mark.parametrize("H,W,shadow", [(1, 1, True), (7, 129, True), (33, 65, False), (255, 257, True), (300, 1000, True)])
def test_render_pipeline_vs_oracle_odd_sizes(dev, H, W, shadow):
    from eogs2_amd.shade import shade

    fx = _synthetic(H, W, 100 + H, shadow)
    got = run_shade(shade, fx, torch.float32, dev)
    ref = run_shade(O.render_pipeline, fx, torch.float64, "cpu")
    for k in got:
        # parameter gradients are sums of H*W signed terms: compare against the magnitude they are reduced from
        tol = TOL if k not in ("g_M", "g_inshadow") else 2e-5 * max(1.0, (H * W) ** 0.5 / 30)
        close(got[k], ref[k], tol, k)


@pytest.mark.parametrize("mode", ["sun", "random"])
@pytest.mark.parametrize("H,W", [(1, 3), (130, 67), (512, 700)])
def test_masked_losses_vs_oracle(dev, mode, H, W):
    from eogs2_amd.shade import randomcam_l, suncamera_l

    g = torch.Generator().manual_seed(H * 7 + W)
    fx = dict(mode=np.array(mode), rgb_a=torch.rand((3, H, W), generator=g).numpy(), rgb_b=torch.rand((3, H, W), generator=g).numpy(),
              alt_diff=(0.3 * torch.randn((H, W), generator=g)).numpy(), uv=(1.2 * (2 * torch.rand((H, W, 2), generator=g) - 1)).numpy(),
              upstream=np.array([1.5, -0.4], np.float32))
    ref = run_mloss(O.suncamera_l, O.randomcam_l, fx, torch.float64, "cpu")
    got = run_mloss(suncamera_l, randomcam_l, fx, torch.float32, dev)
    check_mloss(got, {**fx, **ref}, TOL)


def test_full_size_properties(dev):
    """1024^2: determinism, linearity of backward in the upstream gradients, identity pipeline, shadow bounds."""
    from eogs2_amd.shade import render_pipeline, shade, translucentshadows_l

    H = W = 1024
    fx = _synthetic(H, W, 7)
    a = run_shade(shade, fx, torch.float32, dev)
    b = run_shade(shade, fx, torch.float32, dev)
    for k in a:
        assert np.array_equal(a[k], b[k]), f"{k} not bitwise reproducible"
    fx2 = dict(fx)
    for k in ("g_shaded", "g_cc", "g_shadow"):
        fx2[k] = 2.0 * fx[k]
    c = run_shade(shade, fx2, torch.float32, dev)
    for k in ("g_raw", "g_alt_diff", "g_M", "g_inshadow"):
        assert np.array_equal(c[k], 2.0 * a[k]), f"backward not linear in the upstream gradient: {k}"
    assert a["shadow"].min() > 0.0 and a["shadow"].max() <= 1.0
    # a camera with neither colour correction nor shadows returns its input
    raw = torch.tensor(fx["raw"], device=dev)
    out = render_pipeline(types.SimpleNamespace(use_cc=False, use_exposure=False, use_shadow=False), raw, None)
    assert torch.equal(out["final"], raw) and out["shadowmap"] is None
    # translucent-shadow loss of a constant map inside the clip range is the binary entropy
    L = translucentshadows_l(torch.full((H, W), 0.25, device=dev))
    assert abs(float(L) - 0.8112781244591328) < 2e-6


def test_render_pipeline_entry_point_and_parameter_gradients(dev):
    """The reference-named entry with a Conv2d colour correction: gradients reach weight, bias and the in-shadow tint
    exactly as through the reference's op sequence (run here in fp32 on the GPU)."""
    from eogs2_amd.shade import render_pipeline

    torch.manual_seed(3)
    H, W = 96, 160
    raw = torch.rand((3, H, W), device=dev)
    alt = torch.randn((H, W), device=dev) * 2

    def make_cam():
        cam = types.SimpleNamespace(use_cc=True, use_exposure=False, use_shadow=True)
        cam.color_correction = torch.nn.Conv2d(3, 3, 1, bias=True).to(dev)
        with torch.no_grad():
            cam.color_correction.weight.copy_(torch.tensor([[1.1, 0.1, -0.2], [0.0, 0.9, 0.2], [0.3, -0.1, 1.0]], device=dev).reshape(3, 3, 1, 1))
            cam.color_correction.bias.copy_(torch.tensor([0.02, -0.03, 0.05], device=dev))
        cam.inshadow_color_correction = torch.nn.Parameter(torch.tensor([0.05, 0.2, 0.4], device=dev).reshape(3, 1, 1))
        return cam

    w = torch.randn((3, H, W), device=dev)
    res = []
    for fused in (True, False):
        cam = make_cam()
        r, a = raw.clone().requires_grad_(True), alt.clone().requires_grad_(True)
        if fused:
            out = render_pipeline(cam, r, a)
            shaded, shadow = out["final"], out["shadowmap"]
        else:  # affine_cameras.py:311-334 as PyTorch ops
            cc = cam.color_correction(r.unsqueeze(0))
            shadow = torch.exp(0.4 * a.clip(max=0.0))
            shaded = (shadow * cc + (1 - shadow) * cam.inshadow_color_correction * cc).squeeze(0)
        ((shaded * w).sum() + shadow.sum()).backward()
        res.append([shaded.detach(), r.grad, a.grad, cam.color_correction.weight.grad.reshape(3, 3),
                    cam.color_correction.bias.grad, cam.inshadow_color_correction.grad.reshape(3)])
    for x, y, name in zip(res[0], res[1], ["shaded", "g_raw", "g_alt", "g_weight", "g_bias", "g_inshadow"]):
        close(x.cpu().numpy(), y.cpu().numpy(), 1e-4 if name.startswith("g_w") or name in ("g_bias", "g_inshadow") else TOL, name)
