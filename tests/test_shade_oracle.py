"""CPU: the float64 restatement (oracle/shade_oracle.py) against the vectors the reference's own modules produced
(tests/golden/make_golden_shade.py): render pipeline, masked resample losses, translucent-shadow regulariser."""
import numpy as np
import pytest
import torch

from oracle import shade_oracle as O
from shade_cases import close, expected_matrix_grad, fixtures, load, run_shade

TOL = 2e-5  # the vectors are fp32 results of the reference


@pytest.mark.parametrize("path", fixtures("cc_") + fixtures("exposure_") + fixtures("identity_"), ids=lambda p: p.split("shade_")[-1][:-4])
def test_render_pipeline_matches_reference_vectors(path):
    fx = load(path)
    got = run_shade(O.render_pipeline, fx, torch.float64, "cpu")
    for k in ("cc", "shaded", "g_raw"):
        close(got[k], fx[k], TOL, k)
    if "alt_diff" in fx:
        for k in ("shadow", "g_alt_diff", "g_inshadow"):
            close(got[k], fx[k], TOL, k)
    gM = expected_matrix_grad(fx)
    if gM is not None:
        close(got["g_M"], gM, TOL, "g_M")


def run_mloss(fn_sun, fn_random, fx, dtype, dev):
    t = lambda a: torch.tensor(a, dtype=dtype, device=dev)
    a, b = t(fx["rgb_a"]).requires_grad_(True), t(fx["rgb_b"]).requires_grad_(True)
    alt, uv, up = t(fx["alt_diff"]).requires_grad_(True), t(fx["uv"]), t(fx["upstream"])
    L = fn_sun(a, b, alt, uv) if str(fx["mode"]) == "sun" else fn_random(alt, a, b, uv)
    (up[0] * L[0] + up[1] * L[1]).backward()
    n = lambda x, like: (torch.zeros_like(like) if x is None else x).detach().double().cpu().numpy()  # None: no path = zero
    return dict(L_alt=float(L[0]), L_rgb=float(L[1]), g_alt_diff=n(alt.grad, alt), g_rgb_a=n(a.grad, a), g_rgb_b=n(b.grad, b))


def check_mloss(got, fx, tol):
    assert abs(got["L_alt"] - float(fx["L_alt"])) <= tol * max(abs(float(fx["L_alt"])), 1e-30) + 1e-30
    assert abs(got["L_rgb"] - float(fx["L_rgb"])) <= tol * max(abs(float(fx["L_rgb"])), 1e-30) + 1e-30
    for k in ("g_alt_diff", "g_rgb_a", "g_rgb_b"):
        if np.abs(fx[k]).max() == 0:
            assert np.abs(got[k]).max() == 0, k
        else:
            close(got[k], fx[k], tol, k)


@pytest.mark.parametrize("path", fixtures("mloss_"), ids=lambda p: p.split("shade_")[-1][:-4])
def test_masked_losses_match_reference_vectors(path):
    fx = load(path)
    check_mloss(run_mloss(O.suncamera_l, O.randomcam_l, fx, torch.float64, "cpu"), fx, TOL)


@pytest.mark.parametrize("path", fixtures("tshadow_"), ids=lambda p: p.split("shade_")[-1][:-4])
def test_translucent_shadows_matches_reference_vectors(path):
    fx = load(path)
    a = torch.tensor(fx["a"], dtype=torch.float64).requires_grad_(True)
    L = O.translucentshadows_l(a)
    (float(fx["upstream"]) * L).backward()
    assert abs(float(L) - float(fx["L"])) <= TOL * abs(float(fx["L"]))
    close(a.grad.numpy(), fx["g_a"], TOL, "g_a")
