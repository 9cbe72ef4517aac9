"""CPU: the photometric-loss oracle (oracle/loss_oracle.py) against golden vectors produced by the reference's own
`utils/loss_utils.py` (tests/golden/make_golden_loss.py). fp32 reference vs float64 restatement: values 1e-5, gradients 1e-4 of the tensor scale (the reference's fp32
E[x^2] - mu^2 cancellation is worth ~3e-5 on smooth images)."""
import numpy as np
import pytest
import torch

from util import GOLDEN_LOSS, assert_close, load_golden

from oracle import loss_oracle as lo


@pytest.mark.parametrize("name", GOLDEN_LOSS)
def test_loss_oracle_matches_reference_vectors(name):
    c = load_golden(name)
    gt = torch.from_numpy(c["gt"])
    lam = float(c["lambda_dssim"])

    def run(fn):
        x = torch.from_numpy(c["img"]).clone().requires_grad_(True)
        v = fn(x, gt)
        (v if v.ndim == 0 else (v * torch.arange(1, v.numel() + 1, dtype=v.dtype)).sum()).backward()
        return v.detach(), x.grad

    for key, fn in (("l1", lo.l1_loss), ("ssim", lo.ssim), ("lphotom", lambda a, b: lo.lphotom(a, b, lam))):
        v, g = run(fn)
        assert abs(float(v) - float(c[key])) <= 1e-5 * max(abs(float(c[key])), 1e-3), key
        assert_close(g, torch.from_numpy(c["g_" + key]), f"{name}:g_{key}", rtol=1e-4, allow_flips=False)
    if "ssim_per_image" in c:
        v, g = run(lambda a, b: lo.ssim(a, b, size_average=False))
        assert np.allclose(v.numpy(), c["ssim_per_image"], rtol=1e-5)
        assert_close(g, torch.from_numpy(c["g_ssim_per_image"]), f"{name}:g_ssim_per_image", rtol=1e-4, allow_flips=False)


def test_loss_host_wrapper_has_no_cpu_fallback():
    from eogs2_amd.losses import ssim

    with pytest.raises(RuntimeError):
        ssim(torch.rand(3, 16, 16), torch.rand(3, 16, 16))


def test_optimizer_has_no_cpu_fallback():
    from eogs2_amd.optim import FusedAdam, compact_rows

    p = torch.nn.Parameter(torch.zeros(8, 3))
    p.grad = torch.ones(8, 3)
    with pytest.raises(RuntimeError):
        FusedAdam([{"params": [p], "lr": 1e-3, "name": "xyz"}], lr=0.0, eps=1e-15).step()
    with pytest.raises(RuntimeError):
        compact_rows(torch.ones(8, dtype=torch.bool), [torch.zeros(8, 3)])
