"""GPU (MI355X): the fused photometric loss (include/eogs_loss.h, eogs2_amd/losses.py) against
  (1) golden vectors produced by the reference's own loss_utils.py,
  (2) the float64 oracle on larger seeded inputs (odd sizes: partial tiles, images smaller than the window),
  (3) properties at the full 3 x 1024 x 1024 size.
Tolerance 1e-4 relative to the tensor's scale (north_star), values 1e-5."""
import numpy as np
import pytest
import torch

from util import GOLDEN_LOSS, assert_close, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from eogs2_amd import _lib

    assert _lib.get().backend == "hip-gfx950"
    return torch.device("cuda:0")


def _val_grad(fn, img, gt, weights=None):
    x = img.clone().requires_grad_(True)
    v = fn(x, gt)
    (v if v.ndim == 0 else (v * weights).sum()).backward()
    return v.detach(), x.grad


def _close_val(a, b, what, rtol=1e-5):
    a, b = float(a), float(b)
    assert abs(a - b) <= rtol * max(abs(b), 1e-3), f"{what}: {a} vs {b}"


@pytest.mark.parametrize("name", GOLDEN_LOSS)
def test_loss_matches_reference_vectors(dev, name):
    from eogs2_amd import losses

    c = load_golden(name)
    img, gt = torch.from_numpy(c["img"]).to(dev), torch.from_numpy(c["gt"]).to(dev)
    lam = float(c["lambda_dssim"])
    fns = {"l1": losses.l1_loss, "ssim": losses.ssim,
           "lphotom": lambda a, b: losses.lphotom(a, b, losses.l1_loss(a, b), lam)}
    for key, fn in fns.items():
        v, g = _val_grad(fn, img, gt)
        _close_val(v, c[key], f"{name}:{key}")
        assert_close(g, torch.from_numpy(c["g_" + key]), f"{name}:g_{key}", allow_flips=False)
    # the fused pair: same value and gradient as lphotom, one kernel each way
    v, g = _val_grad(lambda a, b: losses.photometric_loss(a, b, lam)[0], img, gt)
    _close_val(v, c["lphotom"], f"{name}:photometric_loss")
    assert_close(g, torch.from_numpy(c["g_lphotom"]), f"{name}:g_photometric_loss", allow_flips=False)
    _close_val(losses.photometric_loss(img, gt, lam)[1], c["l1"], f"{name}:Ll1")
    if "ssim_per_image" in c:
        w = torch.arange(1, img.shape[0] + 1, dtype=torch.float32, device=dev)
        v, g = _val_grad(lambda a, b: losses.ssim(a, b, size_average=False), img, gt, w)
        assert np.allclose(v.cpu().numpy(), c["ssim_per_image"], rtol=1e-5)
        assert_close(g, torch.from_numpy(c["g_ssim_per_image"]), f"{name}:g_ssim_per_image", allow_flips=False)


@pytest.mark.parametrize("shape", [(3, 257, 131), (1, 7, 300), (5, 96, 96), (2, 3, 65, 33)])
def test_loss_matches_oracle(dev, shape):
    from oracle import loss_oracle as lo

    from eogs2_amd import losses

    g = torch.Generator().manual_seed(sum(shape))
    gt = torch.rand(shape, generator=g)
    img = (gt + 0.1 * torch.randn(shape, generator=g)).clamp(0, 1)
    for key, hip_fn, ora_fn in (
        ("l1", losses.l1_loss, lo.l1_loss),
        ("ssim", losses.ssim, lo.ssim),
        ("photometric", lambda a, b: losses.photometric_loss(a, b, 0.2)[0], lambda a, b: lo.lphotom(a, b, 0.2)),
    ):
        v, gr = _val_grad(hip_fn, img.to(dev), gt.to(dev))
        vo, go = _val_grad(ora_fn, img, gt)
        _close_val(v, vo, f"{shape}:{key}")
        assert_close(gr, go, f"{shape}:g_{key}", allow_flips=False)


def test_loss_full_size_properties(dev):
    from eogs2_amd import losses

    g = torch.Generator().manual_seed(9)
    x = torch.rand(3, 1024, 1024, generator=g).to(dev)
    y = torch.rand(3, 1024, 1024, generator=g).to(dev)
    # identical images: SSIM = 1, L1 = 0, loss = 0, zero L1 gradient (sign(0) = 0)
    v, gr = _val_grad(lambda a, b: losses.photometric_loss(a, b, 0.2)[0], x, x.clone())
    assert abs(float(v)) <= 1e-6 and float(losses.ssim(x, x)) == pytest.approx(1.0, abs=1e-6)
    assert float(gr.abs().max()) <= 1e-6 / x.numel() * 1e3
    # symmetry of SSIM, bitwise determinism, linearity of the gradient in the upstream scalar
    assert float(losses.ssim(x, y)) == pytest.approx(float(losses.ssim(y, x)), rel=1e-6)
    v1, g1 = _val_grad(lambda a, b: losses.photometric_loss(a, b, 0.2)[0], x, y)
    v2, g2 = _val_grad(lambda a, b: losses.photometric_loss(a, b, 0.2)[0], x, y)
    assert torch.equal(v1, v2) and torch.equal(g1, g2)
    _, g3 = _val_grad(lambda a, b: 3.0 * losses.photometric_loss(a, b, 0.2)[0], x, y)
    assert_close(g3, 3.0 * g1, "upstream scaling", rtol=1e-6, allow_flips=False)
    # fused == composition of the two separate entry points
    _, gl = _val_grad(losses.l1_loss, x, y)
    _, gs = _val_grad(losses.ssim, x, y)
    assert_close(g1, 0.8 * gl - 0.2 * gs, "fused vs separate", rtol=1e-5, allow_flips=False)


def test_loss_argument_errors(dev):
    from eogs2_amd import losses

    a = torch.rand(3, 16, 16, device=dev)
    with pytest.raises(RuntimeError):
        losses.ssim(a, torch.rand(3, 16, 17, device=dev))
    with pytest.raises(NotImplementedError):
        losses.ssim(a, a, window_size=7)
    with pytest.raises(IndexError):
        losses.ssim(a, a, size_average=False)
    # non-fp32 / non-contiguous inputs are accepted like the reference's conv2d path would after .float()
    b = torch.rand(3, 16, 32, device=dev)[:, :, ::2]
    assert float(losses.l1_loss(b, a)) >= 0
