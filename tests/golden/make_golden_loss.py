"""Generates tests/golden/loss_*.npz by running the REFERENCE's own loss functions (imported from /root/reference,
which only exists in the build container) on seeded inputs. Only inputs and outputs are stored.

    python tests/golden/make_golden_loss.py
"""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/src/gaussiansplatting/utils/loss_utils.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_loss_utils", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def case(name, shape, seed, kind, lam=0.2):
    ref = load_ref()
    g = torch.Generator().manual_seed(seed)
    gt = torch.rand(shape, generator=g)
    if kind == "random":
        img = torch.rand(shape, generator=g)
    elif kind == "near":      # a render close to its ground truth: SSIM near 1, as late in training
        img = (gt + 0.03 * torch.randn(shape, generator=g)).clamp(0, 1)
    else:                      # smooth structures + exact ties (|x-y| = 0 exercises sign(0) = 0)
        yy, xx = torch.meshgrid(torch.linspace(0, 3, shape[-2]), torch.linspace(0, 5, shape[-1]), indexing="ij")
        gt = (0.5 + 0.4 * torch.sin(xx * 2.1) * torch.cos(yy * 1.7)).expand(shape).contiguous()
        img = gt.clone()
        img[..., ::2, :] += 0.1 * torch.rand(shape, generator=g)[..., ::2, :]
    out = dict(img=img.numpy(), gt=gt.numpy(), lambda_dssim=np.float32(lam))
    x = img.clone().requires_grad_(True)
    v = ref.l1_loss(x, gt); v.backward(); out.update(l1=v.item(), g_l1=x.grad.numpy().copy())
    x = img.clone().requires_grad_(True)
    v = ref.ssim(x, gt); v.backward(); out.update(ssim=v.item(), g_ssim=x.grad.numpy().copy())
    x = img.clone().requires_grad_(True)
    Ll1 = ref.l1_loss(x, gt)
    v = (1.0 - lam) * Ll1 + lam * (1.0 - ref.ssim(x, gt))  # image_utils.py:27-28 (module not importable: needs nothing else)
    v.backward(); out.update(lphotom=v.item(), g_lphotom=x.grad.numpy().copy())
    if len(shape) == 4:
        x = img.clone().requires_grad_(True)
        v = ref.ssim(x, gt, size_average=False)
        wts = torch.arange(1, shape[0] + 1, dtype=torch.float32)
        (v * wts).sum().backward()
        out.update(ssim_per_image=v.detach().numpy(), g_ssim_per_image=x.grad.numpy().copy())
    np.savez_compressed(os.path.join(OUT, f"loss_{name}.npz"), **out)
    print(name, shape, "l1", out["l1"], "ssim", out["ssim"])


if __name__ == "__main__":
    case("random_3x40x56", (3, 40, 56), 1, "random")
    case("near_3x33x70", (3, 33, 70), 2, "near")
    case("ties_1x64x64", (1, 64, 64), 3, "ties")
    case("batch_2x3x24x37", (2, 3, 24, 37), 4, "near")
    case("tiny_3x5x7", (3, 5, 7), 5, "random")
