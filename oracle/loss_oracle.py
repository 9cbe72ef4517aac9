"""CPU oracle of the photometric loss — TEST INFRASTRUCTURE ONLY (never imported by eogs2_amd/).

A float64 restatement of the reference's algorithm:
  l1_loss  src/gaussiansplatting/utils/loss_utils.py:18-19
  gaussian / create_window (sigma 1.5, 11 taps, fp32 outer product)  loss_utils.py:26-42
  ssim / _ssim (five depthwise conv2d, zero padding 5, C1=0.01^2, C2=0.03^2)  loss_utils.py:45-85
  lphotom  src/gaussiansplatting/utils/image_utils.py:27-28
Gradients come from torch autograd over this restatement. Pinned against the reference's own functions through
tests/golden/loss_*.npz (tests/golden/make_golden_loss.py imports the reference module to generate them).
"""
from math import exp

import torch
import torch.nn.functional as F


def window_2d(window_size=11, sigma=1.5):
    g = torch.tensor([exp(-((x - window_size // 2) ** 2) / float(2 * sigma**2)) for x in range(window_size)],
                     dtype=torch.float32)
    g = (g / g.sum()).unsqueeze(1)
    return g.mm(g.t()).double()  # the reference builds the 2-D window in fp32 (loss_utils.py:36-38)


def l1_loss(x, y):
    return (x.double() - y.double()).abs().mean()


def ssim_map(x, y, window_size=11):
    x, y = x.double(), y.double()
    squeeze = x.ndim == 3
    if squeeze:
        x, y = x[None], y[None]
    C = x.shape[1]
    w = window_2d(window_size).expand(C, 1, window_size, window_size).contiguous()
    conv = lambda t: F.conv2d(t, w, padding=window_size // 2, groups=C)
    mu1, mu2 = conv(x), conv(y)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s11, s22, s12 = conv(x * x) - mu1_sq, conv(y * y) - mu2_sq, conv(x * y) - mu12
    C1, C2 = 0.01**2, 0.03**2
    m = ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s11 + s22 + C2))
    return m[0] if squeeze else m


def ssim(x, y, window_size=11, size_average=True):
    m = ssim_map(x, y, window_size)
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)


def lphotom(x, y, lambda_dssim):
    return (1.0 - lambda_dssim) * l1_loss(x, y) + lambda_dssim * (1.0 - ssim(x, y))
