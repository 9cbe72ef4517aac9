"""Image-space photometric loss over the C-ABI of include/eogs_loss.h (first piece).

Same names, arguments and return values as the reference:

* `l1_loss(network_output, gt)`                         — src/gaussiansplatting/utils/loss_utils.py:18-19
* `ssim(img1, img2, window_size=11, size_average=True)` — utils/loss_utils.py:45-85
* `lphotom(image, gt_image, Ll1, lambda_dssim)`         — utils/image_utils.py:27-28
* `photometric_loss(image, gt_image, lambda_dssim)`     — the fused form of the reference's usual pair
  `Ll1 = l1_loss(image, gt); loss = lphotom(image, gt, Ll1, lambda_dssim)` in one forward and one backward kernel;
  returns `(loss, Ll1)`.

Gradients flow to the first argument only (the rendered image); the reference never differentiates the ground truth.
PyTorch is plumbing (memory, streams, autograd wiring); there is no CPU or eager fallback.
"""
import ctypes

import torch

from . import _lib
from ._abi import LOSS_L1, LOSS_SSIM
from .rasterizer import _Ctx, _ptr


def _f32c(t, dev):
    if t.device != dev:
        raise RuntimeError(f"loss input on {t.device}, expected {dev}")
    return t.detach().to(torch.float32).contiguous()


class _Photometric(torch.autograd.Function):
    """(out[3], plane_sums[planes,2]); out = {w_l1*l1_mean + w_ssim*ssim_mean + bias, l1_mean, ssim_mean}."""

    @staticmethod
    def forward(ctx, img, gt, mode, w_l1, w_ssim, bias, want_plane_sums):
        abi = _lib.get()
        if img.shape != gt.shape:
            raise RuntimeError(f"loss: shapes differ: {tuple(img.shape)} vs {tuple(gt.shape)}")
        if img.ndim < 2 or img.numel() == 0:
            raise RuntimeError("loss: expected non-empty (..., H, W) images")
        H, W = int(img.shape[-2]), int(img.shape[-1])
        planes = img.numel() // (H * W)
        dev = img.device
        x, y = _f32c(img, dev), _f32c(gt, dev)
        with _Ctx(abi, dev) as cx:
            n = ctypes.c_size_t()
            abi.check(abi.loss_bytes(planes, H, W, mode, ctypes.byref(n)))
            ws = torch.empty((n.value,), dtype=torch.uint8, device=dev)
            out = torch.empty((3,), dtype=torch.float32, device=dev)
            psum = torch.empty((planes, 2) if want_plane_sums else (0, 2), dtype=torch.float32, device=dev)
            abi.check(abi.loss_forward(planes, H, W, _ptr(x), _ptr(y), mode, w_l1, w_ssim, bias, _ptr(out),
                                       _ptr(psum) if want_plane_sums else None, _ptr(ws), ws.numel(), cx.stream))
        ctx.cfg = (planes, H, W, mode, w_l1, w_ssim)
        ctx.img_shape, ctx.img_dtype = img.shape, img.dtype
        ctx.save_for_backward(x, y, ws)
        ctx.set_materialize_grads(False)
        return out, psum

    @staticmethod
    def backward(ctx, g_out, g_psum):
        none7 = (None,) * 7
        if not ctx.needs_input_grad[0] or (g_out is None and g_psum is None):
            return none7
        abi = _lib.get()
        planes, H, W, mode, w_l1, w_ssim = ctx.cfg
        x, y, ws = ctx.saved_tensors
        dev = x.device
        with _Ctx(abi, dev) as cx:
            d = torch.empty((planes, H, W), dtype=torch.float32, device=dev)
            if g_psum is not None and g_psum.numel():
                if g_out is not None:
                    raise RuntimeError("loss: use either the scalar outputs or the per-plane sums of one call, not both")
                pg = _f32c(g_psum, dev)
                abi.check(abi.loss_backward(planes, H, W, _ptr(x), _ptr(y), mode, 0.0, 0.0, None, _ptr(pg),
                                            _ptr(ws), ws.numel(), _ptr(d), cx.stream))
            else:
                g = _f32c(g_out, dev)
                abi.check(abi.loss_backward(planes, H, W, _ptr(x), _ptr(y), mode, w_l1, w_ssim, _ptr(g), None,
                                            _ptr(ws), ws.numel(), _ptr(d), cx.stream))
        return (d.view(ctx.img_shape).to(ctx.img_dtype),) + none7[1:]


def l1_loss(network_output, gt):
    out, _ = _Photometric.apply(network_output, gt, LOSS_L1, 1.0, 0.0, 0.0, False)
    return out[1]


def ssim(img1, img2, window_size=11, size_average=True):
    if window_size != 11:
        raise NotImplementedError("the fused SSIM is built for the reference's window_size=11 (loss_utils.py:45)")
    if size_average:
        out, _ = _Photometric.apply(img1, img2, LOSS_SSIM, 0.0, 1.0, 0.0, False)
        return out[2]
    if img1.ndim != 4:
        # loss_utils.py:85 reduces three trailing dimensions of a 4-D map
        raise IndexError("Dimension out of range: size_average=False needs (N, C, H, W) inputs")
    N, C, H, W = img1.shape
    _, psum = _Photometric.apply(img1, img2, LOSS_SSIM, 0.0, 1.0, 0.0, True)
    return psum[:, 1].view(N, C).sum(1) / float(C * H * W)


def photometric_loss(image, gt_image, lambda_dssim):
    """(loss, Ll1) with loss = (1 - lambda) * L1 + lambda * (1 - SSIM), one fused forward + one fused backward."""
    lam = float(lambda_dssim)
    out, _ = _Photometric.apply(image, gt_image, LOSS_L1 | LOSS_SSIM, 1.0 - lam, -lam, lam, False)
    return out[0], out[1].detach()


def lphotom(image, gt_image, Ll1, lambda_dssim):
    """Reference signature (image_utils.py:27-28): `Ll1` is the caller's own L1 term and keeps its own graph."""
    return (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - ssim(image, gt_image))


__all__ = ["l1_loss", "ssim", "lphotom", "photometric_loss"]
