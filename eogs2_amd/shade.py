"""Image chain between the raw render and the scalar losses over the C-ABI of include/eogs_shade.h.

Same names, arguments and return values as the reference (paths under src/gaussiansplatting/):

* `render_pipeline(cam, raw_render, sun_altitude_diff=None)` — `AffineCamera.render_pipeline`
  (scene/cameras/affine_cameras.py:303-348): colour correction (1x1 conv or exposure matrix), `ShadowMap` (:33-40) and the
  in-shadow tint in one kernel each way; returns the same dict (`shadowmap`, `shaded`, `cc`, `final`). `cam` is
  duck-typed: `use_cc`/`color_correction`, `use_exposure`/`exposure`, `use_shadow`, `inshadow_color_correction`.
* `suncamera_l(raw_render, sun_rgb_sample, sun_altitude_diff, sun_uv)` — `Suncamera_L.forward` (loss/shadow.py:37-51).
* `randomcam_l(new_altitude_diff, rgb_render, new_rgb_sample, new_uv)` — the arithmetic of `RandomcamRendering_Loss.forward`
  after its resample (loss/main_loss.py:151-164 with `_forward`, :83-96).
* `translucentshadows_l(shadowmap)` — `Translucentshadows_L.forward` (loss/shadow.py:13-17).

Each is one forward and one backward kernel (plus a tiny fixed-order reduction) where the reference runs 6-15 elementwise
PyTorch kernels and autograd replays them. PyTorch is plumbing; there is no CPU or eager fallback.
"""
import ctypes

import torch

from . import _lib
from ._abi import MLOSS_RANDOM, MLOSS_SUN
from .rasterizer import _Ctx, _ptr


def _f32c(t, dev):
    if t.device != dev:
        raise RuntimeError(f"shade input on {t.device}, expected {dev}")
    return t.detach().to(torch.float32).contiguous()


def _ws(abi, H, W, dev):
    n = ctypes.c_size_t()
    abi.check(abi.shade_bytes(H, W, ctypes.byref(n)))
    return torch.empty((n.value,), dtype=torch.uint8, device=dev)


class _Shade(torch.autograd.Function):
    """(cc, shaded, shadow) = shade(raw[3,H,W], alt_diff[H,W] | None, M[3,4], inshadow[3])"""

    @staticmethod
    def forward(ctx, raw, alt_diff, M, inshadow):
        abi = _lib.get()
        if raw.ndim != 3 or raw.shape[0] != 3:
            raise RuntimeError(f"render_pipeline: raw_render must be (3, H, W), got {tuple(raw.shape)}")
        _, H, W = raw.shape
        if alt_diff is not None and tuple(alt_diff.shape) != (H, W):
            raise RuntimeError(f"render_pipeline: sun_altitude_diff must be ({H}, {W}), got {tuple(alt_diff.shape)}")
        dev = raw.device
        x, m = _f32c(raw, dev), _f32c(M, dev).reshape(3, 4)
        d = _f32c(alt_diff, dev) if alt_diff is not None else None
        ins = _f32c(inshadow, dev).reshape(3) if alt_diff is not None else None
        with _Ctx(abi, dev) as cx:
            cc = torch.empty((3, H, W), dtype=torch.float32, device=dev)
            shaded = torch.empty((3, H, W), dtype=torch.float32, device=dev)
            shadow = torch.empty((H, W), dtype=torch.float32, device=dev) if d is not None else None
            abi.check(abi.shade_forward(H, W, _ptr(x), _ptr(d), _ptr(m), _ptr(ins), _ptr(cc), _ptr(shaded), _ptr(shadow),
                                        cx.stream))
        ctx.cfg = (H, W, d is not None, M.shape, inshadow.shape if inshadow is not None else None)
        ctx.save_for_backward(x, d, m, ins)
        ctx.set_materialize_grads(False)
        return cc, shaded, shadow

    @staticmethod
    def backward(ctx, g_cc, g_shaded, g_shadow):
        if g_cc is None and g_shaded is None and g_shadow is None:
            return None, None, None, None
        abi = _lib.get()
        H, W, has_shadow, m_shape, ins_shape = ctx.cfg
        x, d, m, ins = ctx.saved_tensors
        dev = x.device
        with _Ctx(abi, dev) as cx:
            gs = _f32c(g_shaded, dev) if g_shaded is not None else torch.zeros((3, H, W), dtype=torch.float32, device=dev)
            gc = _f32c(g_cc, dev) if g_cc is not None else None
            gsh = _f32c(g_shadow, dev) if (g_shadow is not None and has_shadow) else None
            g_raw = torch.empty((3, H, W), dtype=torch.float32, device=dev)
            g_alt = torch.empty((H, W), dtype=torch.float32, device=dev) if has_shadow else None
            g_par = torch.empty((15,), dtype=torch.float32, device=dev)
            ws = _ws(abi, H, W, dev)
            abi.check(abi.shade_backward(H, W, _ptr(x), _ptr(d), _ptr(m), _ptr(ins), _ptr(gs), _ptr(gc), _ptr(gsh),
                                         _ptr(g_raw), _ptr(g_alt), _ptr(g_par), _ptr(ws), ws.numel(), cx.stream))
        g_M = g_par[:12].view(3, 4).reshape(m_shape)
        g_ins = g_par[12:].reshape(ins_shape) if has_shadow else None
        return g_raw, g_alt, g_M, g_ins


def shade(raw_render, sun_altitude_diff, M, inshadow):
    """cc = M[:, :3] @ raw + M[:, 3]; shadow = exp(0.4 min(d, 0)); shaded = shadow cc + (1 - shadow) inshadow cc."""
    return _Shade.apply(raw_render, sun_altitude_diff, M, inshadow)


def render_pipeline(cam, raw_render, sun_altitude_diff=None):
    """AffineCamera.render_pipeline (affine_cameras.py:303-348) for a duck-typed camera."""
    dev = raw_render.device
    if getattr(cam, "use_cc", False):
        conv = cam.color_correction
        M = torch.cat([conv.weight.reshape(3, 3), conv.bias.reshape(3, 1)], dim=1)
    elif getattr(cam, "use_exposure", False):
        M = cam.exposure[0]
    else:
        M = torch.eye(3, 4, device=dev)
    use_shadow = bool(getattr(cam, "use_shadow", False)) and sun_altitude_diff is not None
    if use_shadow:
        cc, shaded, shadow = shade(raw_render, sun_altitude_diff, M, cam.inshadow_color_correction.reshape(3))
    else:
        cc, shaded, shadow = shade(raw_render, None, M, None)
    return {"shadowmap": shadow, "shaded": shaded, "cc": cc, "final": shaded}


class _MaskedLoss(torch.autograd.Function):
    """out[3] = {L_alt, L_rgb, mask count}"""

    @staticmethod
    def forward(ctx, alt_diff, rgb_a, rgb_b, uv, mode):
        abi = _lib.get()
        if alt_diff.ndim != 2:
            raise RuntimeError("masked resample loss: altitude difference must be (H, W)")
        H, W = alt_diff.shape
        if tuple(rgb_a.shape) != (3, H, W) or tuple(rgb_b.shape) != (3, H, W) or tuple(uv.shape) != (H, W, 2):
            raise RuntimeError("masked resample loss: expected rgb (3, H, W) twice and uv (H, W, 2)")
        dev = alt_diff.device
        d, a, b, u = _f32c(alt_diff, dev), _f32c(rgb_a, dev), _f32c(rgb_b, dev), _f32c(uv, dev)
        with _Ctx(abi, dev) as cx:
            out = torch.empty((3,), dtype=torch.float32, device=dev)
            ws = _ws(abi, H, W, dev)
            abi.check(abi.mloss_forward(H, W, mode, _ptr(d), _ptr(a), _ptr(b), _ptr(u), _ptr(out), _ptr(ws), ws.numel(),
                                        cx.stream))
        ctx.cfg = (H, W, mode)
        ctx.save_for_backward(d, a, b, u, out)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, g_out):
        if g_out is None:
            return (None,) * 5
        abi = _lib.get()
        H, W, mode = ctx.cfg
        d, a, b, u, out = ctx.saved_tensors
        dev = d.device
        with _Ctx(abi, dev) as cx:
            up = _f32c(g_out, dev)[:2].contiguous()
            g_alt = torch.empty((H, W), dtype=torch.float32, device=dev)
            g_a = torch.empty((3, H, W), dtype=torch.float32, device=dev)
            g_b = torch.empty((3, H, W), dtype=torch.float32, device=dev) if ctx.needs_input_grad[2] else None
            abi.check(abi.mloss_backward(H, W, mode, _ptr(d), _ptr(a), _ptr(b), _ptr(u), _ptr(out), _ptr(up), _ptr(g_alt),
                                         _ptr(g_a), _ptr(g_b), cx.stream))
        return g_alt, g_a, g_b, None, None


def suncamera_l(raw_render, sun_rgb_sample, sun_altitude_diff, sun_uv):
    """(L_sun_altitude_resample, L_sun_rgb_resample), loss/shadow.py:37-51. An empty visibility map gives zeros (the
    reference's forward falls off its `if` and returns None there)."""
    out = _MaskedLoss.apply(sun_altitude_diff, raw_render, sun_rgb_sample, sun_uv, MLOSS_SUN)
    return out[0], out[1]


def randomcam_l(new_altitude_diff, rgb_render, new_rgb_sample, new_uv):
    """(L_new_altitude_resample, L_new_rgb_resample), loss/main_loss.py:151-164: occlusion map |diff| < 0.30 inside the
    virtual view, masked mean absolute differences; zeros for an empty map."""
    out = _MaskedLoss.apply(new_altitude_diff, rgb_render, new_rgb_sample, new_uv, MLOSS_RANDOM)
    return out[0], out[1]


class _TShadow(torch.autograd.Function):
    @staticmethod
    def forward(ctx, shadowmap):
        abi = _lib.get()
        dev = shadowmap.device
        a = _f32c(shadowmap, dev)
        n = a.numel()
        if n == 0:
            raise RuntimeError("translucentshadows_l: empty shadow map")
        with _Ctx(abi, dev) as cx:
            out = torch.empty((1,), dtype=torch.float32, device=dev)
            ws = _ws(abi, 1, 1, dev)
            abi.check(abi.tshadow_forward(n, _ptr(a), _ptr(out), _ptr(ws), ws.numel(), cx.stream))
        ctx.save_for_backward(a)
        ctx.shape, ctx.dtype = shadowmap.shape, shadowmap.dtype
        return out[0]

    @staticmethod
    def backward(ctx, g):
        abi = _lib.get()
        (a,) = ctx.saved_tensors
        dev = a.device
        with _Ctx(abi, dev) as cx:
            up = _f32c(g, dev).reshape(1)
            g_a = torch.empty_like(a)
            abi.check(abi.tshadow_backward(a.numel(), _ptr(a), _ptr(up), _ptr(g_a), cx.stream))
        return g_a.view(ctx.shape).to(ctx.dtype)


def translucentshadows_l(shadowmap):
    """-mean(a log2 b + (1 - a) log2(1 - b)), b = clip(a, 0.05, 0.95) (loss/shadow.py:13-17)."""
    return _TShadow.apply(shadowmap)


__all__ = ["shade", "render_pipeline", "suncamera_l", "randomcam_l", "translucentshadows_l"]
