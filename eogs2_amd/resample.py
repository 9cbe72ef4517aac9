"""Virtual-camera resample over the C-ABI of include/eogs_resample.h (second piece).

* `resample(virtual_render, cam2virt, rendered_uva, n_out=4, fill_channel=3, fill_value=-100.0) -> (sample, uv)` —
  steps 2-3 of the reference's `render_resample_virtual_camera`
  (src/gaussiansplatting/gaussian_renderer/renderer_cc_shadow.py:32-50) in one HIP kernel each way.
* `render_resample_virtual_camera(virtual_camera, cam2virt, rendered_uva, gaussians, pipe, background,
  return_extra=False)` — the reference's function, same signature and return values, built from
  `eogs2_amd.render.render` (step 1) and `resample`.

Gradients flow to `virtual_render` and `rendered_uva` (hence to the true camera's altitude render); `cam2virt` is
treated as a constant, as in the reference where it is built from fixed camera matrices. No CPU / eager fallback.
"""
import ctypes

import torch

from . import _lib
from .rasterizer import _Ctx, _ptr


def _f32c(t, dev):
    if t.device != dev:
        raise RuntimeError(f"resample input on {t.device}, expected {dev}")
    return t.detach().to(torch.float32).contiguous()


class _Resample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, virtual_render, rendered_uva, cam2virt, n_out, fill_channel, fill_value):
        abi = _lib.get()
        if virtual_render.ndim != 3 or rendered_uva.ndim != 3 or rendered_uva.shape[-1] != 3:
            raise RuntimeError("resample: virtual_render must be (C, Hv, Wv) and rendered_uva (H, W, 3)")
        if tuple(cam2virt.shape) != (3, 3):
            raise RuntimeError("resample: cam2virt must be 3x3")
        C, Hv, Wv = virtual_render.shape
        H, W = rendered_uva.shape[:2]
        dev = virtual_render.device
        vr, uva, M = _f32c(virtual_render, dev), _f32c(rendered_uva, dev), _f32c(cam2virt, dev)
        with _Ctx(abi, dev) as cx:
            sample = torch.empty((n_out, H, W), dtype=torch.float32, device=dev)
            uv = torch.empty((H, W, 2), dtype=torch.float32, device=dev)
            abi.check(abi.resample_forward(C, Hv, Wv, H, W, n_out, _ptr(vr), _ptr(uva), _ptr(M), fill_channel,
                                           float(fill_value), _ptr(sample), _ptr(uv), cx.stream))
        ctx.cfg = (C, Hv, Wv, H, W, n_out, fill_channel)
        ctx.save_for_backward(vr, uva, M)
        ctx.set_materialize_grads(False)
        return sample, uv

    @staticmethod
    def backward(ctx, g_sample, g_uv):
        if g_sample is None and g_uv is None:
            return (None,) * 6
        abi = _lib.get()
        C, Hv, Wv, H, W, n_out, fill_channel = ctx.cfg
        vr, uva, M = ctx.saved_tensors
        dev = vr.device
        with _Ctx(abi, dev) as cx:
            gs = _f32c(g_sample, dev) if g_sample is not None else torch.zeros((n_out, H, W), dtype=torch.float32, device=dev)
            guv = _f32c(g_uv, dev) if g_uv is not None else None
            g_vr = torch.empty((C, Hv, Wv), dtype=torch.float32, device=dev)
            g_uva = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
            n = ctypes.c_size_t()
            abi.check(abi.resample_bytes(H, W, ctypes.byref(n)))
            ws = torch.empty((n.value,), dtype=torch.uint8, device=dev)
            abi.check(abi.resample_backward(C, Hv, Wv, H, W, n_out, _ptr(vr), _ptr(uva), _ptr(M), fill_channel, _ptr(gs),
                                            _ptr(guv), _ptr(g_vr), _ptr(g_uva), _ptr(ws), ws.numel(), cx.stream))
        return g_vr, g_uva, None, None, None, None


def resample(virtual_render, cam2virt, rendered_uva, n_out=4, fill_channel=3, fill_value=-100.0):
    return _Resample.apply(virtual_render, rendered_uva, cam2virt, n_out, fill_channel, fill_value)


def render_resample_virtual_camera(virtual_camera, cam2virt, rendered_uva, gaussians, pipe, background, return_extra=False,
                                   altitude_only=False):
    """renderer_cc_shadow.py:5-60: render the virtual camera, reproject the true camera's (u, v, altitude) grid into it,
    sample the virtual image there. Returns (rgb_sample[3,H,W], altitude_sample[H,W], virtual_uv[H,W,2]).

    `altitude_only=True` (not in the reference's signature): for a caller that does not consume `rgb_sample` — the sun camera
    while `iterstart_L_sun_resample` has not been reached, i.e. always with the shipped configuration (train_pan.py:305-324,
    gs_config/train.yaml:123) — the virtual camera renders its altitude channel alone (EOGS_FLAG_ALT_ONLY: one blended
    channel instead of five, forward and backward, on the iteration's largest render) and one plane is resampled;
    `rgb_sample` comes back as None. The altitude sample, the coordinates and every gradient equal the full call's."""
    from .render import render

    if altitude_only:
        virtual_render = render(virtual_camera, gaussians, pipe, background, altitude_only=True)["render"]  # [1, Hv, Wv]
        sample, virtual_uv = resample(virtual_render, cam2virt, rendered_uva, n_out=1, fill_channel=0)
        if return_extra:
            return None, sample[0], virtual_uv, virtual_render
        return None, sample[0], virtual_uv
    virtual_render = render(virtual_camera, gaussians, pipe, background)["render"]
    sample, virtual_uv = resample(virtual_render, cam2virt, rendered_uva)
    if return_extra:
        return sample[:3], sample[3], virtual_uv, virtual_render
    return sample[:3], sample[3], virtual_uv


__all__ = ["resample", "render_resample_virtual_camera"]
