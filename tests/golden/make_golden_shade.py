"""Generates tests/golden/shade_*.npz by running the REFERENCE's own modules (imported from /root/reference, which only
exists in the build container) on seeded inputs. Only inputs, outputs and autograd gradients are stored.

    python tests/golden/make_golden_shade.py

`AffineCamera.render_pipeline` is called unbound on a plain namespace carrying the attributes it reads (the class's
constructor needs the dataset stack); `loss/shadow.py` is loaded without executing `loss/__init__.py` (which pulls the
renderer and the dataset readers). `RandomcamRendering_Loss.forward` renders before it compares, so its comparison part
is driven here: the occlusion map is built with the statement of main_loss.py:153-155 and handed to the reference's own
`_forward`.
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REFROOT = "/root/reference/src/gaussiansplatting"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_ref():
    sys.path.insert(0, REFROOT)
    spec = importlib.util.spec_from_file_location("ref_affine_cameras", os.path.join(REFROOT, "scene/cameras/affine_cameras.py"))
    cams = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cams)
    pkg = types.ModuleType("loss")
    pkg.__path__ = [os.path.join(REFROOT, "loss")]
    sys.modules["loss"] = pkg
    shadow = importlib.import_module("loss.shadow")
    # main_loss.py imports the renderer at module level; RandomcamRendering_Loss._forward needs none of it
    for name in ("gaussian_renderer", "gaussian_renderer.renderer_cc_shadow"):
        m = types.ModuleType(name)
        m.render_resample_virtual_camera = m.render_resample_virtual_camera_wshadowmapping = None
        sys.modules.setdefault(name, m)
    main_loss = importlib.import_module("loss.main_loss")
    return cams, shadow, main_loss


def np_(t):
    return None if t is None else t.detach().numpy().copy()


def shade_case(cams, name, H, W, seed, kind, with_shadow):
    g = torch.Generator().manual_seed(seed)
    raw = torch.rand((3, H, W), generator=g).requires_grad_(True)
    cam = types.SimpleNamespace(use_cc=False, use_exposure=False, use_shadow=with_shadow, shadow_map=cams.ShadowMap())
    params = {}
    if kind == "cc":
        cam.use_cc = True
        cam.color_correction = torch.nn.Conv2d(3, 3, 1, bias=True)
        with torch.no_grad():
            cam.color_correction.weight.copy_(torch.eye(3).reshape(3, 3, 1, 1) + 0.2 * torch.randn((3, 3, 1, 1), generator=g))
            cam.color_correction.bias.copy_(0.1 * torch.randn((3,), generator=g))
        params = dict(weight=cam.color_correction.weight, bias=cam.color_correction.bias)
    elif kind == "exposure":
        cam.use_exposure = True
        cam.exposure = torch.nn.Parameter((torch.eye(3, 4) + 0.2 * torch.randn((3, 4), generator=g))[None])
        params = dict(exposure=cam.exposure)
    cam.inshadow_color_correction = torch.nn.Parameter(torch.zeros(3).reshape(3, 1, 1) + 0.05 + 0.3 * torch.rand((3, 1, 1), generator=g))
    alt = None
    if with_shadow:
        alt = 2.0 * torch.randn((H, W), generator=g)
        alt[0, :4] = 0.0  # clip(max=0) at equality
        alt.requires_grad_(True)
        params["inshadow"] = cam.inshadow_color_correction
    out = cams.AffineCamera.render_pipeline(cam, raw, alt)
    g_shaded = torch.randn((3, H, W), generator=g)
    g_cc = torch.randn((3, H, W), generator=g)
    loss = (out["shaded"] * g_shaded).sum() + (out["cc"] * g_cc).sum()
    g_shadow = None
    if with_shadow:
        g_shadow = torch.randn((H, W), generator=g)
        loss = loss + (out["shadowmap"] * g_shadow).sum()
    assert out["final"] is not None and torch.equal(out["final"], out["shaded"])
    loss.backward()
    d = dict(kind=kind, raw=np_(raw), g_shaded=np_(g_shaded), g_cc=np_(g_cc), cc=np_(out["cc"]), shaded=np_(out["shaded"]),
             g_raw=np_(raw.grad), inshadow=np_(cam.inshadow_color_correction.reshape(3)))
    if with_shadow:
        d.update(alt_diff=np_(alt), shadow=np_(out["shadowmap"]), g_shadow=np_(g_shadow), g_alt_diff=np_(alt.grad),
                 g_inshadow=np_(cam.inshadow_color_correction.grad.reshape(3)))
    for k, v in params.items():
        if k != "inshadow":
            d[k] = np_(v)
            d["g_" + k] = np_(v.grad)
    np.savez_compressed(os.path.join(OUT, f"shade_{name}.npz"), **d)
    print(name, kind, with_shadow, float(out["shaded"].mean()))


def mloss_case(shadow, main_loss, name, H, W, seed, mode, empty=False):
    g = torch.Generator().manual_seed(seed)
    a = torch.rand((3, H, W), generator=g).requires_grad_(True)
    b = torch.rand((3, H, W), generator=g).requires_grad_(True)
    b.data[:, 1, :5] = a.data[:, 1, :5]  # exact ties: sign(0) = 0
    alt = (0.2 * torch.randn((H, W), generator=g))
    alt[2, :3] = 0.0
    uv = 1.3 * (2 * torch.rand((H, W, 2), generator=g) - 1)
    if empty:
        uv = uv.abs() + 1.0
    alt.requires_grad_(True)
    up = torch.tensor([0.7, -1.3])
    if mode == "sun":
        L = shadow.Suncamera_L(1.0, 1.0).forward(raw_render=a, sun_rgb_sample=b, sun_altitude_diff=alt, sun_uv=uv)
    else:
        occ = ((alt.abs() < 0.30) * (uv.abs() < 1).all(-1)).detach()  # main_loss.py:153-156
        fn = main_loss.RandomcamRendering_Loss(1.0, 1.0, "rawrender")
        L = fn._forward(new_occlusion_map=occ, new_altitude_diff=alt, new_rgb_diff_map=a - b)
    d = dict(mode=mode, rgb_a=np_(a), rgb_b=np_(b), alt_diff=np_(alt), uv=np_(uv), upstream=np_(up))
    if empty:
        assert L == (0, 0)
        d.update(L_alt=0.0, L_rgb=0.0, g_alt_diff=np.zeros((H, W), np.float32), g_rgb_a=np.zeros((3, H, W), np.float32),
                 g_rgb_b=np.zeros((3, H, W), np.float32))
    else:
        (up[0] * L[0] + up[1] * L[1]).backward()
        d.update(L_alt=L[0].item(), L_rgb=L[1].item(), g_alt_diff=np_(alt.grad), g_rgb_a=np_(a.grad), g_rgb_b=np_(b.grad))
    np.savez_compressed(os.path.join(OUT, f"shade_mloss_{name}.npz"), **d)
    print(name, mode, d["L_alt"], d["L_rgb"])


def tshadow_case(shadow, name, shape, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(shape, generator=g)
    a.view(-1)[:6] = torch.tensor([0.0, 1.0, 0.05, 0.95, 0.01, 0.99])
    a.requires_grad_(True)
    L = shadow.Translucentshadows_L(1.0).forward(a)
    (2.5 * L).backward()
    np.savez_compressed(os.path.join(OUT, f"shade_tshadow_{name}.npz"), a=np_(a), L=L.item(), upstream=np.float32(2.5), g_a=np_(a.grad))
    print(name, L.item())


if __name__ == "__main__":
    cams, shadow, main_loss = load_ref()
    shade_case(cams, "cc_shadow_24x20", 24, 20, 11, "cc", True)
    shade_case(cams, "exposure_noshadow_17x33", 17, 33, 12, "exposure", False)
    shade_case(cams, "identity_shadow_9x40", 9, 40, 13, "identity", True)
    shade_case(cams, "exposure_shadow_31x18", 31, 18, 14, "exposure", True)
    mloss_case(shadow, main_loss, "sun_24x20", 24, 20, 21, "sun")
    mloss_case(shadow, main_loss, "random_19x37", 19, 37, 22, "random")
    mloss_case(shadow, main_loss, "random_empty_8x8", 8, 8, 23, "random", empty=True)
    tshadow_case(shadow, "30x41", (30, 41), 31)
