"""Generates tests/golden/resample_*.npz by running the REFERENCE's own `render_resample_virtual_camera`
(src/gaussiansplatting/gaussian_renderer/renderer_cc_shadow.py:5-50, imported from /root/reference, which only exists in the
build container) on seeded inputs. Only inputs, outputs and autograd gradients are stored.

    python tests/golden/make_golden_resample.py

`renderer_cc_shadow.py` does `from .renderer import render` (the rasterizer-backed renderer with the dataset stack behind
it): the file is loaded as a submodule of a stub package `gaussian_renderer` whose `renderer.render` returns the case's
seeded virtual render (step 1 of the function is the rasterizer, pinned elsewhere); steps 2-3 — the UVA -> UVA
reprojection, `grid_sample(align_corners=True)` and the -100 fill outside the virtual view — are the reference's own
statements, executed by autograd in fp32 as the reference runs them. The true camera's (u, v, altitude) grid is built as
train_pan.py:281 builds it; gradients reach the virtual render and the altitude.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REFROOT = "/root/reference/src/gaussiansplatting"
OUT = os.path.dirname(os.path.abspath(__file__))
_STATE = {}


def load_ref():
    pkg = types.ModuleType("gaussian_renderer")
    pkg.__path__ = [os.path.join(REFROOT, "gaussian_renderer")]
    sys.modules["gaussian_renderer"] = pkg
    rmod = types.ModuleType("gaussian_renderer.renderer")
    rmod.render = lambda cam, gaussians, pipe, bg: {"render": _STATE["virtual_render"]}
    sys.modules["gaussian_renderer.renderer"] = rmod
    spec = importlib.util.spec_from_file_location("gaussian_renderer.renderer_cc_shadow",
                                                  os.path.join(REFROOT, "gaussian_renderer", "renderer_cc_shadow.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def make_case(ref, name, H, W, f, seed, shear=0.08, shift=0.0, channels=5, offdiag=True):
    g = torch.Generator().manual_seed(seed)
    vr = torch.rand(channels, H * f, W * f, generator=g)
    vr[3] = vr[3] * 40 - 10  # altitude-like channel
    alt = torch.rand(H, W, generator=g) * 2 - 0.5
    U, V = torch.meshgrid(torch.linspace(-1, 1, W), torch.linspace(-1, 1, H), indexing="xy")
    M = torch.eye(3)
    M[:2, 2] = torch.tensor([shear, -0.7 * shear])
    M[0, 0], M[1, 1] = 1.0 + shift, 1.0 - 0.5 * shift  # pushes part of the grid outside [-1, 1]
    if offdiag:
        M[0, 1], M[1, 0] = 0.03, -0.02
    else:
        M[1, 1] = M[0, 0]  # both axes shrunk: every pixel lands inside the virtual view
    w_rgb = torch.randn(3, H, W, generator=g)
    w_alt = torch.randn(H, W, generator=g)
    w_uv = torch.randn(H, W, 2, generator=g)
    vr.requires_grad_(True)
    alt.requires_grad_(True)
    _STATE["virtual_render"] = vr
    uva = torch.stack((U, V, alt), dim=-1)  # train_pan.py:281
    rgb, a, uv, vr_out = ref.render_resample_virtual_camera(None, M, uva, None, None, None, return_extra=True)
    assert vr_out is vr
    ((rgb * w_rgb).sum() + (a * w_alt).sum() + (uv * w_uv).sum()).backward()
    n = lambda t: t.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"resample_{name}.npz"), virtual_render=n(vr), altitude=n(alt), U=n(U), V=n(V), cam2virt=n(M),
                        w_rgb=n(w_rgb), w_alt=n(w_alt), w_uv=n(w_uv), rgb_sample=n(rgb), altitude_sample=n(a), virtual_uv=n(uv),
                        g_virtual_render=n(vr.grad), g_altitude=n(alt.grad))
    print(f"resample_{name}: {H}x{W} <- {H * f}x{W * f}, filled pixels {int((a == -100).sum())}")


def main():
    ref = load_ref()
    make_case(ref, "inside_40x56_x2", 40, 56, 2, seed=1, shear=0.01, shift=-0.06, offdiag=False)
    make_case(ref, "partly_outside_33x47", 33, 47, 1, seed=2, shift=0.3)
    make_case(ref, "shrunk_64x64_x2", 64, 64, 2, seed=3, shift=-0.2)
    make_case(ref, "four_channels_17x90", 17, 90, 1, seed=4, shift=0.1, channels=4)


if __name__ == "__main__":
    main()
