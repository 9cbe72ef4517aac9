"""Generates tests/golden/*.npz — run in the BUILD container only (needs /root/reference).

Each fixture = inputs + expected outputs of the hot path, produced by the REFERENCE'S OWN Python
wrapper (DGR/diff_gaussian_rasterization/__init__.py, imported from /root/reference) with its
compiled `_C` extension replaced by a stub over the CPU oracle (oracle/librast_oracle.so): the
reference's argument packing, saved-tensor order, return-tuple order and grad_viewmatrix assembly are
therefore the reference's code, the arithmetic below `_C` is the oracle's restatement (the CUDA
sources cannot be built or run here: no nvcc, no glm, no NVIDIA GPU — SURVEY.md §8c).

The stub returns a [P,6] dL_dT tensor whose only non-zero row is the oracle's sum over Gaussians:
the wrapper consumes dL_dT only through a linear map followed by .sum(0), so this equals the
intended 6*idx layout (the reference kernel's own dL_dT[idx+k] layout is a data race).

Usage: python tests/golden/make_golden.py [case names: default all]
"""
import ctypes
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/src/gaussiansplatting/submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py"

import oracle  # noqa: E402
from eogs2_amd.synthetic import make_scene  # noqa: E402

abi = oracle.abi()


def _p(t):
    return None if t is None or t.numel() == 0 else ctypes.c_void_p(t.data_ptr())


def _c(t):
    return None if t is None or t.numel() == 0 else t.detach().float().contiguous()


def stub_forward(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                 projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, antialiasing, debug):
    P = means3D.shape[0]
    color = torch.zeros(5, H, W)
    invd = torch.zeros(1, H, W)
    radii = torch.zeros(P, dtype=torch.int32)
    u8 = lambda n: torch.zeros(n, dtype=torch.uint8)
    geom = binning = img = u8(0)
    R = ctypes.c_int64(0)
    if P:
        if colors.numel() == 0:
            raise RuntimeError("For non-RGB, provide precomputed Gaussian colors!")
        flags = (1 if antialiasing else 0) | (2 if debug else 0)
        n = ctypes.c_size_t()
        abi.check(abi.geom_bytes(P, ctypes.byref(n))); geom = u8(n.value)
        abi.check(abi.image_bytes(H, W, ctypes.byref(n))); img = u8(n.value)
        m, sc, ro, cv, op, vm, pm = map(_c, (means3D, scales, rotations, cov3D_precomp, opacity, viewmatrix, projmatrix))
        col, b = _c(colors), _c(bg)
        abi.check(abi.forward_prepare(P, H, W, _p(m), _p(sc), _p(ro), _p(cv), _p(op), _p(col), float(scale_modifier), _p(vm), _p(pm), None,
                                      flags, _p(radii), _p(geom), geom.numel(), None, 0, ctypes.byref(R), None))
        abi.check(abi.binning_bytes(P, H, W, R.value, ctypes.byref(n))); binning = u8(n.value)
        abi.check(abi.forward_render(P, H, W, R.value, _p(b), flags, _p(geom), geom.numel(), _p(binning), binning.numel(),
                                     _p(img), img.numel(), None, 0, _p(color), _p(invd), None))
    return R.value, color, radii, geom, binning, img, invd


def stub_backward(bg, means3D, radii, colors, opacities, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                  projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_invdepth, sh, degree, campos, geom, R,
                  binning, img, antialiasing, debug):
    P = means3D.shape[0]
    H, W = dL_dout_color.shape[1:]
    z = lambda *s: torch.zeros(*s)
    d_m2, d_col, d_op, d_m3, d_cov = z(P, 3), z(P, 5), z(P, 1), z(P, 3), z(P, 6)
    d_sc, d_rot, d_T, d_sh = z(P, 3), z(P, 4), z(P, 6), z(P, 0, 3)
    if P:
        flags = (1 if antialiasing else 0) | (2 if debug else 0)
        dTs, dvm = z(6), z(12)
        have_sr = scales.numel() != 0
        args = [_c(t) for t in (bg, means3D, None, colors, opacities, scales, rotations)]
        gcol, gdep = _c(dL_dout_color), _c(dL_dout_invdepth)
        vm, pm, cv = _c(viewmatrix), _c(projmatrix), _c(cov3D_precomp)
        abi.check(abi.backward(
            P, H, W, R, _p(args[0]), _p(args[1]), _p(radii), _p(args[3]), _p(args[4]), _p(args[5]), _p(args[6]),
            float(scale_modifier), _p(cv), _p(vm), _p(pm), None, flags, None, None, _p(gcol), _p(gdep),
            _p(geom), geom.numel(), _p(binning), binning.numel(), _p(img), img.numel(),
            _p(d_m2), _p(d_col), _p(d_op), _p(d_m3), _p(d_cov), _p(d_sc) if have_sr else None, _p(d_rot) if have_sr else None,
            _p(dTs), _p(dvm), None, 0, None))
        d_T[0] = dTs
    return d_m2, d_col, d_op, d_m3, d_cov, d_sh, d_sc, d_rot, d_T


def load_reference_wrapper():
    pkg = types.ModuleType("ref_dgr")
    pkg.__path__ = [os.path.dirname(REF)]
    sys.modules["ref_dgr"] = pkg
    c = types.ModuleType("ref_dgr._C")
    c.rasterize_gaussians = stub_forward
    c.rasterize_gaussians_backward = stub_backward
    c.mark_visible = lambda pos, vm, pm: torch.ones(pos.shape[0], dtype=torch.bool)
    sys.modules["ref_dgr._C"] = c
    spec = importlib.util.spec_from_file_location("ref_dgr", REF, submodule_search_locations=[os.path.dirname(REF)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_dgr"] = mod
    spec.loader.exec_module(mod)
    return mod


CASES = {
    # name: dict(P,H,W,seed,opacity,scale_mult,aa,precomp_cov,depth_grad,xyz_mult)
    "baseline_1k_128": dict(P=1000, H=128, W=128, seed=0, opacity="init", scale_mult=2.0),
    "ragged_aa_invdepth": dict(P=300, H=40, W=56, seed=1, opacity="trained", scale_mult=3.0, aa=True, depth_grad=True),
    "precomp_cov": dict(P=200, H=64, W=48, seed=2, opacity="trained", scale_mult=3.0, precomp_cov=True),
    "dense_termination": dict(P=2000, H=32, W=32, seed=3, opacity=0.6, scale_mult=12.0),
    "offscreen": dict(P=400, H=48, W=80, seed=4, opacity="trained", scale_mult=2.0, xyz_mult=1.6, depth_grad=True),
    "empty": dict(P=0, H=32, W=32, seed=5, opacity="init", scale_mult=1.0),
    "single": dict(P=1, H=17, W=33, seed=6, opacity=0.9, scale_mult=0.2),
    # raster_settings.scale_modifier != 1 (round 6): the covariance is built from mod * scale (forward.cu:117-151) and the
    # backward returns dL/dscale WITHOUT the factor mod (backward.cu:331-383: dL_dscale = R^T . dL_dM^T, M = S R built with the
    # modified scales) — the reference's quirk, pinned here through its own wrapper
    "scale_modifier_1p7": dict(P=600, H=56, W=72, seed=7, opacity="trained", scale_mult=1.5, aa=True, scale_modifier=1.7),
    "scale_modifier_0p5": dict(P=500, H=48, W=40, seed=8, opacity="trained", scale_mult=4.0, depth_grad=True, scale_modifier=0.5),
}


def run_case(ref, name, cfg):
    P, H, W = cfg["P"], cfg["H"], cfg["W"]
    sc = make_scene(P, H, W, seed=cfg["seed"], opacity=cfg["opacity"], scale_mult=cfg.get("scale_mult", 1.0))
    if cfg.get("xyz_mult"):
        sc["means3D"] = sc["means3D"] * torch.tensor([cfg["xyz_mult"], cfg["xyz_mult"], 1.0])
    aa = bool(cfg.get("aa", False))
    g = torch.Generator().manual_seed(77 + cfg["seed"])
    dL_dinvd = torch.randn(1, H, W, generator=g) / (H * W) * 100.0 if cfg.get("depth_grad") else None
    cov6 = None
    if cfg.get("precomp_cov"):
        from oracle.torch_dense import cov3d_full

        S = cov3d_full(sc["scales"], sc["rotations"], 1.0)
        cov6 = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).contiguous()

    leaf = lambda t: t.clone().requires_grad_(True)
    means3D, opac, colors = leaf(sc["means3D"]), leaf(sc["opacities"]), leaf(sc["colors"])
    scales = rotations = covl = None
    if cov6 is None:
        scales, rotations = leaf(sc["scales"]), leaf(sc["rotations"])
    else:
        covl = leaf(cov6)
    means2D = torch.zeros(P, 3, requires_grad=True)
    vm = leaf(sc["viewmatrix"])
    rs = ref.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(0.5), tanfovy=math.tan(0.5), bg=sc["bg"],
        scale_modifier=float(cfg.get("scale_modifier", 1.0)),
        viewmatrix=vm, projmatrix=vm.detach(), sh_degree=0, campos=torch.zeros(3), prefiltered=False, debug=False,
        antialiasing=aa)
    color, radii, invd = ref.GaussianRasterizer(rs)(
        means3D=means3D, means2D=means2D, opacities=opac, shs=None, colors_precomp=colors, scales=scales,
        rotations=rotations, cov3D_precomp=covl)
    loss = (color * sc["dL_dcolor"]).sum()
    if dL_dinvd is not None:
        loss = loss + (invd * dL_dinvd).sum()
    if P:
        loss.backward()
    out = dict(
        H=np.int32(H), W=np.int32(W), antialiasing=np.bool_(aa),
        means3D=sc["means3D"].numpy(), opacities=sc["opacities"].numpy(), colors=sc["colors"].numpy(), bg=sc["bg"].numpy(),
        viewmatrix=sc["viewmatrix"].numpy(), dL_dcolor=sc["dL_dcolor"].numpy(),
        out_color=color.detach().numpy(), out_radii=radii.numpy(), out_invdepth=invd.detach().numpy(),
    )
    if cov6 is None:
        out.update(scales=sc["scales"].numpy(), rotations=sc["rotations"].numpy())
    else:
        out.update(cov3D_precomp=cov6.numpy())
    if dL_dinvd is not None:
        out.update(dL_dinvdepth=dL_dinvd.numpy())
    if "scale_modifier" in cfg:
        out.update(scale_modifier=np.float32(cfg["scale_modifier"]))
    if P:
        grads = dict(g_means3D=means3D.grad, g_means2D=means2D.grad, g_opacities=opac.grad, g_colors=colors.grad, g_viewmatrix=vm.grad)
        if cov6 is None:
            grads.update(g_scales=scales.grad, g_rotations=rotations.grad)
        else:
            grads.update(g_cov3D_precomp=covl.grad)
        out.update({k: v.numpy() for k, v in grads.items()})
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: P={P} {H}x{W} visible={int((radii > 0).sum())} -> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    torch.manual_seed(0)
    ref = load_reference_wrapper()
    for name, cfg in CASES.items():
        if len(sys.argv) == 1 or name in sys.argv[1:]:
            run_case(ref, name, cfg)
