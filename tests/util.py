"""Shared helpers for the parity tests."""
import glob
import math
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_ALL_NPZ = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
GOLDEN = [n for n in _ALL_NPZ if not n.startswith(("loss_", "shade_", "tsdf_", "resample_"))]  # rasterizer cases (make_golden.py)
GOLDEN_LOSS = [n for n in _ALL_NPZ if n.startswith("loss_")]       # photometric-loss cases (make_golden_loss.py)

# north_star tolerance: "within 1e-4 rel".  Applied as |a-b| <= RTOL * max|b| per QUANTITY — channel of an image, column of a
# per-Gaussian gradient (gradient sums are order-dependent fp32 sums, so the scale of the quantity is the meaningful unit).
RTOL = 1e-4
# A pixel can legitimately differ by one blended/skipped Gaussian when alpha sits within a few ulp of the
# 1/255 threshold (expf differs between libm and the GPU): such a flip moves a pixel by <= alpha*T*|c| ~ 4e-3|c|.
# We allow at most FLIP_FRAC of the elements outside RTOL, and none beyond FLIP_RTOL.
FLIP_FRAC = 2e-5
FLIP_RTOL = 1e-2
# Every case is held to RTOL. (Rounds 1-5: the fast backward kernels walked front to back and differed from the reference's
# back-to-front recursion by a few 1e-4 on image-sized opaque Gaussians stacked hundreds deep; such forwards were switched to a
# second kernel by a threshold, and test processes that forced the fast kernels onto them carried a 5e-4 table here. Since
# round 6 every backward kernel is the reference's recursion: no switch, no table. DESIGN.md 5.)
GRAD_RTOL = {}


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: z[k] for k in z.files}


def closeness(a, b, with_items=False):
    """(max error, fraction of ITEMS beyond RTOL, largest scale[, number of items]). The error of an element is measured
    against the largest reference magnitude of its own QUANTITY: the channel of a [C, H, W] image, the column of a [P, k]
    tensor (round 4; the altitude channel of out_color is 20-150x the RGB channels, tests/parity_cases.py quantity_scale).
    An item is what one moved blend decision touches as a whole: a pixel (all its channels) of an image, a row (one
    Gaussian) of a [P, k] tensor, an element otherwise."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    if b.numel() == 0:
        return (0.0, 0.0, 0.0, 0) if with_items else (0.0, 0.0, 0.0)
    if b.ndim == 3 and b.shape[1] == 1 and b.shape[0] > 1:  # [P, 1, k] (f_dc and its gradient): per-Gaussian rows, not an image
        a, b = a.reshape(b.shape[0], -1), b.reshape(b.shape[0], -1)
    if b.ndim == 3:
        scale = b.abs().amax(dim=(1, 2), keepdim=True)
    elif b.ndim == 2 and b.shape[1] > 1:
        scale = b.abs().amax(dim=0, keepdim=True)
    else:
        scale = b.abs().max().reshape(())
    scale = scale.clamp_min(1e-30)
    err = (a - b).abs() / scale
    bad = err > RTOL
    if b.ndim == 3:
        bad = bad.any(dim=0)
    elif b.ndim == 2 and b.shape[1] > 1:
        bad = bad.any(dim=1)
    out = (float(err.max()), float(bad.double().mean()), float(scale.max()))
    return out + (int(bad.numel()),) if with_items else out


def assert_close(a, b, what, rtol=RTOL, allow_flips=True, flip_floor=0, flip_rtol=FLIP_RTOL):
    """flip_floor: number of threshold-flip items (pixels / Gaussians) tolerated regardless of tensor size (used where the
    two sides evaluate the activations with different exp implementations, so opacities differ by an ulp).
    flip_rtol: bound on the size of such an outlier, relative to its quantity's scale."""
    assert tuple(a.shape) == tuple(b.shape), f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    mx, frac, scale, n = closeness(a, b, with_items=True)
    if allow_flips:
        ok = (mx <= rtol) or (frac <= max(FLIP_FRAC, (flip_floor + 0.5) / max(n, 1)) and mx <= flip_rtol)
    else:
        ok = mx <= rtol
    assert ok, f"{what}: max err {mx:.3e} (largest scale {scale:.3e}), fraction of pixels / rows beyond {RTOL:g}: {frac:.3e}"
    return mx


def run_case(case, device, rasterizer_mod, settings_cls):
    """Runs one golden-style case through a GaussianRasterizer implementation; returns outputs + grads."""
    t = lambda k: torch.from_numpy(case[k]).to(device)
    H, W = int(case["H"]), int(case["W"])
    P = case["means3D"].shape[0]
    leaf = lambda k: t(k).clone().requires_grad_(True)
    means3D, opac, colors = leaf("means3D"), leaf("opacities"), leaf("colors")
    scales = rotations = cov = None
    if "cov3D_precomp" in case:
        cov = leaf("cov3D_precomp")
    else:
        scales, rotations = leaf("scales"), leaf("rotations")
    means2D = torch.zeros(P, 3, device=device, requires_grad=True)
    vm = leaf("viewmatrix")
    rs = settings_cls(
        image_height=H, image_width=W, tanfovx=math.tan(0.5), tanfovy=math.tan(0.5), bg=t("bg"),
        scale_modifier=float(case.get("scale_modifier", 1.0)), viewmatrix=vm, projmatrix=vm.detach(), sh_degree=0, campos=torch.zeros(3, device=device), prefiltered=False,
        debug=False, antialiasing=bool(case["antialiasing"]))
    color, radii, invd = rasterizer_mod(rs)(
        means3D=means3D, means2D=means2D, opacities=opac, shs=None, colors_precomp=colors, scales=scales,
        rotations=rotations, cov3D_precomp=cov)
    loss = (color * t("dL_dcolor")).sum()
    if "dL_dinvdepth" in case:
        loss = loss + (invd * t("dL_dinvdepth")).sum()
    out = dict(out_color=color.detach(), out_radii=radii, out_invdepth=invd.detach())
    nr = getattr(color.grad_fn, "num_rendered", None)
    if nr is not None:
        out["_num_rendered"] = int(nr)  # the library's opaque token: tells which kernels ran (eogs_rast_path_info)
        out["_num_rendered_exact"] = int(getattr(color.grad_fn, "num_rendered_exact", nr))  # this forward's own counts
    if P:
        loss.backward()
        out.update(g_means3D=means3D.grad, g_means2D=means2D.grad, g_opacities=opac.grad, g_colors=colors.grad,
                   g_viewmatrix=vm.grad)
        if cov is None:
            out.update(g_scales=scales.grad, g_rotations=rotations.grad)
        else:
            out.update(g_cov3D_precomp=cov.grad)
    return out


# ---- raw-parameter (fused activations) helpers: SURVEY.md §8 row f1 ----
SH_C0 = 0.28209479177387814


def raw_params_from_scene(scene, seed=0):
    """Raw model parameters whose activations reproduce `scene` (up to fp32 rounding): log-scales, unnormalised
    quaternions, opacity logits, DC spherical-harmonic colours [P,1,3]; plus an alt_affine whose offset differs from
    the view matrix's (the reference's `affine` vs `world_view_transform + last_row`)."""
    g = torch.Generator().manual_seed(77 + seed)
    dev = scene["means3D"].device
    P = scene["means3D"].shape[0]
    op = scene["opacities"].double().cpu()
    raw = dict(
        xyz=scene["means3D"].clone(),
        log_scaling=torch.log(scene["scales"]),
        raw_rotation=scene["rotations"] * (0.5 + 1.5 * torch.rand(P, 1, generator=g)).to(dev),
        opacity_logit=torch.log(op / (1 - op)).float().to(dev),
        f_dc=((scene["colors"][:, :3] - 0.5) / SH_C0).reshape(P, 1, 3).contiguous(),
    )
    alt = scene["viewmatrix"][:, 2].clone()
    alt[3] += 0.125
    return raw, alt.contiguous()


def run_raw(raw, alt_affine, scene, H, W, antialiasing, fused, dL_dinvdepth=None):
    """fused=True: eogs2_amd.fused.rasterize_raw; fused=False: the reference's PyTorch ops
    (gaussian_model.py:41-52, renderer.py:91-96) followed by GaussianRasterizer. Returns outputs and raw-parameter grads."""
    from eogs2_amd.fused import rasterize_raw
    from eogs2_amd.rasterizer import GaussianRasterizer
    from eogs2_amd.synthetic import settings_for

    leaves = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    vm = scene["viewmatrix"].clone().requires_grad_(True)
    rs = settings_for(dict(scene, viewmatrix=vm), H, W, antialiasing=antialiasing)
    rs = rs._replace(projmatrix=vm.detach())
    P = raw["xyz"].shape[0]
    means2D = torch.zeros(P, 3, device=vm.device, requires_grad=True)
    if fused:
        color, radii, invd = rasterize_raw(leaves["xyz"], means2D, leaves["f_dc"], leaves["opacity_logit"],
                                           leaves["log_scaling"], leaves["raw_rotation"], alt_affine, rs)
    else:
        rgb = (leaves["f_dc"] * SH_C0 + 0.5).squeeze(1)
        altitude = (leaves["xyz"] @ alt_affine[:3] + alt_affine[3]).unsqueeze(-1)
        colors = torch.cat([rgb, altitude, torch.ones_like(altitude)], dim=-1)
        color, radii, invd = GaussianRasterizer(rs)(
            means3D=leaves["xyz"], means2D=means2D, opacities=torch.sigmoid(leaves["opacity_logit"]),
            colors_precomp=colors, scales=torch.exp(leaves["log_scaling"]),
            rotations=torch.nn.functional.normalize(leaves["raw_rotation"]))
    loss = (color * scene["dL_dcolor"]).sum()
    if dL_dinvdepth is not None:
        loss = loss + (invd * dL_dinvdepth).sum()
    loss.backward()
    out = dict(out_color=color.detach(), out_radii=radii, out_invdepth=invd.detach(), g_means2D=means2D.grad,
               g_viewmatrix=vm.grad)
    out.update({"g_" + k: v.grad for k, v in leaves.items()})
    return out


def render_unfused(cam, pc, pipe, bg):
    """The unfused counterpart of `eogs2_amd.render.render` for tests: activations and the [rgb, altitude, 1] features as
    PyTorch ops (what gaussian_model.py:109-137 / renderer.py:88-96 compute), then the drop-in GaussianRasterizer."""
    import math

    from eogs2_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

    vm = cam.world_view_transform.clone()
    vm[3] = vm[3] + cam.last_row
    rs = GaussianRasterizationSettings(int(cam.image_height), int(cam.image_width), math.tan(0.5), math.tan(0.5), bg, 1.0,
                                       vm, vm, 0, cam.camera_center, False, pipe.debug, pipe.antialiasing)
    vsp = torch.zeros_like(pc._xyz, requires_grad=True)
    alt = cam.ECEF_to_UVA(pc._xyz)[:, 2:3]
    feats = torch.cat([pc._features_dc.squeeze(1) * SH_C0 + 0.5, alt, torch.ones_like(alt)], dim=1)
    img, radii, _ = GaussianRasterizer(rs)(means3D=pc._xyz, means2D=vsp, colors_precomp=feats, opacities=pc.get_opacity,
                                           scales=pc.get_scaling, rotations=pc.get_rotation)
    return {"render": img, "viewspace_points": vsp, "radii": radii, "visibility_filter": torch.nonzero(radii > 0)}

