"""Synthetic EOGS-shaped scenes for benches, smoke and parity tests.

Real scenes (data.zip) are not available, so inputs follow the reference's own
initialisation statistics:

* positions uniform in the normalised scene box (GS/scene/dataset_readers/dataset_affine.py:277-281):
  x,y in [-0.9,0.9], z in [-0.05,0.15]
* isotropic scale = sqrt(mean squared distance to the 3 nearest neighbours)
  (GS/scene/gaussian_model.py:179-186); evaluated in closed form for a uniform density
  (E[r_k^2] = Gamma(k+2/3)/Gamma(k) (3/(4 pi rho))^(2/3), k=1..3) times a log-normal jitter
* rotation (1,0,0,0) + small perturbation, renormalised (gaussian_model.py:53)
* opacity 0.01 ("init", gs_config/train.yaml:55) or sigmoid(N(0,2)) ("trained")
* camera = the Nadir affine coef=[[0,1,0],[1,0,0],[0,0,350]], intercept 0
  (scripts/dataset_creation/to_affine.py:244-249) plus the shear A[0:2,2] ~ N(0,0.1) of
  sample_random_camera (GS/scene/cameras/affine_cameras.py:403-411); the torch viewmatrix holds the
  TRANSPOSED affine (affine_cameras.py:151-157) and projmatrix is the same tensor (:186-187)
* colors_precomp = [rgb, altitude, 1] (GS/gaussian_renderer/renderer.py:88-95),
  bg = rand(5), bg[3] = altitude_min, bg[4] = 0 (GS/train_pan.py:272-277)
* dL/dcolor ~ N(0,1)/(H W)

`kind="surface"` (round 6) is the shape a TRAINED scene has, which the statistics above are not: densify / prune
(GS/scene/gaussian_model.py:466-717) leave the Gaussians on the scene's surface — a terrain with buildings, 1-2 % of the box
thick — as flat disks along it (one axis 0.05-0.2 x the others), with in-plane sizes spread over orders of magnitude
(log-normal, sigma 1: split keeps the small ones small, flat ground keeps a few splats tens of pixels wide) and opacities
that are either nearly transparent (about to be pruned, gs_config/train.yaml:55 resets to 0.01) or nearly opaque.
"""
import math

import torch

ALT_SCALE = 350.0
_E_R2 = (0.9027452929509336 + 1.5045754882515558 + 2.0061006510020744) / 3.0  # mean_k Gamma(k+2/3)/Gamma(k)


def make_camera(H, W, seed=0, shear_std=0.1, device="cpu"):
    g = torch.Generator().manual_seed(1000 + seed)
    A = torch.tensor([[0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, ALT_SCALE]])
    A[0:2, 2] = torch.randn(2, generator=g) * shear_std
    b = torch.zeros(3)
    vm = torch.zeros(4, 4)
    vm[:3, :3] = A.t()
    vm[3, :3] = b
    vm[3, 3] = 1.0
    return vm.to(device)


def _height_field(x, y, g):
    """Terrain + flat-roofed buildings inside the scene box's altitude range; returns (z, dz/dx, dz/dy) of the smooth part
    (roofs and ground between the buildings' walls are what a nadir view sees)."""
    z = 0.02 + 0.03 * (torch.sin(3.1 * x + 1.0) * torch.cos(2.3 * y) + 1.0) * 0.5
    zx = 0.015 * 3.1 * torch.cos(3.1 * x + 1.0) * torch.cos(2.3 * y)
    zy = -0.015 * 2.3 * torch.sin(3.1 * x + 1.0) * torch.sin(2.3 * y)
    nb = 80
    cx, cy = 1.7 * torch.rand(nb, generator=g) - 0.85, 1.7 * torch.rand(nb, generator=g) - 0.85
    hw, hh = 0.02 + 0.08 * torch.rand(nb, generator=g), 0.02 + 0.08 * torch.rand(nb, generator=g)
    top = 0.02 + 0.07 * torch.rand(nb, generator=g)
    for k in range(nb):  # (later buildings override earlier ones: one roof height per point)
        inside = ((x - cx[k]).abs() < hw[k]) & ((y - cy[k]).abs() < hh[k])
        z = torch.where(inside, z.new_full((), 0.05) + top[k], z)
        zx = torch.where(inside, torch.zeros_like(zx), zx)
        zy = torch.where(inside, torch.zeros_like(zy), zy)
    return z, zx, zy


def _qmul(a, b):
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], dim=-1)


def _surface_scene(P, H, W, seed, g, scale_mult):
    x, y = 1.8 * torch.rand(P, generator=g) - 0.9, 1.8 * torch.rand(P, generator=g) - 0.9
    z, zx, zy = _height_field(x, y, g)
    z = z + 0.003 * torch.randn(P, generator=g)  # 1-2 % of the box's altitude range thick
    xyz = torch.stack([x, y, z.clamp(-0.05, 0.15)], dim=1)
    # disks along the surface: local z axis -> the surface normal (the shortest rotation), then a random turn about it.
    # The Gaussian's axes are the COLUMNS of the standard rotation matrix of (r, x, y, z) (forward.cu:126-150: Sigma = R S^2 R^T)
    nrm = torch.stack([-zx, -zy, torch.ones_like(zx)], dim=1)
    nrm = nrm / nrm.norm(dim=1, keepdim=True)
    q_align = torch.stack([1.0 + nrm[:, 2], -nrm[:, 1], nrm[:, 0], torch.zeros(P)], dim=1)
    q_align = q_align / q_align.norm(dim=1, keepdim=True)
    th = 2.0 * math.pi * torch.rand(P, generator=g)
    q_turn = torch.stack([torch.cos(0.5 * th), torch.zeros(P), torch.zeros(P), torch.sin(0.5 * th)], dim=1)
    q = _qmul(q_align, q_turn)
    q = q / q.norm(dim=1, keepdim=True)
    # in-plane size: log-normal around 1.2 x the mean spacing of P points on the 1.8 x 1.8 ground, sigma 1 (clipped at 4 sigma);
    # the two in-plane axes differ by a further factor e^(0.3 N); the normal axis is 0.05-0.2 x their mean
    d = math.sqrt(3.24 / max(P, 1))
    base = 1.2 * d * scale_mult * torch.exp(torch.randn(P, generator=g).clamp(-4.0, 4.0))
    s1 = base * torch.exp(0.3 * torch.randn(P, generator=g))
    s2 = base * torch.exp(0.3 * torch.randn(P, generator=g))
    s3 = 0.5 * (s1 + s2) * (0.05 + 0.15 * torch.rand(P, generator=g))
    scales = torch.stack([s1, s2, s3], dim=1)
    # opacity: 35 % nearly transparent (around 0.02), 65 % nearly opaque (around 0.9)
    low = torch.rand(P, generator=g) < 0.35
    logit = torch.where(low, -3.9 + 0.5 * torch.randn(P, generator=g), 2.2 + 0.7 * torch.randn(P, generator=g))
    op = torch.sigmoid(logit).unsqueeze(1)
    return xyz, scales, q, op


def make_scene(P, H, W, seed=0, opacity="init", device="cpu", scale_mult=1.0, anisotropy=0.3, kind="volume"):
    """Returns a dict of fp32 tensors on `device` shaped like the rasterizer's inputs. kind="surface" (or opacity="surface",
    for callers that pass the opacity law through: bench.py --opacity surface): the trained-scene shape (module docstring;
    opacities and anisotropy are then the generator's own)."""
    g = torch.Generator().manual_seed(seed)
    if kind == "surface" or opacity == "surface":
        xyz, scales, q, op = _surface_scene(P, H, W, seed, g, scale_mult)
        return _finish_scene(P, H, W, seed, g, device, xyz, scales, q, op)
    assert kind == "volume", kind
    lo = torch.tensor([-0.9, -0.9, -0.05])
    hi = torch.tensor([0.9, 0.9, 0.15])
    xyz = lo + (hi - lo) * torch.rand(P, 3, generator=g)
    rho = max(P, 1) / float(torch.prod(hi - lo))
    s0 = math.sqrt(_E_R2 * (3.0 / (4.0 * math.pi * rho)) ** (2.0 / 3.0))
    scales = s0 * scale_mult * torch.exp(anisotropy * torch.randn(P, 3, generator=g))
    q = torch.tensor([1.0, 0.0, 0.0, 0.0]) + 0.3 * torch.randn(P, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    if opacity == "init":
        op = torch.full((P, 1), 0.01)
    elif opacity == "trained":
        op = torch.sigmoid(2.0 * torch.randn(P, 1, generator=g))
    else:
        op = torch.full((P, 1), float(opacity))
    return _finish_scene(P, H, W, seed, g, device, xyz, scales, q, op)


def _finish_scene(P, H, W, seed, g, device, xyz, scales, q, op):
    rgb = torch.rand(P, 3, generator=g)
    vm = make_camera(H, W, seed)
    alt = (xyz @ vm[:3, :3] + vm[3, :3])[:, 2:3]
    colors = torch.cat([rgb, alt, torch.ones(P, 1)], dim=1)
    bg = torch.rand(5, generator=g)
    bg[3] = alt.min() if P else 0.0
    bg[4] = 0.0
    dL_dcolor = torch.randn(5, H, W, generator=g) / (H * W)
    out = dict(means3D=xyz, scales=scales, rotations=q, opacities=op, colors=colors, bg=bg,
               viewmatrix=vm, dL_dcolor=dL_dcolor)
    return {k: v.to(device=device, dtype=torch.float32).contiguous() for k, v in out.items()}


def settings_for(scene, H, W, antialiasing=False, debug=False):
    from .rasterizer import GaussianRasterizationSettings

    vm = scene["viewmatrix"]
    return GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(0.5), tanfovy=math.tan(0.5),
        bg=scene["bg"], scale_modifier=1.0, viewmatrix=vm, projmatrix=vm, sh_degree=0,
        campos=torch.zeros(3, device=vm.device), prefiltered=False, debug=debug, antialiasing=antialiasing,
    )
