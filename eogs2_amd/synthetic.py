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


def make_scene(P, H, W, seed=0, opacity="init", device="cpu", scale_mult=1.0, anisotropy=0.3):
    """Returns a dict of fp32 tensors on `device` shaped like the rasterizer's inputs."""
    g = torch.Generator().manual_seed(seed)
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
