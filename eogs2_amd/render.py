"""`render()` with the reference's signature and result dict, on top of this package's rasterizer.

Mirrors src/gaussiansplatting/gaussian_renderer/renderer.py:14-140. The camera and model are duck-typed exactly as
the reference uses them there (`viewpoint_camera.{FoVx,FoVy,world_view_transform,full_proj_transform,
learn_wv_only_lastparam,last_row,image_height,image_width,camera_center,affine|ECEF_to_UVA}`,
`pc.{get_xyz,_xyz,_features_dc,_opacity,_scaling,_rotation,get_opacity,get_scaling,get_rotation,get_covariance,
active_sh_degree,get_exposure_from_name}`, `pipe.{debug,antialiasing,compute_cov3D_python,require_radii}`).

`fused=True` (default) takes the raw-parameter path of `eogs2_amd.fused` whenever the reference would build the
standard inputs (no `override_color`, no `compute_cov3D_python`); otherwise — or with `fused=False` — it performs
the reference's PyTorch ops and calls `GaussianRasterizer`, line for line equivalent to the reference.
"""
import math

import torch

from .fused import rasterize_raw
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

SH_C0 = 0.28209479177387814  # utils/sh_utils.py:25


def SH2RGB(sh):  # utils/sh_utils.py:125-126
    return sh * SH_C0 + 0.5


def _alt_affine(cam):
    # ECEF_to_UVA(xyz)[..., 2] = xyz @ affine[:3, 2] + affine[3, 2]   (scene/cameras/affine_cameras.py:432-438)
    return cam.affine[:, 2].detach().to(torch.float32).contiguous()


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, separate_sh=False, override_color=None,
           use_trained_exp=False, fused=True):
    xyz = pc.get_xyz
    # zero tensor that receives the gradient of the 2D (screen-space) means (renderer.py:30-40)
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass

    viewmatrix = viewpoint_camera.world_view_transform
    projmatrix = viewpoint_camera.full_proj_transform
    if getattr(viewpoint_camera, "learn_wv_only_lastparam", False):
        # renderer.py:47-53: the learnable offset is added to the last row of both matrices
        viewmatrix = viewmatrix.clone()
        projmatrix = projmatrix.clone()
        viewmatrix[-1, :] = viewmatrix[-1, :] + viewpoint_camera.last_row
        projmatrix[-1, :] = projmatrix[-1, :] + viewpoint_camera.last_row
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5),
        tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewmatrix,
        projmatrix=projmatrix,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=pipe.debug,
        antialiasing=pipe.antialiasing,
    )

    if fused and override_color is None and not pipe.compute_cov3D_python and hasattr(viewpoint_camera, "affine"):
        assert bg_color.shape[-1] == 5
        rendered_image, radii, invdepths = rasterize_raw(
            pc._xyz, screenspace_points, pc._features_dc, pc._opacity, pc._scaling, pc._rotation,
            _alt_affine(viewpoint_camera), raster_settings,
        )
    else:
        scales = rotations = cov3D_precomp = None
        if pipe.compute_cov3D_python:
            cov3D_precomp = pc.get_covariance(scaling_modifier)
        else:
            scales = pc.get_scaling
            rotations = pc.get_rotation
        if override_color is None:
            rgb = SH2RGB(pc._features_dc).squeeze(1)
            altitude = viewpoint_camera.ECEF_to_UVA(pc._xyz)[..., 2].unsqueeze(-1)
            colors_precomp = torch.cat([rgb, altitude, torch.ones_like(altitude)], dim=-1)
        else:
            colors_precomp = override_color
        assert bg_color.shape[-1] == colors_precomp.shape[-1]
        rendered_image, radii, invdepths = GaussianRasterizer(raster_settings=raster_settings)(
            means3D=xyz, means2D=screenspace_points, shs=None, colors_precomp=colors_precomp,
            opacities=pc.get_opacity, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp,
        )

    if use_trained_exp:  # renderer.py:112-120
        exposure = pc.get_exposure_from_name(viewpoint_camera.image_name)
        rendered_image = (
            torch.matmul(rendered_image.permute(1, 2, 0), exposure[:3, :3]).permute(2, 0, 1)
            + exposure[:3, 3, None, None]
        )
    out = {"render": rendered_image, "viewspace_points": screenspace_points}
    if pipe.require_radii:
        out["visibility_filter"] = (radii > 0).nonzero()
        out["radii"] = radii
    return out


__all__ = ["render", "SH2RGB"]
