"""Fused front end behind the reference's `render()` call.

The reference's `render` (src/gaussiansplatting/gaussian_renderer/renderer.py:14-140) stays reference Python
(SURVEY.md §2 row 4). What this module adds is ONE thing: when that function would build the standard rasterizer
inputs from the model's raw parameters — activations, DC-SH colour, altitude channel, constant channel
(`renderer.py:88-96`, `scene/gaussian_model.py:109-137`) — the same result comes from a single call into the
raw-parameter path of the HIP library (`eogs2_amd.fused.rasterize_raw`, `EOGS_FLAG_RAW_PARAMS`), and none of those
PyTorch ops (nor their autograd replay) run.

`render(...)` has the reference's signature and result dict so that a caller can switch by changing one import.
Inputs the fused path does not cover (`override_color`, `compute_cov3D_python`, a camera without `affine`) are NOT
re-implemented here: they go to `fallback`, the caller's own `render` (INTEGRATION.md shows the two-line hook), or
raise if none was given.
"""
import math

import torch

from .fused import rasterize_raw
from .rasterizer import GaussianRasterizationSettings


def fusable(viewpoint_camera, pipe, override_color=None):
    """True when `renderer.py:80-96` would take its default branch: colours from `_features_dc`, covariance from
    scale / rotation, altitude from the camera's affine map."""
    return (override_color is None and not getattr(pipe, "compute_cov3D_python", False)
            and hasattr(viewpoint_camera, "affine"))


def _camera_matrices(cam):
    """The two 4x4 matrices handed to the rasterizer; with `learn_wv_only_lastparam` the camera's learnable `last_row`
    offsets their last row (`renderer.py:47-53`), out of place so that autograd reaches `last_row`."""
    vm, pm = cam.world_view_transform, cam.full_proj_transform
    if getattr(cam, "learn_wv_only_lastparam", False):
        shift = torch.zeros_like(vm)
        shift[-1] = cam.last_row
        vm, pm = vm + shift, pm + shift
    return vm, pm


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, separate_sh=False, override_color=None,
           use_trained_exp=False, fallback=None, altitude_only=False):
    """`altitude_only=True` (not in the reference's signature; `eogs2_amd.fused.rasterize_raw`): "render" is the altitude image
    [1,H,W] alone — for a render whose RGB nobody consumes (the sun camera before `iterstart_L_sun_resample`)."""
    if altitude_only and (use_trained_exp or not fusable(viewpoint_camera, pipe, override_color)):
        raise NotImplementedError("altitude_only renders exist on the raw-parameter path only, without per-image exposure")
    if not fusable(viewpoint_camera, pipe, override_color):
        if fallback is None:
            raise NotImplementedError(
                "eogs2_amd.render.render covers the raw-parameter path only (no override_color, no "
                "compute_cov3D_python, camera with .affine); pass fallback=<the reference's render> for the rest")
        return fallback(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, separate_sh, override_color,
                        use_trained_exp)
    if bg_color.shape[-1] != 5:
        raise ValueError("background must have 5 channels (rgb, altitude, constant)")
    vm, pm = _camera_matrices(viewpoint_camera)
    settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(0.5 * viewpoint_camera.FoVx), tanfovy=math.tan(0.5 * viewpoint_camera.FoVy), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=vm, projmatrix=pm, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug, antialiasing=pipe.antialiasing)
    # the leaf whose .grad the densification statistics read (gaussian_model.py:719-723); a leaf keeps its .grad
    viewspace_points = torch.zeros_like(pc._xyz, requires_grad=True)
    # altitude channel = xyz @ affine[:3, 2] + affine[3, 2]   (scene/cameras/affine_cameras.py:432-438)
    alt_affine = viewpoint_camera.affine[:, 2].detach().to(torch.float32).contiguous()
    # (the reference's render() drops the rasterizer's inverse-depth image, renderer.py:101,126: it is not rendered here)
    image, radii, _ = rasterize_raw(pc._xyz, viewspace_points, pc._features_dc, pc._opacity, pc._scaling, pc._rotation,
                                    alt_affine, settings, altitude_only=altitude_only, invdepth=False)
    if use_trained_exp:  # per-image 3x4 exposure on the rgb planes (renderer.py:112-120)
        e = pc.get_exposure_from_name(viewpoint_camera.image_name)
        image = torch.einsum("chw,cd->dhw", image, e[:3, :3]) + e[:3, 3].reshape(3, 1, 1)
    out = {"render": image, "viewspace_points": viewspace_points}
    if pipe.require_radii:
        out["radii"] = radii
        # renderer.py:128-130 returns the INDICES (`nonzero`, which waits for the device to learn how many there are). While
        # a HIP graph is being recorded nothing may wait: the boolean mask is returned instead — `t[visibility_filter]`
        # reads and writes the same elements either way (train_pan.py:679-686).
        capturing = radii.is_cuda and torch.cuda.is_current_stream_capturing()
        out["visibility_filter"] = (radii > 0) if capturing else torch.nonzero(radii > 0)
    return out


__all__ = ["render", "fusable"]
