"""Opt-in fused front end: rasterize straight from the model's raw parameters.

The reference runs, before every render, a chain of small PyTorch kernels over all Gaussians —
`exp(_scaling)`, `sigmoid(_opacity)`, `normalize(_rotation)` (scene/gaussian_model.py:41,49,52,109-137),
`SH2RGB(_features_dc)`, `ECEF_to_UVA(_xyz)[..., 2]`, `ones_like`, `cat` (gaussian_renderer/renderer.py:91-96) —
and autograd replays their backward after the rasterizer's. `rasterize_raw` hands the raw tensors to the C-ABI
with `EOGS_FLAG_RAW_PARAMS` (include/eogs_rast.h): the per-Gaussian HIP kernels apply the activations while
loading and chain their derivatives while storing, so the ~17 elementwise launches (and their [P,·]
intermediates) per render disappear. Results equal the unfused composition to fp32 rounding
(tests/test_gpu_parity.py::test_fused_matches_unfused).

Same no-fallback rule as `rasterizer.py`: arithmetic only in the HIP library.
"""
import torch

from .rasterizer import NUM_CHANNELS, _run_backward, _run_forward, last_exact_token


class _RasterizeRaw(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, means2D, f_dc, opacity_logit, log_scaling, raw_rotation, viewmat, alt_affine, raster_settings,
                altitude_only=False, invdepth=True):
        rs = raster_settings
        P = xyz.shape[0]
        if f_dc.numel() != P * 3:
            raise RuntimeError("f_dc must have dimensions (num_points, 3) or (num_points, 1, 3)")
        num_rendered, color, radii, invdepths, geom, binning, img = _run_forward(
            rs, viewmat, xyz, f_dc.reshape(P, 3), opacity_logit, log_scaling, raw_rotation, None,
            alt_affine=alt_affine, raw=True, alt_only=altitude_only, want_invdepth=bool(invdepth),
        )
        ctx.altitude_only = bool(altitude_only)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered  # layout of the workspaces (may be a capacity: see _run_forward)
        ctx.num_rendered_exact = last_exact_token(xyz.device) if num_rendered else 0
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii)
        ctx.save_for_backward(xyz, f_dc, opacity_logit, log_scaling, raw_rotation, alt_affine, radii,
                              geom, binning, img, color, invdepths)
        return color, radii, invdepths

    @staticmethod
    def backward(ctx, grad_out_color, _, grad_out_depth):
        rs = ctx.raster_settings
        (xyz, f_dc, opacity_logit, log_scaling, raw_rotation, alt_affine, radii,
         geom, binning, img, color, invdepths) = ctx.saved_tensors
        P = xyz.shape[0]
        want_vm = ctx.needs_input_grad[6]
        if P == 0:
            z = torch.zeros_like
            return (z(xyz), z(xyz), z(f_dc), z(opacity_logit), z(log_scaling), z(raw_rotation),
                    torch.zeros_like(rs.viewmatrix) if want_vm else None, None, None, None, None)
        d_means2D, d_fdc, d_logit, d_xyz, _cov, d_logscale, d_rawrot, grad_viewmatrix = _run_backward(
            rs, ctx.num_rendered, grad_out_color, grad_out_depth, xyz, None, opacity_logit,
            log_scaling, raw_rotation, None, radii, geom, binning, img, color, invdepths, want_vm, alt_affine=alt_affine, raw=True,
            alt_only=ctx.altitude_only,
        )
        return (d_xyz, d_means2D, d_fdc.view(f_dc.shape), d_logit.view(opacity_logit.shape), d_logscale, d_rawrot,
                grad_viewmatrix, None, None, None, None)


def rasterize_raw(xyz, means2D, f_dc, opacity_logit, log_scaling, raw_rotation, alt_affine, raster_settings, altitude_only=False,
                  invdepth=True):
    """(color[5,H,W], radii[P], invdepths[1,H,W]) from raw parameters.

    `invdepth=False`: the inverse-depth image is not rendered (third result None; the C-ABI's out_invdepth = NULL) — for callers
    that drop it, as the reference's own `render()` does (gaussian_renderer/renderer.py:101,126): one multiply-add less per
    evaluated (pixel, Gaussian) in the forward, the colour image and every gradient the same bits.

    `altitude_only=True` (EOGS_FLAG_ALT_ONLY, include/eogs_rast.h): only the altitude feature is rendered — `color` is then
    [1,H,W], equal to channel 3 of the full render, and the third result is None (there is no inverse-depth image to read
    or to differentiate) — for renders that are consumed through
    their altitude alone: the reference's 2H x 2W sun-camera render before `iterstart_L_sun_resample` (train_pan.py:305-324,
    gs_config/train.yaml:123). Forward and backward then blend one channel instead of five.

    Equivalent to the reference's
        GaussianRasterizer(raster_settings)(means3D=xyz, means2D=means2D, shs=None,
            colors_precomp=cat([SH2RGB(f_dc), (xyz @ A[:3,2] + A[3,2])[:, None], 1]),
            opacities=sigmoid(opacity_logit), scales=exp(log_scaling), rotations=normalize(raw_rotation))
    with `alt_affine = camera.affine[:, 2]` (4 floats; no gradient flows into it — the reference's `affine` is a buffer).
    `means2D` only receives the screen-space gradient, as in the reference.
    """
    if alt_affine.numel() != 4:
        raise RuntimeError("alt_affine must have 4 elements (camera.affine[:, 2])")
    return _RasterizeRaw.apply(xyz, means2D, f_dc, opacity_logit, log_scaling, raw_rotation,
                               raster_settings.viewmatrix, alt_affine, raster_settings, bool(altitude_only), bool(invdepth))


__all__ = ["rasterize_raw", "NUM_CHANNELS"]
