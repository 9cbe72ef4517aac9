"""TEST INFRASTRUCTURE — restatement of the reference's TSDF integration (checker for include/eogs_tsdf.h).

Follows src/gaussiansplatting/tsdf.py statement by statement: sample_sdf :325-368 (with _world_to_view / _view_to_world
:233-241), integrate :459-498, update_tsdf :500-520. PARITY UNPINNED against reference outputs: tsdf.py imports `iio`
and `omegaconf` at module level (absent here) and the reference holds no TSDF fixtures; the statements below use the same
torch ops (F.linear, F.grid_sample, linalg.inv/norm, index gather/scatter) in the dtype handed in.
Only tests/ and bench.py's comparison leg may import this.
"""
import torch
import torch.nn.functional as F


def sample_sdf(pts_world_coords, coef, intercept, model_scale, altitude_img, weight_img):
    pts = pts_world_coords / model_scale
    view = F.linear(pts, coef, intercept)
    features = torch.cat([altitude_img, weight_img], dim=1)
    sampled = F.grid_sample(features, view[None, :, None, :2], mode="bilinear", align_corners=True).squeeze()
    altitude_values, weights = sampled[0, :], sampled[1, :]
    valid_mask = (view[:, :2].abs() <= 1.0).all(dim=1)
    view_new = view.clone()
    view_new[:, 2] = altitude_values
    Ainv = torch.linalg.inv(coef)
    world_new = F.linear(view_new, Ainv, -(Ainv @ intercept))
    distances = torch.linalg.norm(world_new - pts, dim=1)
    distances = distances * torch.sign(view[:, 2] - altitude_values)
    return distances * model_scale, valid_mask, weights


def integrate(tsdf_vol, weight_vol, axes, coef, intercept, model_scale, trunc_margin, altitude_img, weight_img):
    """Returns new (tsdf_vol, weight_vol); inputs untouched."""
    world = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(-1, 3)
    sdf, mask, weights = sample_sdf(world, coef, intercept, model_scale, altitude_img, weight_img)
    mask = mask & (sdf >= -trunc_margin)
    tsdf_value = torch.minimum(torch.ones_like(sdf), sdf / trunc_margin)[mask]
    t, w = tsdf_vol.clone().reshape(-1), weight_vol.clone().reshape(-1)
    w_old, t_old, obs = w[mask], t[mask], weights[mask]
    w_new = w_old + obs
    t[mask] = (w_old * t_old + obs * tsdf_value) / w_new
    w[mask] = w_new
    return t.reshape(tsdf_vol.shape), w.reshape(weight_vol.shape)
