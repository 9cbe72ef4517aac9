"""TEST INFRASTRUCTURE — float64 restatement of the reference's per-pixel image chain (checker for include/eogs_shade.h).

Follows, statement by statement (paths under /root/reference/src/gaussiansplatting/):
  render_pipeline       scene/cameras/affine_cameras.py:303-348 with ShadowMap.forward :33-40
  suncamera_l           loss/shadow.py:37-51
  randomcam_l           loss/main_loss.py:151-164 with _forward :83-96
  translucentshadows_l  loss/shadow.py:13-17
Pinned by vectors produced by the reference's own modules (tests/golden/make_golden_shade.py -> tests/golden/shade_*.npz,
checked in tests/test_shade_oracle.py). Only tests/, __graft_entry__.smoke() and bench.py's baseline leg may import this.
"""
import torch


def render_pipeline(raw_render, sun_altitude_diff, M, inshadow):
    """-> (cc, shaded, shadow); M is [3,4] = colour-correction weight | bias (or exposure[0], or identity)."""
    cc = torch.einsum("ck,khw->chw", M[:, :3], raw_render) + M[:, 3].view(3, 1, 1)  # :311-323
    if sun_altitude_diff is None:
        return cc, cc, None
    shadow = torch.exp(0.4 * sun_altitude_diff.clip(max=0.0))  # :38
    shaded = shadow * cc + (1 - shadow) * inshadow.view(3, 1, 1) * cc  # :334
    return cc, shaded, shadow


def _masked(alt_diff, rgb_diff, mask):
    mask = mask.detach()
    if not mask.any():
        z = alt_diff.sum() * 0.0
        return z, z
    n = mask.sum()
    return (alt_diff.abs() * mask).sum() / n, (rgb_diff.abs() * mask).sum() / n


def suncamera_l(raw_render, sun_rgb_sample, sun_altitude_diff, sun_uv):
    mask = (sun_altitude_diff > -1e-2) * (sun_uv.abs() < 1).all(-1)  # shadow.py:39
    return _masked(sun_altitude_diff, raw_render - sun_rgb_sample, mask)


def randomcam_l(new_altitude_diff, rgb_render, new_rgb_sample, new_uv):
    mask = (new_altitude_diff.abs() < 0.30) * (new_uv.abs() < 1).all(-1)  # main_loss.py:153-155
    return _masked(new_altitude_diff, rgb_render - new_rgb_sample, mask)


def translucentshadows_l(shadowmap):
    a = shadowmap
    b = shadowmap.clip(0.05, 0.95)
    return -(a * torch.log2(b) + (1 - a) * torch.log2(1 - b)).mean()  # shadow.py:14-16
