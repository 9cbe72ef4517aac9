"""CPU oracle of the virtual-camera resample — TEST INFRASTRUCTURE ONLY (never imported by eogs2_amd/).

The reference's own three statements (src/gaussiansplatting/gaussian_renderer/renderer_cc_shadow.py:32-50), in float64,
on top of torch.nn.functional.grid_sample — the third-party routine the reference itself calls (PyTorch 2.x ATen
grid_sampler_2d, bilinear, zeros padding, align_corners=True). Gradients via autograd.
"""
import torch


def resample(virtual_render, cam2virt, rendered_uva, n_keep=4):
    vr, M, uva = virtual_render.double(), cam2virt.double(), rendered_uva.double()
    virtual_uv = torch.einsum("...ij,...j->...i", M, uva)[..., :2]
    s = torch.nn.functional.grid_sample(vr.unsqueeze(0), virtual_uv.unsqueeze(0), align_corners=True).squeeze(0)
    rgb, alt = s[:3], s[3]
    alt = torch.where((virtual_uv.abs() > 1).any(-1), torch.full_like(alt, -100.0), alt)  # `alt[mask] = -100`
    return torch.cat([rgb, alt[None]], 0)[:n_keep], virtual_uv
