"""Independent dense pure-PyTorch alpha-blend renderer with autograd.

TEST INFRASTRUCTURE ONLY (same rule as rast_oracle.c): used by tests/ to pin the
C oracle on forward and on every gradient, and by bench.py as the reported
"pure-PyTorch CPU alpha-blend" baseline. Never imported by the
product path.

It is written from the reference's *rules*, not from its kernel structure: no
tiles-as-threadblocks, no sorted key list, no hand-written backward — tensors
[Gaussians x pixels] and autograd. The discrete rules it reproduces (DGR/ =
src/gaussiansplatting/submodules/diff-gaussian-rasterization/):

* projection uva = xyz @ vm[:3,:3] + vm[3,:3]; pixel centre ((ndc+1)*S-1)/2 in double
  (DGR/cuda_rasterizer/auxiliary.h:40-43,70-78)
* cov2D = T Sigma T^T + 0.3 I, T = diag(W/2,H/2) A[0:2,:] (forward.cu:74-112,219-223)
* radius = ceil(3 sqrt(max eig)), eig via sqrt(max(0.1, mid^2-det)) (forward.cu:242-245)
* a Gaussian is a candidate for a pixel iff the pixel's 16x16 tile lies in its tile rect
  (auxiliary.h:45-55, rasterizer_impl.cu:91-109)
* order: depth = 200 - altitude ascending, ties by index (rasterizer_impl.cu:102-106,306-311)
* skip if power > 0; alpha = min(0.99, o G) with straight-through gradient (backward.cu:624 has no
  clamp mask); skip if alpha < 1/255; stop BEFORE the Gaussian that would make T < 1e-4
  (forward.cu:366-382)
* out = C + T bg; invdepth = sum alpha T / depth (forward.cu:385-409)
"""
import math

import torch

TILE = 16
C = 5


def quat_to_rot(q):
    """Standard rotation matrix of (r,x,y,z), NOT normalised (forward.cu:126)."""
    r, x, y, z = q.unbind(-1)
    R = torch.stack(
        [
            1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y),
        ],
        dim=-1,
    )
    return R.view(*q.shape[:-1], 3, 3)


def cov3d_full(scales, rotations, scale_modifier):
    R = quat_to_rot(rotations)
    S = scales * scale_modifier
    M = R * S[..., None, :]  # R diag(S)
    return M @ M.transpose(-1, -2)


def cov6_to_full(c6):
    i = torch.tensor([[0, 1, 2], [1, 3, 4], [2, 4, 5]], device=c6.device)
    return c6[:, i]


def project(means3D, viewmatrix, H, W, means2D=None):
    uva = means3D @ viewmatrix[:3, :3] + viewmatrix[3, :3]
    ndc = uva[:, :2]
    if means2D is not None:
        ndc = ndc + means2D[:, :2]  # zero-valued leaf: its grad is dL/d(ndc), the reference's dL_dmeans2D
    size = torch.tensor([W, H], dtype=torch.float64, device=means3D.device)
    # narrowed to the working dtype like the reference (fp32); float64 inputs keep float64 (conditioning probes)
    pix = (((ndc.double() + 1.0) * size - 1.0) * 0.5).to(means3D.dtype)
    depth = (200.0 - uva[:, 2].double()).to(means3D.dtype)
    return pix, depth


def render_dense(
    means3D, opacities, colors, bg, viewmatrix, H, W,
    scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0,
    antialiasing=False, means2D=None, T_override=None, block=64, want_aux=False, crop=None,
):
    """Returns (color[5,H,W], radii[P] int32, invdepth[1,H,W]) (+ dict of aux tensors).
    crop=(y0, x0, h, w) (multiples of `block`... of 16 at least) renders only that window of the H x W image and
    returns images of the window's size."""
    dev = means3D.device
    P = means3D.shape[0]
    if P == 0:
        z = torch.zeros
        return z(C, H, W), z(0, dtype=torch.int32), z(1, H, W)
    opacities = opacities.reshape(P)
    pix, depth = project(means3D, viewmatrix, H, W, means2D)
    if cov3D_precomp is not None:
        Sigma = cov6_to_full(cov3D_precomp)
    else:
        Sigma = cov3d_full(scales, rotations, scale_modifier)
    if T_override is None:
        s = torch.tensor([W / 2.0, H / 2.0], dtype=means3D.dtype, device=dev)
        T = viewmatrix[:3, :2].t() * s[:, None]  # rows i: s_i * A[i,:]
    else:
        T = T_override
    cov2 = T @ Sigma @ T.t()  # [P,2,2]
    a0, b0, c0 = cov2[:, 0, 0], cov2[:, 0, 1], cov2[:, 1, 1]
    det0 = a0 * c0 - b0 * b0
    a, c, b = a0 + 0.3, c0 + 0.3, b0
    det = a * c - b * b
    opac = opacities
    if antialiasing:
        opac = opacities * torch.sqrt(torch.clamp(det0 / det, min=0.000025))
    conic_a, conic_b, conic_c = c / det, -b / det, a / det

    with torch.no_grad():
        mid = 0.5 * (a + c)
        root = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
        radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(mid + root, mid - root)))
        gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
        ri = radius.to(torch.int32).to(means3D.dtype)

        def tdiv(v):  # truncating float->int division as in getRect
            return torch.trunc(v / TILE).to(torch.int64)

        x0 = tdiv(pix[:, 0] - ri).clamp(0, gx)
        y0 = tdiv(pix[:, 1] - ri).clamp(0, gy)
        x1 = tdiv(pix[:, 0] + ri + (TILE - 1)).clamp(0, gx)
        y1 = tdiv(pix[:, 1] + ri + (TILE - 1)).clamp(0, gy)
        visible = (det != 0) & ((x1 - x0) * (y1 - y0) > 0)
        radii = torch.where(visible, radius.to(torch.int32), torch.zeros_like(radius, dtype=torch.int32))
        if bool((visible & (depth < 0)).any()):
            raise RuntimeError("Point is too high: altitude > 200")
        order = torch.sort(depth, stable=True).indices  # ties keep index order
        order = order[visible[order]]

    cy0, cx0, ch, cw = (0, 0, H, W) if crop is None else crop
    assert cy0 % TILE == 0 and cx0 % TILE == 0
    out_color = torch.zeros(C, ch, cw, device=dev, dtype=means3D.dtype)
    out_invd = torch.zeros(1, ch, cw, device=dev, dtype=means3D.dtype)
    out_T = torch.ones(ch, cw, device=dev, dtype=means3D.dtype)
    assert block % TILE == 0
    for by in range(cy0, min(cy0 + ch, H), block):
        for bx in range(cx0, min(cx0 + cw, W), block):
            ty0, ty1 = by // TILE, min((by + block + TILE - 1) // TILE, gy)
            tx0, tx1 = bx // TILE, min((bx + block + TILE - 1) // TILE, gx)
            with torch.no_grad():
                cand = (x0[order] < tx1) & (x1[order] > tx0) & (y0[order] < ty1) & (y1[order] > ty0)
                ids = order[cand]
            ys = torch.arange(by, min(by + block, H, cy0 + ch), device=dev)
            xs = torch.arange(bx, min(bx + block, W, cx0 + cw), device=dev)
            oy, ox = ys - cy0, xs - cx0
            if ids.numel() == 0:
                out_color[:, oy[:, None], ox[None, :]] = bg[:, None, None].expand(C, ys.numel(), xs.numel())
                continue
            PY, PX = torch.meshgrid(ys, xs, indexing="ij")
            pxf, pyf = PX.reshape(-1).to(means3D.dtype), PY.reshape(-1).to(means3D.dtype)
            ptx, pty = (PX.reshape(-1) // TILE), (PY.reshape(-1) // TILE)
            dx = pix[ids, 0][:, None] - pxf[None, :]
            dy = pix[ids, 1][:, None] - pyf[None, :]
            power = -0.5 * (conic_a[ids][:, None] * dx * dx + conic_c[ids][:, None] * dy * dy) - conic_b[ids][:, None] * dx * dy
            G = torch.exp(power)
            raw = opac[ids][:, None] * G
            alpha = raw + (torch.clamp(raw, max=0.99) - raw).detach()
            with torch.no_grad():
                in_rect = (
                    (ptx[None, :] >= x0[ids][:, None]) & (ptx[None, :] < x1[ids][:, None])
                    & (pty[None, :] >= y0[ids][:, None]) & (pty[None, :] < y1[ids][:, None])
                )
                valid = in_rect & (power <= 0) & (alpha >= 1.0 / 255.0)
                av = torch.where(valid, alpha, torch.zeros_like(alpha))
                T_after = torch.cumprod(1 - av, dim=0)
                stop = valid & (T_after < 0.0001)
                include = valid & (torch.cumsum(stop.to(torch.int32), dim=0) == 0)
            a_inc = torch.where(include, alpha, torch.zeros_like(alpha))
            T_incl = torch.cumprod(1 - a_inc, dim=0)
            T_before = torch.cat([torch.ones_like(T_incl[:1]), T_incl[:-1]], dim=0)
            w = a_inc * T_before  # [N, npix]
            T_final = T_incl[-1]
            col = colors[ids].t() @ w + bg[:, None] * T_final[None, :]
            invd = (1.0 / depth[ids]).detach()[None, :] @ w  # depth is a constant in the reference backward (backward.cu:305-307 commented out)
            shp = (ys.numel(), xs.numel())
            out_color[:, oy[:, None], ox[None, :]] = col.view(C, *shp)
            out_invd[:, oy[:, None], ox[None, :]] = invd.view(1, *shp)
            out_T[oy[:, None], ox[None, :]] = T_final.detach().view(*shp)
    if want_aux:
        return out_color, radii, out_invd, {"final_T": out_T, "pix": pix, "depth": depth,
                                           "conic": torch.stack([conic_a, conic_b, conic_c], -1), "opac": opac,
                                           "num_rendered": int(((x1 - x0) * (y1 - y0))[visible].sum())}
    return out_color, radii, out_invd


def blend_forward_cpu(*args, **kw):
    """Forward only, no autograd graph (CPU-baseline timing entry)."""
    with torch.no_grad():
        return render_dense(*args, **kw)
