"""Shared pieces of the GPU parity tests: the seeded case tables, the comparison with its tolerances, and the
ATTRIBUTION of every out-of-tolerance element to a blend / skip decision that sits on a threshold.

North_star's tolerance is 1e-4 relative. The reference algorithm has three discontinuities per (pixel, Gaussian) pair
(DGR/cuda_rasterizer/forward.cu:366-382): skip if power > 0, skip if alpha < 1/255, stop the pixel BEFORE the Gaussian that
would make T < 1e-4. Two fp32 implementations (libm expf on the host, v_exp_f32 on the GPU, different but equally valid
association of the exponent) can land on different sides of a threshold for a pair that sits within a few ulp of it; the
pixel then differs by up to alpha T |c| and so does every gradient fed by that pixel. Such elements are allowed ONLY when
attributed: `threshold_map` recomputes, on the CPU from the case's inputs, which pixels hold a pair within a stated number
of ulp of a threshold, and an out-of-tolerance element must belong to such a pixel (images) or to a Gaussian whose tile
rect contains such a pixel (per-Gaussian gradients). Everything else must meet the tolerance outright.
"""
import numpy as np
import torch

from util import GRAD_RTOL, RTOL

ULP = 2.0 ** -23

SEEDED = [
    # P, H, W, seed, opacity, scale_mult, aa, depth_grad
    (5000, 160, 208, 10, "init", 2.0, False, False),
    (5000, 160, 208, 11, "trained", 2.0, True, True),
    (3000, 64, 64, 12, 0.7, 8.0, False, False),      # long lists (>256/tile), early termination
    (20000, 256, 256, 13, "trained", 1.0, False, False),
    (777, 33, 47, 14, "trained", 4.0, False, True),    # ragged image, ragged P
    (400, 200, 168, 15, "trained", 14.0, False, False),  # rects > 64 internal tiles: row-span listing path
    (1, 64, 64, 16, 0.9, 30.0, False, False),            # one Gaussian covering every tile
    (4000, 517, 1021, 17, "trained", 1.0, True, True),   # odd sizes: partial 8x8 and 16x16 tiles on both edges
    (1500, 96, 96, 18, 0.003, 3.0, False, False),        # opacity < 1/255: visible radii, nothing ever blended
]


def seeded_case(P, H, W, seed, opacity, scale_mult, aa, dgrad):
    from eogs2_amd.synthetic import make_scene

    sc = make_scene(P, H, W, seed=seed, opacity=opacity, scale_mult=scale_mult)
    case = {k: v.numpy() for k, v in sc.items()}
    case.update(H=H, W=W, antialiasing=aa)
    if dgrad:
        case["dL_dinvdepth"] = (torch.randn(1, H, W, generator=torch.Generator().manual_seed(seed)) / (H * W) * 100).numpy()
    return case, f"seed{seed}"


def sweep_case(seed):
    """Random small configurations (sizes, opacity law, footprint, anisotropy, rotation, antialiasing, inverse-depth
    gradient): every listing kind (mask / row spans / whole rect), partial tiles and long lists get hit by chance."""
    from eogs2_amd.synthetic import make_scene

    g = torch.Generator().manual_seed(seed)
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    P, H, W = r(1, 3000), r(9, 200), r(9, 260)
    opacity = ["init", "trained", 0.3, 0.02, 0.95][r(0, 4)]
    scale_mult = [0.5, 1.0, 2.5, 6.0, 15.0][r(0, 4)]
    # log-normal axis ratios up to ~e^(3*1.2): beyond that the fp32 covariance backward of the reference algorithm is
    # itself ill-conditioned (HIP and oracle then sit equally far, tens of per cent, from a float64 evaluation)
    aniso = [0.0, 0.3, 0.7, 1.2][r(0, 3)]
    aa, dgrad = bool(r(0, 1)), bool(r(0, 1))
    sc = make_scene(P, H, W, seed=seed, opacity=opacity, scale_mult=scale_mult, anisotropy=aniso)
    if r(0, 1):
        q = torch.randn(P, 4, generator=g)
        sc["rotations"] = (q / q.norm(dim=1, keepdim=True)).contiguous()
    case = {k: v.numpy() for k, v in sc.items()}
    case.update(H=H, W=W, antialiasing=aa)
    if dgrad:
        case["dL_dinvdepth"] = (torch.randn(1, H, W, generator=g) / (H * W) * 100).numpy()
    stress = scale_mult >= 6.0 or aniso >= 0.7  # cancellation-heavy gradient sums: tests/util.py GRAD_RTOL
    return case, ("seed15" if stress else f"sweep{seed}")


class ThresholdMap:
    """Which pixels hold a (pixel, Gaussian) pair within a stated number of ulp of a blend / skip / stop threshold,
    evaluated per 16 x 16 tile ON DEMAND (at 1 M Gaussians / 1024^2 a full map costs minutes; the tiles behind a handful
    of out-of-tolerance elements cost seconds).

    near pixel: some candidate Gaussian of the pixel (its 16-px tile lies in the Gaussian's tile rect, power <= 0, the
    pixel not yet terminated) has |alpha 255 - 1| <= (k_alpha[0] + k_alpha[1] |power|) ulp — the exponent carries a
    relative rounding error of a few ulp, which the exponential turns into |power| times that — or a transmittance test
    with |T' / 1e-4 - 1| <= (k_T[0] + k_T[1] n) ulp, n = Gaussians blended so far at the pixel (T' is a product of n
    rounded factors). Evaluated in fp32 with the reference's own formulas (oracle/torch_dense.py restates them).
    touched Gaussian: its tile rect contains a near pixel (a flipped pixel changes the gradients of the flipped
    Gaussian and of everything blended behind it there)."""

    def __init__(self, case, k_alpha=(16.0, 8.0), k_T=(16.0, 4.0)):
        from oracle.torch_dense import TILE, cov3d_full, cov6_to_full, project

        self.k_alpha, self.k_T, self.TILE = k_alpha, k_T, TILE
        t = lambda k: torch.from_numpy(np.asarray(case[k]))
        self.H, self.W = int(case["H"]), int(case["W"])
        H, W = self.H, self.W
        means3D, opac0, vm = t("means3D"), t("opacities").reshape(-1), t("viewmatrix")
        self.P = means3D.shape[0]
        self.tiles = {}  # (ty, tx) -> bool [th, tw]
        self.gx, self.gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
        if self.P == 0:
            return
        self.pix, depth = project(means3D, vm, H, W)
        Sigma = cov6_to_full(t("cov3D_precomp")) if "cov3D_precomp" in case else cov3d_full(t("scales"), t("rotations"), 1.0)
        s = torch.tensor([W / 2.0, H / 2.0])
        T = vm[:3, :2].t() * s[:, None]
        cov2 = T @ Sigma @ T.t()
        a0, b0, c0 = cov2[:, 0, 0], cov2[:, 0, 1], cov2[:, 1, 1]
        det0 = a0 * c0 - b0 * b0
        a, c, b = a0 + 0.3, c0 + 0.3, b0
        det = a * c - b * b
        self.opac = opac0 * torch.sqrt(torch.clamp(det0 / det, min=0.000025)) if bool(case["antialiasing"]) else opac0
        self.ca, self.cb, self.cc = c / det, -b / det, a / det
        mid = 0.5 * (a + c)
        root = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
        radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(mid + root, mid - root)))
        ri = radius.to(torch.int32).float()
        tdiv = lambda v: torch.trunc(v / TILE).to(torch.int64)
        self.x0, self.y0 = tdiv(self.pix[:, 0] - ri).clamp(0, self.gx), tdiv(self.pix[:, 1] - ri).clamp(0, self.gy)
        self.x1 = tdiv(self.pix[:, 0] + ri + (TILE - 1)).clamp(0, self.gx)
        self.y1 = tdiv(self.pix[:, 1] + ri + (TILE - 1)).clamp(0, self.gy)
        self.visible = (det != 0) & ((self.x1 - self.x0) * (self.y1 - self.y0) > 0)
        order = torch.sort(depth, stable=True).indices
        self.order = order[self.visible[order]]
        # depth-ordered copies of the rects: one boolean pass per tile
        self.ox0, self.ox1, self.oy0, self.oy1 = (v[self.order] for v in (self.x0, self.x1, self.y0, self.y1))

    def tile(self, ty, tx):
        """near mask of the 16 x 16 tile (ty, tx) (clipped at the image border)."""
        key = (int(ty), int(tx))
        if key in self.tiles:
            return self.tiles[key]
        TILE, H, W = self.TILE, self.H, self.W
        ys, xs = torch.arange(ty * TILE, min((ty + 1) * TILE, H)), torch.arange(tx * TILE, min((tx + 1) * TILE, W))
        out = torch.zeros(ys.numel(), xs.numel(), dtype=torch.bool)
        ids = self.order[(self.ox0 <= tx) & (self.ox1 > tx) & (self.oy0 <= ty) & (self.oy1 > ty)] if self.P else None
        if ids is not None and ids.numel():
            PY, PX = torch.meshgrid(ys, xs, indexing="ij")
            pxf, pyf = PX.reshape(-1).float(), PY.reshape(-1).float()
            dx = self.pix[ids, 0][:, None] - pxf[None, :]
            dy = self.pix[ids, 1][:, None] - pyf[None, :]
            power = -0.5 * (self.ca[ids][:, None] * dx * dx + self.cc[ids][:, None] * dy * dy) - self.cb[ids][:, None] * dx * dy
            alpha = torch.clamp(self.opac[ids][:, None] * torch.exp(power), max=0.99)
            cand = power <= 0  # (every listed Gaussian's rect contains the whole tile)
            valid = cand & (alpha >= 1.0 / 255.0)
            av = torch.where(valid, alpha, torch.zeros_like(alpha))
            T_after = torch.cumprod(1 - av, dim=0)
            nblend = torch.cumsum(valid.to(torch.int32), dim=0).float()
            # a pixel keeps evaluating candidates until it stops; widen "not yet stopped" by the T margin itself
            tolT = (self.k_T[0] + self.k_T[1] * nblend) * ULP
            alive = torch.cumsum((valid & (T_after < 0.0001 * (1 - tolT))).to(torch.int32), dim=0) == 0
            tolA = (self.k_alpha[0] + self.k_alpha[1] * power.abs()) * ULP
            near_a = cand & alive & ((alpha * 255.0 - 1.0).abs() <= tolA)
            near_t = valid & alive & ((T_after / 0.0001 - 1.0).abs() <= tolT)
            out = (near_a | near_t).any(dim=0).view(ys.numel(), xs.numel())
        self.tiles[key] = out
        return out

    def pixels_near(self, ys, xs):
        """bool per (y, x) pair: is that pixel a near-threshold pixel?"""
        res = torch.zeros(len(ys), dtype=torch.bool)
        for i, (y, x) in enumerate(zip(ys.tolist(), xs.tolist())):
            res[i] = self.tile(y // self.TILE, x // self.TILE)[y % self.TILE, x % self.TILE]
        return res

    def gaussians_touched(self, idx):
        """bool per Gaussian index: does its tile rect contain a near-threshold pixel?"""
        res = torch.zeros(len(idx), dtype=torch.bool)
        for i, g in enumerate(idx.tolist()):
            if not bool(self.visible[g]):
                continue
            hit = False
            for ty in range(int(self.y0[g]), int(self.y1[g])):
                for tx in range(int(self.x0[g]), int(self.x1[g])):
                    if bool(self.tile(ty, tx).any()):
                        hit = True
                        break
                if hit:
                    break
            res[i] = hit
        return res


def threshold_map(case):
    """(near[H,W] bool, touched[P] bool), everything evaluated (small cases / diagnostics)."""
    tm = ThresholdMap(case)
    near = torch.zeros(tm.H, tm.W, dtype=torch.bool)
    for ty in range(tm.gy):
        for tx in range(tm.gx):
            m = tm.tile(ty, tx)
            near[ty * tm.TILE:ty * tm.TILE + m.shape[0], tx * tm.TILE:tx * tm.TILE + m.shape[1]] = m
    touched = tm.gaussians_touched(torch.arange(tm.P)) if tm.P else torch.zeros(0, dtype=torch.bool)
    return near, touched


def oracle_run(case, backend=None):
    """The case through the same host wrapper over the CPU oracle (checker library; tests only)."""
    import oracle
    from util import run_case

    from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib

    hip = _lib.get
    _lib.get = backend or oracle.abi
    try:
        ref = run_case(case, torch.device("cpu"), GaussianRasterizer, GaussianRasterizationSettings)
    finally:
        _lib.get = hip
    return {k: v.cpu().numpy() for k, v in ref.items() if not k.startswith("_")}


SENS_ULPS, SENS_DRAWS, SENS_FACTOR = 4.0, 4, 4.0


def sensitivity_map(case, base=None):
    """{output name: max over SENS_DRAWS of |oracle(inputs (1 + SENS_ULPS ulp randn)) - oracle(inputs)|}.

    Backward-error view of an fp32 evaluation: it returns the exact result for inputs perturbed by a few ulp. Where the
    ORACLE's own output moves by more than the tolerance under such a perturbation (cancelling sums in the covariance
    backward of strongly anisotropic Gaussians, backward.cu:239-394; or a pair sitting on a threshold), two correct
    fp32 implementations cannot be expected to agree to the tolerance, and the movement bounds by how much."""
    import oracle

    base = base if base is not None else oracle_run(case)
    out = {k: np.zeros_like(np.asarray(v), dtype=np.float64) for k, v in base.items() if k != "out_radii"}
    # the summation order of the per-Gaussian sums: fp32 atomicAdds in the reference (backward.cu:598-640, order changes run
    # to run), fp32 per-tile records here, double in the restatement. One evaluation with fp32 accumulation in pixel order
    # (an order the reference itself can produce) measures how far that moves each output.
    lib = oracle.abi().cdll
    lib.eogs_oracle_accum_float(1)
    try:
        res = oracle_run(case)
    finally:
        lib.eogs_oracle_accum_float(0)
    for k in out:
        out[k] = np.maximum(out[k], np.abs(np.asarray(res[k], dtype=np.float64) - np.asarray(base[k], dtype=np.float64)))
    # the HIP path's own formulation of dL/dalpha — front to back, the sum behind a Gaussian taken as (rendered total -
    # running prefix) instead of the reference's back-to-front recursion (render.hip; algebraically identical) — evaluated
    # by the ORACLE in fp32: where a Gaussian's contribution is orders below the pixel's total (image-sized opaque Gaussians
    # stacked hundreds deep) the subtraction carries an absolute error of an ulp of the TOTAL. tools/suffix_probe.py.
    lib.eogs_oracle_suffix_by_subtraction(1)
    try:
        res = oracle_run(case)
    finally:
        lib.eogs_oracle_suffix_by_subtraction(0)
    for k in out:
        out[k] = np.maximum(out[k], np.abs(np.asarray(res[k], dtype=np.float64) - np.asarray(base[k], dtype=np.float64)))
    # rounding alone: the same restatement built with fused multiply-adds (oracle/Makefile). Expressions of the reference
    # that cancel (`denom - c_xx * c_yy` = -c_xy^2 computed from two rounded products, backward.cu:239-251) are numerically
    # unstable rather than ill-conditioned: an input perturbation moves both products together and does not show it, a
    # second valid rounding does.
    fma = oracle.abi_fma()
    if fma is not None:
        res = oracle_run(case, backend=lambda: fma)
        for k in out:
            out[k] = np.maximum(out[k], np.abs(np.asarray(res[k], dtype=np.float64) - np.asarray(base[k], dtype=np.float64)))
    for draw in range(SENS_DRAWS):
        g = np.random.default_rng(1000 + draw)
        pert = dict(case)
        # the Gaussians' parameters (each moves all of a Gaussian's pixel terms together) AND the upstream gradient (moves
        # every pixel's term independently: what the rounding of the individual terms of a cancelling per-Gaussian sum does)
        for k in ("means3D", "scales", "rotations", "opacities", "colors", "cov3D_precomp", "dL_dcolor", "dL_dinvdepth"):
            if k in case:
                v = np.asarray(case[k])
                pert[k] = (v * (1.0 + SENS_ULPS * ULP * g.standard_normal(v.shape))).astype(np.float32)
        res = oracle_run(pert)
        for k in out:
            out[k] = np.maximum(out[k], np.abs(np.asarray(res[k], dtype=np.float64) - np.asarray(base[k], dtype=np.float64)))
    return out


class Attribution:
    """Lazily evaluated threshold tiles and sensitivity map of one case (only computed when some element is out of
    tolerance, the threshold tiles only where such elements sit, and the sensitivity map only when the thresholds do not
    explain them). `cache`: a file prefix for the sensitivity map, which depends on the case alone, so processes that
    replay the same case (tests/path_child.py) share it."""

    def __init__(self, case, ref=None, cache=None):
        self.case = case
        self.ref = ref
        self.cache = cache
        self._tm = None
        self._sens = None

    def thresholds(self):
        if self._tm is None:
            self._tm = ThresholdMap(self.case)
        return self._tm

    def sensitivity(self, key):
        import os
        import time

        if self._sens is None:
            f = self.cache + ".sens.npz" if self.cache else None
            if f and os.path.exists(f):
                z = np.load(f)
                self._sens = {k: z[k] for k in z.files}
            else:
                t0 = time.perf_counter()
                self._sens = sensitivity_map(self.case, self.ref)
                print(f"sensitivity map: {time.perf_counter() - t0:.1f} s")
                if f:
                    os.makedirs(os.path.dirname(f), exist_ok=True)
                    np.savez(f, **self._sens)
        return torch.from_numpy(self._sens[key])


def check_close(got, ref, what, rtol, attribution=None, kind=None, flip_rtol=5e-2, key=None, sens_rtol=1e-1, flip_abs=0.0):
    """|got - ref| <= rtol * max|ref| elementwise; elements beyond it must be attributed — to a threshold pixel (kind
    "image": the pixel is a near-threshold pixel; kind "gaussian": the row's Gaussian is touched by one; bounded by
    flip_rtol), or, failing that, to the oracle's own sensitivity: the element moves by at least err / SENS_FACTOR when
    the inputs are perturbed by SENS_ULPS ulp (bounded by sens_rtol).
    Returns (max error, number of attributed outliers)."""
    a = torch.as_tensor(got, dtype=torch.float64).cpu()
    b = torch.as_tensor(np.asarray(ref), dtype=torch.float64)
    assert tuple(a.shape) == tuple(b.shape), f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    if b.numel() == 0:
        return 0.0, 0
    scale = max(float(b.abs().max()), 1e-30)
    err = (a - b).abs() / scale
    bad = err > rtol
    nbad = int(bad.sum())
    if nbad == 0:
        return float(err.max()), 0
    assert attribution is not None and kind is not None, f"{what}: max err {float(err.max()):.3e} (x scale {scale:.3e}), {nbad} elements beyond {rtol:g}"
    tm = attribution.thresholds()
    ok = torch.zeros_like(bad)
    where = torch.nonzero(bad)
    if kind == "image":  # [..., H, W]: the element's pixel must be a near-threshold pixel
        ys, xs = where[:, -2], where[:, -1]
        pix = torch.unique(torch.stack([ys, xs], 1), dim=0)
        hit = tm.pixels_near(pix[:, 0], pix[:, 1])
        lut = {(int(y), int(x)): bool(h) for (y, x), h in zip(pix.tolist(), hit.tolist())}
        ok[tuple(where.t())] = torch.tensor([lut[(int(y), int(x))] for y, x in zip(ys.tolist(), xs.tolist())])
    else:                # [P, ...]: the row's Gaussian must be touched by a near-threshold pixel
        rows = torch.unique(where[:, 0])
        hit = tm.gaussians_touched(rows)
        lut = dict(zip(rows.tolist(), hit.tolist()))
        ok[tuple(where.t())] = torch.tensor([lut[int(r)] for r in where[:, 0].tolist()])
    unexplained = bad & ~ok
    if bool(unexplained.any()) and key is not None:
        delta = attribution.sensitivity(key).reshape(err.shape) / scale
        sens_ok = err <= SENS_FACTOR * delta + rtol
        n_sens = int((unexplained & sens_ok).sum())
        if n_sens:
            assert float(err[unexplained & sens_ok].max()) <= sens_rtol, (
                f"{what}: sensitivity-attributed error {float(err[unexplained & sens_ok].max()):.3e} exceeds {sens_rtol:g}")
            print(f"{what}: {n_sens} elements attributed to ill-conditioning (oracle moves by >= err/{SENS_FACTOR:g} under "
                  f"{SENS_ULPS:g}-ulp input perturbation), max err {float(err[unexplained & sens_ok].max()):.3e}")
        unexplained = unexplained & ~sens_ok
    assert not bool(unexplained.any()), (
        f"{what}: {int(unexplained.sum())} of {nbad} out-of-tolerance elements are neither at a threshold pixel nor "
        f"ill-conditioned (max unexplained err {float(err[unexplained].max()):.3e}, rtol {rtol:g}, scale {scale:.3e})")
    flipped = bad & ok
    if bool(flipped.any()):
        # a flipped blend decision moves a pixel by one contribution alpha T |c| <= |c|_max / 255 (flip_abs, absolute): in a
        # faint image (few low-opacity Gaussians) that can be several per cent of the image's own scale
        lim = max(flip_rtol, flip_abs / scale)
        assert float(err[flipped].max()) <= lim, f"{what}: attributed flip of {float(err[flipped].max()):.3e} exceeds {lim:.3g}"
    return float(err.max()), nbad


IMAGE_KEYS = ("out_color", "out_invdepth")


def compare(out, ref, name, case, stats=None, cache=None):
    """HIP outputs + gradients of one case against the oracle's. radii bit-exact; images and per-Gaussian gradients to
    RTOL (GRAD_RTOL for the stress fixtures), out-of-tolerance elements only where attributed to a threshold pixel."""
    assert np.array_equal(out["out_radii"].cpu().numpy(), np.asarray(ref["out_radii"])), f"{name}: radii differ"
    att = Attribution(case, ref={k: v for k, v in ref.items() if k.startswith(("out_", "g_"))}, cache=cache)
    flips = 0
    for k, v in out.items():
        if k == "out_radii" or k.startswith("_"):
            continue
        r = np.asarray(ref[k])
        if k == "g_viewmatrix":
            # a cancelling sum over all Gaussians of terms that are each within tolerance: the meaningful scale is the
            # sum of magnitudes |means3D|^T @ |dL_dmeans2D| (and sum |dL_dmeans2D| for the last row), not |sum|
            g2 = torch.as_tensor(np.asarray(ref["g_means2D"])).abs().double()
            m = torch.as_tensor(np.asarray(case["means3D"])).abs().double()
            rt = torch.as_tensor(r).double()
            scale = max(float((m.t() @ g2).max()), float(g2.sum(0).max()), float(rt.abs().max()), 1e-30)
            err = float((v.cpu().double() - rt).abs().max()) / scale
            # the [:3,:2] block comes from the covariance backward, the worst-conditioned part; a threshold flip
            # anywhere in the image moves these 16 global sums: x4 when the case has flips
            lim = GRAD_RTOL.get(name, RTOL) * (4 if (flips or name.startswith(("sweep", "seed15"))) else 1)
            assert err <= lim, f"{name}:{k}: {err:.3e} of the magnitude sum (limit {lim:g})"
            continue
        grad = k.startswith("g_")
        flip_abs = 0.0
        if k == "out_color":    # one blended / skipped Gaussian at alpha ~ 1/255: |c|_max / 255 (+ a few per cent)
            flip_abs = 1.05 / 255.0 * float(np.abs(np.asarray(case["colors"])).max()) if np.asarray(case["colors"]).size else 0.0
        elif k == "out_invdepth" and np.asarray(case["means3D"]).size:
            vm = np.asarray(case["viewmatrix"], dtype=np.float64)
            depth = 200.0 - (np.asarray(case["means3D"], dtype=np.float64) @ vm[:3, 2] + vm[3, 2])
            flip_abs = 1.05 / 255.0 / float(depth[depth > 0].min()) if (depth > 0).any() else 0.0
        _, n = check_close(v, r, f"{name}:{k}", GRAD_RTOL.get(name, RTOL) if grad else RTOL, att,
                           "gaussian" if grad else "image", flip_rtol=1e-1 if grad else 2e-2, key=k, flip_abs=flip_abs)
        flips += n
    if stats is not None:
        stats[name] = flips
    return flips
