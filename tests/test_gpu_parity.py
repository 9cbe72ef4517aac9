"""GPU (MI355X): the HIP path, called through the C-ABI via the drop-in Python API, against
  (1) the committed golden vectors (reference wrapper over the oracle),
  (2) the CPU oracle on fresh seeded inputs,
  (3) size-independent properties at the BASELINE size (1M Gaussians / 1024^2).
Tolerance: north_star's 1e-4 rel, see tests/util.py.
"""
import math

import numpy as np
import pytest
import torch

from util import GOLDEN, assert_close, load_golden, run_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from eogs2_amd import _lib

    assert _lib.get().backend == "hip-gfx950"  # the native library is what runs
    return torch.device("cuda:0")


def test_wave64_primitives_selftest(dev):
    import ctypes

    from eogs2_amd import _lib

    abi = _lib.get()
    scratch = torch.zeros(4, dtype=torch.int32, device=dev)
    failed = ctypes.c_uint(99)
    abi.check(abi.selftest(ctypes.c_void_p(scratch.data_ptr()), ctypes.byref(failed),
                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert failed.value == 0, f"wave64 primitive self-test failed: mask {failed.value:#x}"


def _compare(out, ref, name, case):
    """tests/parity_cases.py: tolerance 1e-4 of the tensor scale; beyond it only elements attributed to a threshold pixel."""
    from parity_cases import compare

    return compare(out, ref, name, case)


@pytest.mark.parametrize("name", GOLDEN)
def test_hip_matches_golden(name, dev):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    case = load_golden(name)
    out = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    _compare(out, case, name, case)


from parity_cases import SEEDED, seeded_case, sweep_case  # noqa: E402


@pytest.mark.parametrize("row", SEEDED, ids=lambda r: "-".join(str(x) for x in r))
def test_hip_matches_oracle_seeded(row, dev, monkeypatch):
    import oracle
    from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib

    from parity_cases import oracle_cached

    case, name = seeded_case(*row)
    got = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    _compare(got, oracle_cached(f"seeded_{name}", case), name, case)  # checker: the same wrapper over the CPU oracle


def test_depth_ties_and_overlap_order(dev):
    """Blend order with many exactly-equal depths: ties must resolve by Gaussian index exactly as the reference's
    stable (tile | depth-bits) sort does. Checked through the image against the independent dense renderer, which
    orders by (depth, index) with a stable torch.sort. (num_rendered itself is internal: this library lists a
    Gaussian only in the 8x8 tiles where it can reach alpha >= 1/255, a subset of the reference's tile rect.)"""
    from eogs2_amd import GaussianRasterizer
    from eogs2_amd.synthetic import make_scene, settings_for
    from oracle.torch_dense import render_dense

    P, H, W = 8000, 208, 176  # (the dense CPU renderer is O(P x pixels): 45 s at 30000 / 320 x 272, 18 s at 12000 / 256 x 208)
    sc = make_scene(P, H, W, seed=3, opacity="trained", scale_mult=1.5, device=dev)
    z = sc["means3D"][:, 2].clone()
    z[::7] = z[1::7][: z[::7].shape[0]]          # exact duplicates, 1/7 of the Gaussians
    z = (z * 64).round() / 64                    # and a coarse altitude grid: thousands of ties
    sc["means3D"][:, 2] = z
    sc["colors"][:, 3] = (sc["means3D"] @ sc["viewmatrix"][:3, :3] + sc["viewmatrix"][3, :3])[:, 2]
    color, radii, invd = GaussianRasterizer(settings_for(sc, H, W))(
        sc["means3D"], torch.zeros(P, 3, device=dev), sc["opacities"], colors_precomp=sc["colors"],
        scales=sc["scales"], rotations=sc["rotations"])
    c = {k: v.cpu() for k, v in sc.items()}
    col2, r2, inv2 = render_dense(c["means3D"], c["opacities"], c["colors"], c["bg"], c["viewmatrix"], H, W,
                                  scales=c["scales"], rotations=c["rotations"], block=64)
    assert torch.equal(radii.cpu(), r2)
    assert_close(color, col2, "tie-order image")
    assert_close(invd, inv2, "tie-order invdepth")


@pytest.mark.parametrize("P,label", [(50_000, "8-item LDS path"), (130_000, "streaming path"), (2_000, "4-item LDS path")])
def test_long_block_lists_against_oracle(dev, P, label, exact_counts):
    """The three ways block_lists_kernel orders a block (csrc/binning.hip): up to 4096 entries per 32 x 32-px block in
    registers / LDS, the 8-item build for blocks of 2800 ... 6000 entries on average, and chunked streaming through the
    scratch ping-pong buffer beyond the LDS path. A 128 x 128 image has 16 blocks; the Gaussian count sets the block length.
    Every output and gradient against the oracle (the list order decides the blend order at every pixel)."""
    import oracle
    from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
    from eogs2_amd.synthetic import make_scene

    H = W = 128
    sc = make_scene(P, H, W, seed=41, opacity=0.05, scale_mult=0.6)
    case = {k: v.numpy() for k, v in sc.items()}
    case.update(H=H, W=W, antialiasing=False)
    got = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    entries = (got["_num_rendered"] >> 32) & 0x07FFFFFF
    per_block = entries / 16.0
    assert {"8-item LDS path": 2800 < per_block <= 6000, "streaming path": per_block > 6000,
            "4-item LDS path": per_block <= 2800}[label], per_block
    hip = _lib.get
    _lib.get = oracle.abi
    try:
        ref = run_case(case, torch.device("cpu"), GaussianRasterizer, GaussianRasterizationSettings)
    finally:
        _lib.get = hip
    _compare(got, {k: v.cpu().numpy() for k, v in ref.items() if not k.startswith('_')}, f"blocks-{P}", case)


def test_image_with_more_than_4096_blocks_takes_the_two_pass_sort(dev):
    """2304 x 2176 pixels = 72 x 68 blocks of 32 x 32: 13 bits of block id, beyond the one-pass counting sort (4096 blocks):
    two passes + the block-count kernel (csrc/binning.hip). Every output and gradient against the oracle."""
    import oracle
    from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
    from eogs2_amd.synthetic import make_scene

    H, W, P = 2176, 2304, 20000
    assert ((H + 31) // 32) * ((W + 31) // 32) > 4096
    # (small footprints: what is tested is the block id's 13th bit, and the CPU oracle walks 5 M pixels — with the footprints
    # 6000 Gaussians have at scale 2 it took 33 s of the suite's time, round 4)
    sc = make_scene(P, H, W, seed=43, opacity="trained", scale_mult=0.25)
    case = {k: v.numpy() for k, v in sc.items()}
    case.update(H=H, W=W, antialiasing=False)
    got = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    hip = _lib.get
    _lib.get = oracle.abi
    try:
        ref = run_case(case, torch.device("cpu"), GaussianRasterizer, GaussianRasterizationSettings)
    finally:
        _lib.get = hip
    _compare(got, {k: v.cpu().numpy() for k, v in ref.items() if not k.startswith('_')}, "two-pass", case)


def test_footprints_taller_than_64_block_rows(dev):
    """A 2304-pixel-high image has 72 rows of 32 x 32-px blocks: a footprint as tall as the image spans more block rows than a wave
    has lanes, so the passes that deal a footprint's block rows to the lanes of a wave — preprocess_fwd's cooperative count and
    expand_entries' cooperative walk (csrc/preprocess.hip, csrc/binning.hip: `MY += 64`) — go round twice. 600 ordinary Gaussians
    plus 12 thin ones as tall as the image (sigma 450-700 px along the rows, 2-5 px across, some rotated a few degrees: row-span
    listing), every output and gradient against the oracle."""
    from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer
    from eogs2_amd.synthetic import make_scene

    H, W, P, tall = 2304, 160, 600, 12
    assert (H + 31) // 32 > 64
    sc = make_scene(P, H, W, seed=47, opacity="trained", scale_mult=1.0)
    case = {k: v.numpy() for k, v in sc.items()}
    g = np.random.default_rng(47)
    idx = g.choice(P, tall, replace=False)
    # (synthetic.make_camera: image rows run along world x, columns along world y; a scale s is s * H / 2 or s * W / 2 pixels)
    case["scales"][idx] = np.stack([g.uniform(450, 700, tall) / (0.5 * H), g.uniform(2, 5, tall) / (0.5 * W),
                                    np.full(tall, 0.01)], axis=1).astype(np.float32)
    ang = np.deg2rad(g.uniform(-3, 3, tall))
    case["rotations"][idx] = np.stack([np.cos(ang / 2), np.zeros(tall), np.zeros(tall), np.sin(ang / 2)], axis=1).astype(np.float32)
    case["means3D"][idx, 0] = g.uniform(-0.2, 0.2, tall).astype(np.float32)
    case["opacities"][idx] = g.uniform(0.3, 0.8, (tall, 1)).astype(np.float32)
    case.update(H=H, W=W, antialiasing=False)
    got = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    radii = got["out_radii"].cpu().numpy()
    assert int((radii[idx] > 1000).sum()) == tall, radii[idx]  # they do span the image
    from parity_cases import oracle_run

    _compare(got, oracle_run(case), "tall", case)


def test_entry_sort_in_scratch_and_in_the_binning_workspace_agree(dev, monkeypatch, exact_counts):
    """The entry sort runs in the caller's scratch behind the count readback (ABI v5) or, without scratch / with more
    entries than its capacity, in the binning workspace after the readback: same kernels, same lists — outputs and
    gradients are bit-identical, and the token records where the sort ran."""
    from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer, rasterizer
    from eogs2_amd.synthetic import make_scene

    for P, scale_mult, want_sorted in ((20000, 1.0, 1), (300, 14.0, 0)):  # (0: six entries per Gaussian are not enough)
        sc = make_scene(P, 200, 168, seed=51, opacity="trained", scale_mult=scale_mult)
        case = {k: v.numpy() for k, v in sc.items()}
        case.update(H=200, W=168, antialiasing=False)
        a = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
        assert (a["_num_rendered"] >> 61) & 1 == want_sorted
        with monkeypatch.context() as m:
            m.setattr(rasterizer, "_scratch_for", lambda *args: None)
            b = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
        assert (b["_num_rendered"] >> 61) & 1 == 0
        assert (a["_num_rendered"] & 0x7FFFFFFF) == (b["_num_rendered"] & 0x7FFFFFFF)
        for k in a:
            if not k.startswith("_"):
                assert torch.equal(a[k], b[k]), k


def test_wide_altitude_range_uses_all_sort_passes(dev, monkeypatch):
    """Depth keys spanning several binades (200 - altitude from ~25 to ~305): the top byte of the key differs, so the
    fourth depth-sort pass must run (with EOGS-like altitudes it is skipped). Compared with the oracle."""
    import oracle
    from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
    from eogs2_amd.synthetic import make_scene

    P, H, W = 6000, 128, 160
    sc = make_scene(P, H, W, seed=31, opacity="trained", scale_mult=2.5)
    sc["means3D"][:, 2] = sc["means3D"][:, 2] * 4.0 - 0.1  # z in [-0.3, 0.5] -> altitude in [-105, 175]
    sc["colors"][:, 3] = (sc["means3D"] @ sc["viewmatrix"][:3, :3] + sc["viewmatrix"][3, :3])[:, 2]
    depth = 200.0 - sc["colors"][:, 3]
    assert float(depth.min()) < 64 and float(depth.max()) > 256
    case = {k: v.numpy() for k, v in sc.items()}
    case.update(H=H, W=W, antialiasing=False)
    got = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    hip = _lib.get()
    monkeypatch.setattr(_lib, "get", lambda: oracle.abi())
    ref = run_case(case, torch.device("cpu"), GaussianRasterizer, GaussianRasterizationSettings)
    monkeypatch.setattr(_lib, "get", lambda: hip)
    _compare(got, {k: v.cpu().numpy() for k, v in ref.items() if not k.startswith('_')}, "wide-altitude", case)


@pytest.mark.parametrize("opacity,aniso,scale_mult,aa", [("trained", 1.5, 8.0, False), (0.02, 1.2, 12.0, True),
                                                         ("trained", 2.0, 5.0, True)])
def test_row_span_listing_anisotropic(dev, monkeypatch, opacity, aniso, scale_mult, aa):
    """Large, thin, arbitrarily rotated Gaussians: footprints beyond 64 internal tiles are listed by per-row column
    spans (common.h row_span) — counted in preprocess, re-evaluated in expand. Low opacity makes the alpha >= 1/255
    ellipse much smaller than the 3-sigma rect; every result must still equal the oracle's (which walks the full rect)."""
    import oracle
    from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
    from eogs2_amd.synthetic import make_scene

    P, H, W = 500, 232, 280
    sc = make_scene(P, H, W, seed=41, opacity=opacity, scale_mult=scale_mult, anisotropy=aniso)
    g = torch.Generator().manual_seed(5)
    q = torch.randn(P, 4, generator=g)
    sc["rotations"] = (q / q.norm(dim=1, keepdim=True)).contiguous()
    case = {k: v.numpy() for k, v in sc.items()}
    case.update(H=H, W=W, antialiasing=aa)
    got = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    hip = _lib.get()
    monkeypatch.setattr(_lib, "get", lambda: oracle.abi())
    ref = run_case(case, torch.device("cpu"), GaussianRasterizer, GaussianRasterizationSettings)
    monkeypatch.setattr(_lib, "get", lambda: hip)
    _compare(got, {k: v.cpu().numpy() for k, v in ref.items() if not k.startswith('_')}, f"aniso{aniso}", case)


def _sweep_seeds():
    """24 seeds by default; EOGS_SWEEP_SEEDS=lo-hi[,lo-hi | seed ...] widens the sweep for an occasional long run."""
    import os

    seeds = []
    for part in os.environ.get("EOGS_SWEEP_SEEDS", "200-223").split(","):
        lo, _, hi = part.partition("-")
        seeds += list(range(int(lo), int(hi or lo) + 1))
    return seeds


@pytest.mark.parametrize("seed", _sweep_seeds())
def test_randomised_sweep_against_oracle(dev, monkeypatch, seed):
    """Random small configurations (tests/parity_cases.py sweep_case): every listing kind (mask / row spans / whole rect),
    partial tiles and long lists get hit by chance. Out-of-tolerance elements must be attributed to a threshold pixel."""
    import oracle
    from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib

    from parity_cases import oracle_cached

    case, name = sweep_case(seed)
    got = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    _compare(got, oracle_cached(f"sweep_{seed}", case), name, case)


def test_sun_camera_size_2048(dev):
    """2048 x 2048 (the reference's 2H x 2W sun-camera render, affine_cameras.py:366-367): 65,536 internal tiles,
    16-bit tile keys, two 8-bit tile-sort passes. Checked against size-independent properties and a crop computed by
    the dense renderer."""
    from eogs2_amd import GaussianRasterizer
    from eogs2_amd.synthetic import make_scene, settings_for
    from oracle.torch_dense import render_dense

    P, H, W = 200_000, 2048, 2048
    sc = make_scene(P, H, W, seed=7, opacity="trained", device=dev)
    rs = settings_for(sc, H, W)
    leaves = {k: sc[k].clone().requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "colors"]}
    color, radii, invd = GaussianRasterizer(rs)(leaves["means3D"], torch.zeros(P, 3, device=dev), leaves["opacities"],
                                                colors_precomp=leaves["colors"], scales=leaves["scales"],
                                                rotations=leaves["rotations"])
    torch.autograd.backward([color], [sc["dL_dcolor"]])
    assert torch.isfinite(color).all() and all(torch.isfinite(v.grad).all() for v in leaves.values())
    # accumulated-opacity channel (feature 4 is the constant 1, bg[4] = 0) must stay in [0, 1]
    color = color.detach()
    assert float(color[4].min()) >= -1e-6 and float(color[4].max()) <= 1 + 1e-5
    # a 64 x 64 crop at an 8-aligned offset, recomputed densely from the Gaussians that can reach it
    c = {k: v.cpu() for k, v in sc.items()}
    y0, x0, S = 1024 - 32, 1536, 64
    full = render_dense(c["means3D"], c["opacities"], c["colors"], c["bg"], c["viewmatrix"], H, W, scales=c["scales"],
                        rotations=c["rotations"], block=64, crop=(y0, x0, S, S))[0]
    assert_close(color[:, y0:y0 + S, x0:x0 + S], full, "2048 crop")


def test_backward_is_deterministic(dev):
    """Atomic-free backward: two runs give bitwise identical gradients."""
    from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer
    from eogs2_amd.synthetic import make_scene

    P, H, W = 20000, 256, 256
    sc = make_scene(P, H, W, seed=21, opacity="trained", scale_mult=1.5)
    case = {k: v.numpy() for k, v in sc.items()}
    case.update(H=H, W=W, antialiasing=False)
    a = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    b = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    for k in a:  # grad_viewmatrix included: its 18 sums are reduced from per-workgroup partials in a fixed order
        assert (a[k] == b[k]) if k.startswith("_") else torch.equal(a[k], b[k]), k


def test_heavy_lanes_of_a_surface_shaped_scene(dev, monkeypatch):
    """the size-heterogeneous Gaussians of a trained scene (synthetic.py kind="surface": a few splats tens of
    pixels wide among many of one pixel) take the paths the uniform scenes never reach — gaussian_bwd's wave-summed records
    (more than 64 listed tiles), preprocess's closed-form row counts, expand's footprints walked together. One scene, odd image
    size, through (1) the drop-in API against the oracle with full attribution, (2) the raw-parameter front end against the
    oracle's, (3) the range-split backward against the plain one, bit for bit (a wave-summed Gaussian must not depend on how
    the launch was cut)."""
    import oracle
    from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
    from eogs2_amd.parallel import GradBucket
    from eogs2_amd.synthetic import make_scene, settings_for
    from parity_cases import compare, oracle_run
    from util import raw_params_from_scene, run_raw

    P, H, W = 120_000, 500, 620
    sc = make_scene(P, H, W, seed=41, kind="surface", scale_mult=1.5)
    case = {k: v.numpy() for k, v in sc.items()}
    case.update(H=H, W=W, antialiasing=True)
    got = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    assert int((got["out_radii"] > 60).sum()) > 50  # (the large splats are there: radius = ceil(3 sigma))
    compare(got, oracle_run(case), "surface_heavy_lanes", case)
    # (2) raw-parameter mode
    raw, alt = raw_params_from_scene(sc, seed=41)
    to = lambda d: {k: v.to(dev) for k, v in d.items()}
    hip = run_raw(to(raw), alt.to(dev), to(sc), H, W, False, fused=True)
    oabi = oracle.abi()
    monkeypatch.setattr(_lib, "get", lambda: oabi)
    ref = run_raw(raw, alt, sc, H, W, False, fused=True)
    monkeypatch.undo()
    assert int((hip["out_radii"].cpu() != ref["out_radii"]).sum()) <= 2
    for k in ("out_color", "g_xyz", "g_f_dc", "g_opacity_logit", "g_log_scaling", "g_raw_rotation", "g_means2D"):
        assert_close(hip[k], ref[k], f"surface raw vs oracle:{k}", flip_floor=8, flip_rtol=5e-2)
    # (3) range-split backward == plain backward
    scd = {k: v.to(dev) for k, v in sc.items()}
    names = ("means3D", "colors", "opacities", "scales", "rotations")

    def run(chunks):
        leaves = {k: scd[k].clone().requires_grad_(True) for k in names}
        m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
        b = None
        if chunks:
            b = GradBucket([leaves[k] for k in names], cols=[slice(0, 3), slice(0, 3), slice(0, 1), slice(0, 3), slice(0, 4)],
                           names=names, chunks=chunks)
            b.begin()
        color, _, _ = GaussianRasterizer(settings_for(scd, H, W))(leaves["means3D"], m2, leaves["opacities"], colors_precomp=leaves["colors"],
                                                                 scales=leaves["scales"], rotations=leaves["rotations"])
        (color * scd["dL_dcolor"]).sum().backward()
        if b is not None:
            b.finish()
        return {**{k: v.grad.clone() for k, v in leaves.items()}, "m2": m2.grad.clone()}

    plain, ranged = run(0), run(3)
    for k in plain:
        assert torch.equal(plain[k], ranged[k]), k


@pytest.mark.parametrize("chunks", [1, 3, 4])
def test_range_backward_and_bucket_equal_plain_backward(dev, chunks):
    """eogs_rast_backward_range over ascending Gaussian ranges == eogs_rast_backward, bit for bit, and the data-parallel
    bucket (eogs2_amd.parallel.GradBucket: gradients written straight into the exchange buffer, one (absent) collective
    per range) leaves exactly the plain gradients in every .grad. Single process: no torch.distributed here."""
    from eogs2_amd import GaussianRasterizer
    from eogs2_amd.parallel import GradBucket
    from eogs2_amd.synthetic import make_scene, settings_for

    P, H, W = 5000, 160, 192
    sc = make_scene(P, H, W, seed=31, opacity="trained", scale_mult=2.0, device=dev)
    names = ("means3D", "colors", "opacities", "scales", "rotations")

    def run(bucketed):
        leaves = {k: sc[k].clone().requires_grad_(True) for k in names}
        vm = sc["viewmatrix"].clone().requires_grad_(True)
        rs = settings_for(dict(sc, viewmatrix=vm), H, W)._replace(projmatrix=vm.detach())
        m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
        b = None
        if bucketed:
            b = GradBucket([leaves[k] for k in names], cols=[slice(0, 3), slice(0, 3), slice(0, 1), slice(0, 3), slice(0, 4)],
                           names=names, chunks=chunks)
            b.begin()
        color, _, _ = GaussianRasterizer(rs)(leaves["means3D"], m2, leaves["opacities"], colors_precomp=leaves["colors"],
                                             scales=leaves["scales"], rotations=leaves["rotations"])
        (color * sc["dL_dcolor"]).sum().backward()
        if bucketed:
            b.finish()
            for i, k in enumerate(names):
                assert b._is_block(leaves[k].grad, i) == (k != "colors"), k
        out = {k: v.grad.clone() for k, v in leaves.items()}
        out.update(m2=m2.grad.clone(), vm=vm.grad.clone())
        return out

    plain, ranged = run(False), run(True)
    for k in plain:
        assert torch.equal(plain[k], ranged[k]), k


def test_full_size_properties(dev):
    """BASELINE size (1M Gaussians / 1024^2): properties that need no CPU reference.
    - linearity of backward in dL/dcolor; gradient of an all-zero dL is exactly zero
    - background: out(bg + d) - out(bg) = T_final * d, with 0 <= T_final <= 1
    - invisible Gaussians (radii == 0) receive exactly zero gradient; num visible matches radii
    - colour-channel 4 (constant 1, bg 0) renders accumulated opacity 1 - T_final
    """
    from eogs2_amd import GaussianRasterizer
    from eogs2_amd.synthetic import make_scene, settings_for

    P, H, W = 1_000_000, 1024, 1024
    sc = make_scene(P, H, W, seed=0, opacity="init", device=dev)
    rs = settings_for(sc, H, W)

    def fwd_bwd(bg, dL):
        leaves = {k: sc[k].clone().requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "colors"]}
        color, radii, invd = GaussianRasterizer(rs._replace(bg=bg))(
            leaves["means3D"], torch.zeros(P, 3, device=dev), leaves["opacities"], colors_precomp=leaves["colors"],
            scales=leaves["scales"], rotations=leaves["rotations"])
        (color * dL).sum().backward()
        return color.detach(), radii, {k: v.grad for k, v in leaves.items()}

    dL = sc["dL_dcolor"]
    c0, radii, g1 = fwd_bwd(sc["bg"], dL)
    assert torch.isfinite(c0).all()
    for v in g1.values():
        assert torch.isfinite(v).all()
    # linearity
    _, _, g2 = fwd_bwd(sc["bg"], 2.0 * dL)
    for k in g1:
        assert_close(g2[k], 2.0 * g1[k], f"linearity:{k}", allow_flips=False)
    _, _, g0 = fwd_bwd(sc["bg"], torch.zeros_like(dL))
    for k in g0:
        assert float(g0[k].abs().max()) == 0.0, k
    # background
    d = torch.tensor([0.25, -0.5, 1.0, 3.0, 2.0], device=dev)
    c1, _, _ = fwd_bwd(sc["bg"] + d, dL)
    Tf = (c1 - c0)[4] / d[4]
    assert float(Tf.min()) >= -1e-6 and float(Tf.max()) <= 1 + 1e-6
    assert_close((c1 - c0), Tf[None] * d[:, None, None], "bg-linearity", rtol=1e-4, allow_flips=False)
    # accumulated opacity channel: bg[4] = 0 -> c0[4] = 1 - T_final
    assert_close(c0[4], 1.0 - Tf, "opacity-channel", rtol=1e-4, allow_flips=False)
    # invisible Gaussians
    inv = radii == 0
    for k, v in g1.items():
        assert float(v[inv].abs().sum()) == 0.0, k


def _small_scene(dev, P=3000, H=96, W=128, seed=5):
    from eogs2_amd.synthetic import make_scene, settings_for

    sc = make_scene(P, H, W, seed=seed, opacity="trained", scale_mult=2.0, device=dev)
    return sc, settings_for(sc, H, W), P, H, W


def _render(sc, rs, P, dev, leaves=None):
    from eogs2_amd import GaussianRasterizer

    x = leaves or sc
    return GaussianRasterizer(rs)(x["means3D"], torch.zeros(P, 3, device=dev), x["opacities"], colors_precomp=x["colors"],
                                  scales=x["scales"], rotations=x["rotations"])


def test_altitude_error_is_raised_not_trapped(dev):
    """forward.cu:267-272 traps the GPU; here the same condition is a Python exception and the GPU stays usable."""
    from eogs2_amd import RastError

    sc, rs, P, H, W = _small_scene(dev)
    sc["means3D"][17, 2] = 1.0  # altitude 350 > 200
    with pytest.raises(RastError, match="too high"):
        _render(sc, rs, P, dev)
    sc["means3D"][17, 2] = 0.0
    color, _, _ = _render(sc, rs, P, dev)  # still works afterwards
    assert torch.isfinite(color).all()


def test_debug_flag_and_no_grad_and_mark_visible(dev):
    from eogs2_amd import GaussianRasterizer

    sc, rs, P, H, W = _small_scene(dev)
    ref, radii, _ = _render(sc, rs, P, dev)
    dbg, radii_d, _ = _render(sc, rs._replace(debug=True), P, dev)  # sync + check after every kernel group
    assert torch.equal(ref, dbg) and torch.equal(radii, radii_d)
    with torch.no_grad():
        ng, _, _ = _render(sc, rs, P, dev)
    assert torch.equal(ref, ng) and not ng.requires_grad
    assert GaussianRasterizer(rs).markVisible(sc["means3D"]).all()


def test_retain_graph_double_backward_call(dev):
    """The workspaces saved by forward serve any number of backward calls (live flags depend on forward state only)."""
    sc, rs, P, H, W = _small_scene(dev)
    leaves = {k: sc[k].clone().requires_grad_(True) for k in ["means3D", "scales", "rotations", "opacities", "colors"]}
    color, _, _ = _render(sc, rs, P, dev, leaves)
    torch.autograd.backward([color], [sc["dL_dcolor"]], retain_graph=True)
    g1 = {k: v.grad.clone() for k, v in leaves.items()}
    for v in leaves.values():
        v.grad = None
    torch.autograd.backward([color], [3.0 * sc["dL_dcolor"]])
    for k, v in leaves.items():
        assert_close(v.grad, 3.0 * g1[k], f"second backward:{k}", allow_flips=False)


def test_non_default_stream_and_strided_inputs(dev):
    sc, rs, P, H, W = _small_scene(dev)
    ref, _, _ = _render(sc, rs, P, dev)
    # non-contiguous / fp64 inputs are made contiguous fp32 by the wrapper, like rasterize_points.cu:101-120
    wide = torch.zeros(P, 7, device=dev, dtype=torch.float64)
    wide[:, 1:4] = sc["means3D"]
    sc2 = dict(sc, means3D=wide[:, 1:4])
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        out, _, _ = _render(sc2, rs, P, dev)
    st.synchronize()
    assert torch.equal(ref, out)


# ---- raw-parameter front end (activations fused into the per-Gaussian kernels) ----
def _close_all(a, b, name, means3D, rtol=1e-4):
    assert torch.equal(a["out_radii"].cpu(), b["out_radii"].cpu()), f"{name}: radii differ"
    for k in a:
        if k == "out_radii":
            continue
        if k == "g_viewmatrix":
            g2 = b["g_means2D"].abs().double().cpu()
            m = means3D.abs().double().cpu()
            scale = max(float((m.t() @ g2).max()), float(g2.sum(0).max()), 1e-30)
            err = float((a[k].double().cpu() - b[k].double().cpu()).abs().max()) / scale
            assert err <= 1e-4, f"{name}:{k}: {err:.3e} of the magnitude sum"
            continue
        assert_close(a[k], b[k], f"{name}:{k}", rtol=rtol, flip_floor=4)


@pytest.mark.parametrize("aa", [False, True])
def test_fused_matches_unfused(dev, aa):
    """HIP raw-parameter path == the reference's PyTorch activation ops + the drop-in rasterizer (autograd chain)."""
    from util import raw_params_from_scene, run_raw

    from eogs2_amd.synthetic import make_scene

    H, W, P = 256, 320, 60000
    scene = make_scene(P, H, W, seed=31, opacity="trained", device=dev, scale_mult=1.5)
    raw, alt = raw_params_from_scene(scene)
    dinv = torch.randn(1, H, W, device=dev) / (H * W)
    a = run_raw(raw, alt, scene, H, W, aa, fused=True, dL_dinvdepth=dinv)
    b = run_raw(raw, alt, scene, H, W, aa, fused=False, dL_dinvdepth=dinv)
    _close_all(a, b, f"fused_vs_unfused_aa{int(aa)}", scene["means3D"])
    assert float(a["g_raw_rotation"].abs().max()) > 0 and float(a["g_log_scaling"].abs().max()) > 0


def test_fused_matches_oracle(dev, monkeypatch):
    """HIP raw-parameter path == the oracle's restatement of the same front end (C, double-precision chain)."""
    import oracle
    from util import raw_params_from_scene, run_raw

    from eogs2_amd import _lib
    from eogs2_amd.synthetic import make_scene

    H, W, P = 96, 80, 3000
    scene = make_scene(P, H, W, seed=32, opacity="trained", scale_mult=2.0)
    raw, alt = raw_params_from_scene(scene)
    to = lambda d: {k: v.to(dev) for k, v in d.items()}
    a = run_raw(to(raw), alt.to(dev), to(scene), H, W, True, fused=True)
    oabi = oracle.abi()
    monkeypatch.setattr(_lib, "get", lambda: oabi)
    b = run_raw(raw, alt, scene, H, W, True, fused=True)
    monkeypatch.undo()
    _close_all(a, b, "fused_vs_oracle", scene["means3D"])


def test_render_entry_point(dev):
    """eogs2_amd.render.render (the reference's renderer.py signature): the fused path agrees with the unfused ops + drop-in rasterizer built in the test."""
    import types

    from test_fused_cpu import _Cam, _Model
    from util import raw_params_from_scene, render_unfused

    from eogs2_amd.render import render
    from eogs2_amd.synthetic import make_scene

    H, W, P = 200, 168, 20000
    scene = make_scene(P, H, W, seed=33, opacity="trained", device=dev, scale_mult=1.5)
    raw, _ = raw_params_from_scene(scene)
    pipe = types.SimpleNamespace(debug=False, antialiasing=True, compute_cov3D_python=False, require_radii=True)
    res = {}
    for fused in (True, False):
        cam, pc = _Cam(scene["viewmatrix"], H, W), _Model(raw)
        cam.last_row = cam.last_row.detach().to(dev).requires_grad_(True)
        cam.camera_center = cam.camera_center.to(dev)
        out = render(cam, pc, pipe, scene["bg"]) if fused else render_unfused(cam, pc, pipe, scene["bg"])
        (out["render"] * scene["dL_dcolor"]).sum().backward()
        res[fused] = dict(render=out["render"].detach(), vsp=out["viewspace_points"].grad, last_row=cam.last_row.grad,
                          **{k: v.grad for k, v in pc.params().items()})
        assert torch.equal(out["visibility_filter"], (out["radii"] > 0).nonzero())
    for k in res[True]:
        if k == "last_row":
            scale = float(res[False]["vsp"].abs().sum(0).max())
            assert float((res[True][k] - res[False][k]).abs().max()) <= 1e-4 * scale
        else:
            assert_close(res[True][k], res[False][k], k, flip_floor=4)


@pytest.mark.parametrize("P,H,W", [(1, 1, 1), (2, 1, 7), (70, 5, 3), (300, 8, 8), (65, 2, 33), (1, 40, 1)])
def test_images_smaller_than_a_tile_or_a_block(dev, P, H, W):
    """Images of a few pixels: one partial 8 x 8 tile, one partial 32 x 32-px block, a tile schedule over a single block, Gaussians
    larger than the image. Every output and gradient against the oracle."""
    from parity_cases import compare, oracle_run
    from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer
    from eogs2_amd.synthetic import make_scene

    sc = make_scene(P, H, W, seed=100 + P + H + W, opacity="trained", scale_mult=3.0)
    case = {k: v.numpy() for k, v in sc.items()}
    case.update(H=H, W=W, antialiasing=False)
    got = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    compare(got, oracle_run(case), f"tiny{P}_{H}x{W}", case)
