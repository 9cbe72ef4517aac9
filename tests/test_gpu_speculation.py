"""GPU (MI355X): forwards that do not wait for their count readback (include/eogs_rast.h EOGS_FLAG_DEFER_COUNTS).

After the first forward of a shape the wrapper (eogs2_amd/rasterizer.py `_run_forward`) sizes the binning workspace from the
previous forward's counts plus slack, queues the whole forward and asks for the counts afterwards; a forward whose lists do
not fit gets none on the device (csrc/binning.hip block_lists_kernel) and is repeated with the exact counts. Whichever way
a forward went, its outputs and gradients are those of the forward that waited — bit for bit when the lists have the same
granularity, against the oracle otherwise."""
import numpy as np
import pytest
import torch

from util import run_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from eogs2_amd import _lib

    assert _lib.get().backend == "hip-gfx950"
    return torch.device("cuda:0")


def _case(P, H, W, seed, **kw):
    from eogs2_amd.synthetic import make_scene

    sc = make_scene(P, H, W, seed=seed, **kw)
    case = {k: v.numpy() for k, v in sc.items()}
    case.update(H=H, W=W, antialiasing=False)
    return case


def _run(case, dev):
    from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer

    return run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)


def _exact(case, dev):
    from eogs2_amd import rasterizer

    old = rasterizer.set_speculation(False)
    try:
        return _run(case, dev)
    finally:
        rasterizer.set_speculation(old)


def _same(a, b):
    for k in a:
        if not k.startswith("_"):
            assert torch.equal(a[k], b[k]), k


def test_second_forward_of_a_shape_is_queued_whole_and_is_bit_identical(dev):
    from eogs2_amd import rasterizer

    rasterizer.set_speculation(True, forget=True)
    rasterizer.speculation_stats(reset=True)
    first, second = _case(20000, 200, 168, 3, opacity="trained"), _case(20000, 200, 168, 4, opacity="trained")
    _run(first, dev)
    got = _run(second, dev)
    st = rasterizer.speculation_stats()
    assert st == {"exact": 1, "hit": 1, "redo": 0}, st
    assert got["_num_rendered"] != got["_num_rendered_exact"]  # the workspace layout is the guess, the counts are its own
    assert (got["_num_rendered"] & 0x7FFFFFFF) >= (got["_num_rendered_exact"] & 0x7FFFFFFF)
    _same(got, _exact(second, dev))


@pytest.mark.parametrize("grow", ["slots and entries", "entries beyond the scratch"])
def test_forward_that_outgrows_the_guess_is_repeated(dev, grow):
    """The second scene lists several times what the first did: the device must build no lists into the guessed workspace
    (nothing may be written out of bounds — the forward after it still has to be right) and the wrapper repeats it."""
    from eogs2_amd import rasterizer

    rasterizer.set_speculation(True, forget=True)
    P, H, W = (20000, 200, 168) if grow == "slots and entries" else (300, 200, 168)
    small = _case(P, H, W, 5, opacity="trained", scale_mult=0.4)
    big = _case(P, H, W, 6, opacity="trained", scale_mult=3.0 if grow == "slots and entries" else 14.0)
    rasterizer.speculation_stats(reset=True)
    _run(small, dev)
    got = _run(big, dev)
    assert rasterizer.speculation_stats() == {"exact": 1, "hit": 0, "redo": 1}
    assert got["_num_rendered"] == got["_num_rendered_exact"]
    _same(got, _exact(big, dev))
    again = _run(small, dev)  # now the guess is the big one: fits
    assert rasterizer.speculation_stats()["hit"] == 1
    ref = _exact(small, dev)
    variant = lambda t: (t >> 60) & 0b101  # bit 62: block lists, bit 60: back-to-front backward (csrc/common.h)
    if variant(again["_num_rendered"]) == variant(ref["_num_rendered"]):
        _same(again, ref)
    else:  # the guess also carries the kernel variants of the forward it was counted on: same blend order, other roundings
        for k in ("out_color", "out_invdepth"):
            np.testing.assert_allclose(again[k].cpu().numpy(), ref[k].cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_nothing_visible_after_a_full_scene(dev):
    from eogs2_amd import rasterizer

    rasterizer.set_speculation(True, forget=True)
    full = _case(5000, 96, 96, 7, opacity="trained")
    none = dict(full)
    none["means3D"] = full["means3D"] + np.float32([50.0, 50.0, 0.0])  # everything outside the image
    rasterizer.speculation_stats(reset=True)
    _run(full, dev)
    got = _run(none, dev)
    assert rasterizer.speculation_stats()["hit"] == 1
    assert (got["_num_rendered_exact"] & 0x7FFFFFFF) == 0
    _same(got, _exact(none, dev))
    bg = torch.from_numpy(full["bg"]).to(dev)
    assert torch.equal(got["out_color"], bg[:, None, None].expand_as(got["out_color"]))
    _same(_run(full, dev), _exact(full, dev))  # and a guess of (almost) nothing is outgrown by the full scene


def test_altitude_error_is_still_reported(dev):
    from eogs2_amd import RastError, rasterizer

    rasterizer.set_speculation(True, forget=True)
    ok = _case(3000, 64, 64, 8, opacity="trained")
    bad = dict(ok)
    bad["means3D"] = ok["means3D"].copy()
    bad["means3D"][17, 2] = 1.0  # altitude 350 > 200 (DGR/cuda_rasterizer/forward.cu:267-272)
    _run(ok, dev)
    with pytest.raises(RastError, match="too high"):
        _run(bad, dev)
    _same(_run(ok, dev), _exact(ok, dev))


def test_fused_front_end_speculates_too(dev):
    """The raw-parameter entry point (eogs2_amd.fused) shares _run_forward: second call queued whole, same results."""
    from eogs2_amd import rasterizer
    from eogs2_amd.fused import rasterize_raw
    from eogs2_amd.synthetic import make_scene, settings_for

    P, H, W = 30000, 256, 256
    outs = []
    for spec in (True, False):
        rasterizer.set_speculation(spec, forget=True)
        rasterizer.speculation_stats(reset=True)
        res = None
        for seed in (11, 12):
            sc = make_scene(P, H, W, seed=seed, opacity="trained", device=dev)
            rs = settings_for(sc, H, W)
            xyz = sc["means3D"].clone().requires_grad_(True)
            f_dc = torch.logit(sc["colors"][:, :3].clamp(0.01, 0.99)).requires_grad_(True)
            opl = torch.logit(sc["opacities"].clamp(1e-4, 1 - 1e-4)).requires_grad_(True)
            lsc = sc["scales"].log().requires_grad_(True)
            rot = sc["rotations"].clone().requires_grad_(True)
            alt = torch.tensor([0.0, 0.0, 1.0, 0.0], device=dev)
            m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
            color, radii, invd = rasterize_raw(xyz, m2, f_dc, opl, lsc, rot, alt, rs)
            (color * sc["dL_dcolor"]).sum().backward()
            res = [color.detach(), radii, xyz.grad, f_dc.grad, opl.grad, lsc.grad, rot.grad]
        st = rasterizer.speculation_stats()
        assert st == ({"exact": 1, "hit": 1, "redo": 0} if spec else {"exact": 2, "hit": 0, "redo": 0}), st
        outs.append(res)
    rasterizer.set_speculation(True, forget=True)
    for x, y in zip(*outs):
        assert torch.equal(x, y)
