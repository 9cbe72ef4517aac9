"""GPU (MI355X): fused virtual-camera resample (include/eogs_resample.h, eogs2_amd/resample.py) against the oracle —
the reference's statements over torch.nn.functional.grid_sample in float64 (oracle/resample_oracle.py) — and against
the same statements run in fp32 by PyTorch on the GPU (what the reference executes)."""
import pytest
import torch

from util import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from eogs2_amd import _lib

    assert _lib.get().backend == "hip-gfx950"
    return torch.device("cuda:0")


def _case(H, W, f, seed, shear=0.08, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    vr = torch.rand(5, H * f, W * f, generator=g)
    vr[3] = vr[3] * 40 - 10  # altitude-like channel
    alt = torch.rand(H, W, generator=g) * 2 - 0.5
    U, V = torch.meshgrid(torch.linspace(-1, 1, W), torch.linspace(-1, 1, H), indexing="xy")
    M = torch.eye(3)
    M[:2, 2] = torch.tensor([shear, -0.7 * shear])
    M[0, 0], M[1, 1] = 1.0 + shift, 1.0 - 0.5 * shift  # pushes part of the grid outside [-1, 1]
    w_s = torch.randn(4, H, W, generator=g)
    w_uv = torch.randn(H, W, 2, generator=g)
    return vr, alt, (U, V), M, w_s, w_uv


def _run(fn, vr, alt, UV, M, w_s, w_uv, dev):
    vr = vr.to(dev).requires_grad_(True)
    alt = alt.to(dev).requires_grad_(True)
    uva = torch.stack((UV[0].to(dev), UV[1].to(dev), alt), dim=-1)  # train_pan.py:281
    s, uv = fn(vr, M.to(dev), uva)
    ((s * w_s.to(dev)).sum() + (uv * w_uv.to(dev)).sum()).backward()
    return s.detach().cpu(), uv.detach().cpu(), vr.grad.cpu(), alt.grad.cpu()


@pytest.mark.parametrize("H,W,f,shift", [(64, 80, 2, 0.0), (33, 47, 1, 0.3), (128, 128, 2, -0.2), (17, 300, 1, 0.0)])
def test_resample_matches_oracle(dev, H, W, f, shift):
    from oracle import resample_oracle

    from eogs2_amd.resample import resample

    c = _case(H, W, f, seed=H + W, shift=shift)
    got = _run(resample, *c, dev)
    ref = _run(resample_oracle.resample, *c, torch.device("cpu"))
    if shift:
        assert int((got[0][3] == -100).sum()) > 0  # the out-of-view fill is exercised
    for name, a, b in zip(("sample", "uv", "g_virtual_render", "g_altitude"), got, ref):
        assert_close(a, b, name, rtol=1e-4, allow_flips=False)  # fp32 pixel coordinates up to 2047: weights good to ~1e-4


def test_resample_matches_reference_ops_on_gpu_full_size(dev):
    """1024^2 true camera, 2048^2 sun camera: the reference's fp32 op sequence on the same GPU."""
    from eogs2_amd.resample import resample

    def ref_ops(vr, M, uva):
        uv = torch.einsum("...ij,...j->...i", M, uva)[..., :2]
        s = torch.nn.functional.grid_sample(vr.unsqueeze(0), uv.unsqueeze(0), align_corners=True).squeeze(0)
        rgb, a = s[:3], s[3]
        a[(uv.abs() > 1).any(-1)] = -100
        return torch.cat([rgb, a[None]], 0), uv

    c = _case(1024, 1024, 2, seed=7, shift=0.05)
    got = _run(resample, *c, dev)
    ref = _run(ref_ops, *c, dev)
    for name, a, b in zip(("sample", "uv", "g_virtual_render", "g_altitude"), got, ref):
        # two fp32 evaluations (each within 1e-4 of the float64 oracle at small sizes), atomics in different orders.
        # Where a coordinate rounds onto a cell border the two pick different cells: the sample is continuous there but
        # its derivative with respect to the coordinate is not, so a few pixels in a million may differ arbitrarily.
        err = (a.double() - b.double()).abs() / b.abs().max().clamp_min(1e-30).double()
        assert float((err > 2e-4).double().mean()) <= 2e-5, f"{name}: {float(err.max()):.3e}"
    assert float(got[2][4].abs().max()) == 0.0  # the unused accumulated-opacity channel receives no gradient


def test_resample_entry_point_and_errors(dev):
    import types

    from test_fused_cpu import _Cam, _Model
    from util import raw_params_from_scene

    from eogs2_amd.resample import render_resample_virtual_camera, resample
    from eogs2_amd.synthetic import make_scene

    H, W, P = 96, 128, 3000
    scene = make_scene(P, H, W, seed=3, opacity="trained", device=dev, scale_mult=2.0)
    raw, _ = raw_params_from_scene(scene)
    cam, pc = _Cam(scene["viewmatrix"], H, W), _Model(raw)
    cam.last_row = cam.last_row.detach().to(dev)
    cam.camera_center = cam.camera_center.to(dev)
    pipe = types.SimpleNamespace(debug=False, antialiasing=False, compute_cov3D_python=False, require_radii=False)
    U, V = torch.meshgrid(torch.linspace(-1, 1, W, device=dev), torch.linspace(-1, 1, H, device=dev), indexing="xy")
    uva = torch.stack((U, V, torch.zeros(H, W, device=dev)), dim=-1)
    rgb, alt, uv = render_resample_virtual_camera(cam, torch.eye(3, device=dev), uva, pc, pipe, scene["bg"])
    assert rgb.shape == (3, H, W) and alt.shape == (H, W) and uv.shape == (H, W, 2)
    # identity reprojection onto the same grid reproduces the render itself
    from eogs2_amd.render import render

    direct = render(cam, pc, pipe, scene["bg"])["render"]
    assert_close(rgb, direct[:3], "identity resample", rtol=1e-5, allow_flips=False)
    (rgb.sum() + alt.sum()).backward()
    assert pc._xyz.grad is not None and float(pc._xyz.grad.abs().max()) > 0
    with pytest.raises(RuntimeError):
        resample(direct, torch.eye(4, device=dev), uva)
    with pytest.raises(RuntimeError):
        resample(direct, torch.eye(3, device=dev), uva[..., :2])


# ---- vectors from the REFERENCE's own render_resample_virtual_camera (tests/golden/make_golden_resample.py) ----
import glob  # noqa: E402
import os  # noqa: E402

from util import GOLDEN_DIR  # noqa: E402

RESAMPLE_FIXTURES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "resample_*.npz")))


@pytest.mark.parametrize("path", RESAMPLE_FIXTURES, ids=lambda p: os.path.basename(p)[9:-4])
def test_resample_matches_reference_vectors(dev, path):
    """The HIP resample (forward and backward through the C-ABI) against what the reference's function returned and what
    autograd gave for it: RGB / altitude samples with the -100 fill, coordinates, d/d virtual render, d/d altitude."""
    from test_resample_oracle import compare, load, run

    from eogs2_amd.resample import resample

    c = load(path)
    compare(run(resample, c, dev), c, rtol=1e-4)


def test_entry_point_matches_reference_vectors_with_a_stub_render(dev, monkeypatch):
    """eogs2_amd.resample.render_resample_virtual_camera (the reference's signature) with step 1 replaced by the fixture's
    virtual render, exactly as the fixture was produced."""
    from test_resample_oracle import compare, load

    import eogs2_amd.render as R
    from eogs2_amd.resample import render_resample_virtual_camera

    c = load(RESAMPLE_FIXTURES[1])
    vr = c["virtual_render"].to(dev).requires_grad_(True)
    alt = c["altitude"].to(dev).requires_grad_(True)
    monkeypatch.setattr(R, "render", lambda cam, pc, pipe, bg: {"render": vr})
    uva = torch.stack((c["U"].to(dev), c["V"].to(dev), alt), dim=-1)
    rgb, a, uv, extra = render_resample_virtual_camera(None, c["cam2virt"].to(dev), uva, None, None, None, return_extra=True)
    assert extra is vr
    ((rgb * c["w_rgb"].to(dev)).sum() + (a * c["w_alt"].to(dev)).sum() + (uv * c["w_uv"].to(dev)).sum()).backward()
    got = dict(rgb_sample=rgb.detach().cpu(), altitude_sample=a.detach().cpu(), virtual_uv=uv.detach().cpu(),
               g_virtual_render=vr.grad.cpu(), g_altitude=alt.grad.cpu())
    compare(got, c, rtol=1e-4)


def test_first_form_of_the_backward_tile_kernel_still_matches_the_oracle(dev):
    """EOGS_RESAMPLE_BWD=1 (read once per process) selects the round-2 tile kernel (LDS float atomics) instead of the bucketed
    gather: the oracle comparisons above, in a child process with it set."""
    import os
    import subprocess
    import sys

    if os.environ.get("EOGS_RESAMPLE_BWD") == "1":
        pytest.skip("already the child")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_resample.py"), "-m", "gpu", "-q", "-x",
                        "-k", "matches_oracle or matches_reference_ops"], env=dict(os.environ, EOGS_RESAMPLE_BWD="1"),
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_backward_is_reproducible_bit_for_bit(dev):
    """The gather sums in a fixed order (candidate tiles in tile order, a bucket's entries in pixel order): repeated backward
    passes over the same inputs give the same bits — at 1:1 and under minification, where several true-camera pixels land in one
    virtual cell and the order of the sum is visible, one plane and four."""
    from eogs2_amd.resample import resample

    for f, H, W, n_out in ((1, 300, 420, 4), (1, 300, 420, 1)):
        vr, alt, UV, M, w_s, w_uv = _case(H, W, f, seed=7, shift=0.2)
        M = M.clone()
        M[0, 0], M[1, 1] = 0.45, 0.6  # several pixels per virtual cell
        grads = []
        for _ in range(6):
            v = vr[:5 if n_out == 4 else 1].to(dev).requires_grad_(True)
            a = alt.to(dev).requires_grad_(True)
            uva = torch.stack((UV[0].to(dev), UV[1].to(dev), a), dim=-1)
            s, uv = resample(v, M.to(dev), uva, n_out=n_out, fill_channel=n_out - 1)
            ((s * w_s[:n_out].to(dev)).sum() + (uv * w_uv.to(dev)).sum()).backward()
            grads.append((v.grad.clone(), a.grad.clone()))
        assert float(grads[0][0].abs().max()) > 0
        for gv, ga in grads[1:]:
            assert torch.equal(gv, grads[0][0]) and torch.equal(ga, grads[0][1])
