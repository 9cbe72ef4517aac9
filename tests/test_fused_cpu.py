"""CPU: the raw-parameter (fused activations) front end, SURVEY.md §8 row f1.

The host wiring (eogs2_amd/fused.py, eogs2_amd/render.py) runs over the oracle library (test-only injection) and
is checked against the reference's own PyTorch formulation of the same ops differentiated by autograd
(gaussian_model.py:41-52,109-137; gaussian_renderer/renderer.py:91-96) feeding the unfused rasterizer.
"""
import types

import pytest
import torch

from util import assert_close, raw_params_from_scene, run_raw

from eogs2_amd.synthetic import make_scene


@pytest.mark.parametrize("aa", [False, True])
def test_oracle_fused_matches_autograd_composition(oracle_backend, aa):
    H, W, P = 48, 40, 300
    scene = make_scene(P, H, W, seed=21, opacity="trained", scale_mult=3.0)
    raw, alt = raw_params_from_scene(scene)
    dinv = torch.randn(1, H, W) / (H * W)
    a = run_raw(raw, alt, scene, H, W, aa, fused=True, dL_dinvdepth=dinv)
    b = run_raw(raw, alt, scene, H, W, aa, fused=False, dL_dinvdepth=dinv)
    assert torch.equal(a["out_radii"], b["out_radii"])
    assert int((a["out_radii"] > 0).sum()) > P // 2
    for k in a:
        if k != "out_radii":
            assert_close(a[k], b[k], k, rtol=2e-5)
    assert float(a["g_raw_rotation"].abs().max()) > 0 and float(a["g_opacity_logit"].abs().max()) > 0


class _Cam:
    def __init__(self, vm, H, W):
        self.FoVx = self.FoVy = 1.0
        self.affine = vm
        self.world_view_transform = vm
        self.full_proj_transform = vm
        self.learn_wv_only_lastparam = True
        self.last_row = torch.tensor([0.01, -0.02, 0.03, 0.0], requires_grad=True)
        self.image_height, self.image_width = H, W
        self.camera_center = torch.zeros(3)
        self.image_name = "synthetic"

    def ECEF_to_UVA(self, xyz):
        return xyz @ self.affine[:3, :3] + self.affine[3, :3]


class _Model:
    active_sh_degree = 0

    def __init__(self, raw):
        self._xyz = raw["xyz"].clone().requires_grad_(True)
        self._features_dc = raw["f_dc"].clone().requires_grad_(True)
        self._opacity = raw["opacity_logit"].clone().requires_grad_(True)
        self._scaling = raw["log_scaling"].clone().requires_grad_(True)
        self._rotation = raw["raw_rotation"].clone().requires_grad_(True)

    get_xyz = property(lambda s: s._xyz)
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))

    def params(self):
        return dict(xyz=self._xyz, f_dc=self._features_dc, opacity=self._opacity, scaling=self._scaling,
                    rotation=self._rotation)


def test_render_entry_point_fused_vs_reference_ops(oracle_backend):
    """`render()` with the reference's signature: the fused path and the reference's op sequence give the same
    image, radii and gradients — including the learnable last row of the view matrix (learn_wv_only_lastparam)."""
    from util import render_unfused

    from eogs2_amd.render import render

    H, W, P = 40, 56, 200
    scene = make_scene(P, H, W, seed=5, opacity="trained", scale_mult=3.0)
    raw, _ = raw_params_from_scene(scene)
    pipe = types.SimpleNamespace(debug=False, antialiasing=False, compute_cov3D_python=False, require_radii=True)
    res = {}
    for fused in (True, False):
        cam, pc = _Cam(scene["viewmatrix"], H, W), _Model(raw)
        out = render(cam, pc, pipe, scene["bg"]) if fused else render_unfused(cam, pc, pipe, scene["bg"])
        assert set(out) == {"render", "viewspace_points", "visibility_filter", "radii"}
        (out["render"] * scene["dL_dcolor"]).sum().backward()
        res[fused] = dict(render=out["render"].detach(), radii=out["radii"], vsp=out["viewspace_points"].grad,
                          last_row=cam.last_row.grad, **{k: v.grad for k, v in pc.params().items()})
    assert torch.equal(res[True]["radii"], res[False]["radii"])
    assert float(res[True]["last_row"].abs().max()) > 0
    for k in res[True]:
        if k != "radii":
            assert_close(res[True][k], res[False][k], k, rtol=2e-5)


def test_render_entry_point_delegates_what_it_does_not_fuse(oracle_backend):
    """Inputs outside the raw-parameter path go to the caller's own render (or raise): nothing is re-implemented."""
    from eogs2_amd.render import fusable, render

    scene = make_scene(10, 16, 16, seed=1)
    raw, _ = raw_params_from_scene(scene)
    cam, pc = _Cam(scene["viewmatrix"], 16, 16), _Model(raw)
    pipe = types.SimpleNamespace(debug=False, antialiasing=False, compute_cov3D_python=True, require_radii=True)
    assert not fusable(cam, pipe) and not fusable(cam, types.SimpleNamespace(compute_cov3D_python=False), torch.zeros(10, 5))
    with pytest.raises(NotImplementedError):
        render(cam, pc, pipe, scene["bg"])
    seen = []
    assert render(cam, pc, pipe, scene["bg"], 2.0, fallback=lambda *a: seen.append(a) or "theirs") == "theirs"
    assert seen[0][:5] == (cam, pc, pipe, scene["bg"], 2.0) and len(seen[0]) == 8



def test_count_bookkeeping_of_deferred_and_captured_forwards():
    """Host logic behind EOGS_FLAG_DEFER_COUNTS / GraphedStep (eogs2_amd/rasterizer.py): the capacity a capture is sized from
    is the component-wise maximum of the counts seen, with the flags (list granularity, 8-item build) of the LATEST
    forward; the switches report their previous state."""
    from eogs2_amd import rasterizer as rz

    def tok(slots, entries, block=0, wide=0, sorted_=0):
        return (block << 62) | (sorted_ << 61) | (entries << 32) | (wide << 31) | slots

    assert rz._merge_counts(None, tok(5, 3, block=1)) == tok(5, 3, block=1)
    m = rz._merge_counts(tok(1000, 10, block=1, wide=1, sorted_=1), tok(400, 50))
    assert m == tok(1000, 50)  # max of each count, flags of the later one
    m = rz._merge_counts(tok(400, 50), tok(1000, 10, block=1, wide=1))
    assert m == tok(1000, 50, block=1, wide=1)
    assert (tok(0x7FFFFFFF, 0x07FFFFFF) & rz._SLOTS) == 0x7FFFFFFF and (tok(0, 0x07FFFFFF) & rz._ENTRIES) >> 32 == 0x07FFFFFF
    assert ((1 << 59) & (rz._SLOTS | rz._ENTRIES)) == 0  # bit 59: an altitude-only forward (csrc/common.h nr_alt)
    old = rz.set_speculation(True)
    try:
        assert rz.set_speculation(False) is True and rz.set_speculation(True, forget=True) is False and not rz._spec
    finally:
        rz.set_speculation(old)
    rz.speculation_stats(reset=True)
    assert rz.speculation_stats() == {"exact": 0, "hit": 0, "redo": 0}


def test_gradbucket_zero_keeps_the_gradients_inside_the_exchange_buffer():
    """GradBucket.zero_(): .grad of every parameter is the (cleared) view of its block, backward passes accumulate in
    place — any number of them — and all_reduce() exchanges without a copy: the same tensors every step, which is what a
    step recorded into a graph needs."""
    from eogs2_amd.parallel import GradBucket

    torch.manual_seed(0)
    P = 300
    params = [torch.randn(P, 3, requires_grad=True), torch.randn(P, 1, 3, requires_grad=True), torch.randn(P, 1, requires_grad=True)]
    bucket = GradBucket(params)

    def loss(k):
        return sum(((p * (k + 1.0)) ** 2).sum() for p in params)

    bucket.zero_()
    views = [p.grad for p in params]
    assert all(bucket._is_block(p.grad, i) for i, p in enumerate(params)) and float(bucket.flat.abs().sum()) == 0.0
    for k in range(3):
        loss(k).backward()
    assert all(p.grad is v for p, v in zip(params, views))  # accumulated in place
    bucket.all_reduce()  # (no process group: nothing to exchange, nothing to copy)
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, views))
    want = [2.0 * p.detach() * sum((k + 1.0) ** 2 for k in range(3)) for p in params]
    for p, w in zip(params, want):
        assert torch.allclose(p.grad, w, rtol=1e-6, atol=1e-6)
    bucket.zero_()  # the next step: same tensors, cleared
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, views)) and float(bucket.flat.abs().sum()) == 0.0
    bucket.close()


def test_oracle_refuses_the_altitude_only_render(oracle_backend):
    """the restatement follows the reference, which has no one-channel render. Over the oracle library
    `rasterize_raw(..., altitude_only=True)` must fail with its error, not write five planes into a [1,H,W] image."""
    from eogs2_amd import RastError
    from eogs2_amd.fused import rasterize_raw
    from eogs2_amd.synthetic import settings_for

    H, W, P = 32, 24, 50
    scene = make_scene(P, H, W, seed=4, opacity="trained")
    raw, alt = raw_params_from_scene(scene)
    rs = settings_for(scene, H, W)
    m2 = torch.zeros(P, 3)
    with pytest.raises(RastError, match="ALT_ONLY"):
        rasterize_raw(raw["xyz"], m2, raw["f_dc"], raw["opacity_logit"], raw["log_scaling"], raw["raw_rotation"], alt, rs,
                      altitude_only=True)
    # the full render of the same inputs goes through
    color, radii, invd = rasterize_raw(raw["xyz"], m2, raw["f_dc"], raw["opacity_logit"], raw["log_scaling"],
                                       raw["raw_rotation"], alt, rs)
    assert tuple(color.shape) == (5, H, W) and tuple(invd.shape) == (1, H, W)
