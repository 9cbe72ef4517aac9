"""GPU (MI355X): multi-tensor Adam and row compaction (include/eogs_optim.h, eogs2_amd/optim.py) against PyTorch's own
`torch.optim.Adam` and boolean-mask indexing — the two things the reference uses (gaussian_model.py:228-262,466-505)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (0, 3), "opacity": (1,), "scaling": (3,), "rotation": (4,)}
LRS = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opacity": 5e-2, "scaling": 5e-3, "rotation": 1e-3}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from eogs2_amd import _lib

    assert _lib.get().backend == "hip-gfx950"
    return torch.device("cuda:0")


def _groups(P, device, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [{"params": [torch.nn.Parameter(torch.randn((P,) + s, generator=g).to(device))], "lr": LRS[n], "name": n}
            for n, s in SHAPES.items()]


@pytest.mark.parametrize("P", [1, 1023, 50_001])
def test_fused_adam_matches_torch_adam(dev, P):
    from eogs2_amd.optim import FusedAdam

    ref = torch.optim.Adam(_groups(P, "cpu"), lr=0.0, eps=1e-15)  # the reference's constructor call
    ours = FusedAdam(_groups(P, dev), lr=0.0, eps=1e-15)
    g = torch.Generator().manual_seed(1)
    for it in range(6):
        for gr, go in zip(ref.param_groups, ours.param_groups):
            grad = torch.randn(gr["params"][0].shape, generator=g) * (10.0 ** (it - 3))
            grad[::2] = 0.0  # Gaussians no view touched this iteration
            gr["params"][0].grad = grad
            go["params"][0].grad = grad.to(dev)
        if it == 3:  # the reference changes learning rates between steps (update_learning_rate)
            ref.param_groups[0]["lr"] = ours.param_groups[0]["lr"] = 3e-5
        ref.step()
        ours.step()
    for gr, go in zip(ref.param_groups, ours.param_groups):
        pr, po = gr["params"][0], go["params"][0]
        # fp32 on both sides, identical formula; rounding differs in lerp / division: 2e-6 of the tensor's scale
        close = lambda a, b: b.numel() == 0 or bool(((a - b).abs() <= 2e-6 * b.abs().max().clamp_min(1e-30)).all())
        assert close(po.detach().cpu(), pr.detach()), gr["name"]
        if pr.numel():
            for k in ("exp_avg", "exp_avg_sq"):
                assert close(ours.state[po][k].cpu(), ref.state[pr][k]), (gr["name"], k)
            assert int(ours.state[po]["step"]) == int(ref.state[pr]["step"]) == 6


@pytest.mark.parametrize("N,frac", [(1, 1.0), (255, 0.5), (256, 0.0), (100_003, 0.9), (70_000, 0.01), (4096, 1.0)])
def test_compact_rows_equals_boolean_indexing(dev, N, frac):
    from eogs2_amd.optim import compact_rows

    g = torch.Generator().manual_seed(N)
    mask = (torch.rand(N, generator=g) < frac).to(dev)
    tensors = [torch.randn(N, 3, generator=g).to(dev), torch.randn(N, 1, 3, generator=g).to(dev),
               torch.randn(N, 0, 3).to(dev), torch.randn(N, generator=g).to(dev),
               torch.randint(0, 1000, (N, 4), generator=g, dtype=torch.int32).to(dev),
               torch.randn(N, 14, generator=g).to(dev)[:, ::2]]  # a strided view: made contiguous like tensor[mask] would
    out = compact_rows(mask, tensors)
    for o, t in zip(out, tensors):
        assert o.shape == t[mask].shape and torch.equal(o, t[mask])


def test_prune_optimizer_keeps_training_state(dev):
    """Same result as the reference's per-tensor `_prune_optimizer` / `prune_points`, then the optimizer keeps stepping."""
    from eogs2_amd.optim import FusedAdam, prune_optimizer

    P = 10_000
    opt = FusedAdam(_groups(P, dev, seed=3), lr=0.0, eps=1e-15)
    g = torch.Generator().manual_seed(4)
    for gr in opt.param_groups:
        gr["params"][0].grad = torch.randn(gr["params"][0].shape, generator=g).to(dev)
    opt.step()
    keep = (torch.rand(P, generator=g) < 0.7).to(dev)
    before = {gr["name"]: (gr["params"][0].detach().clone(), {k: v.clone() for k, v in opt.state[gr["params"][0]].items()})
              for gr in opt.param_groups}
    accum, radii = torch.rand(P, 1, generator=g).to(dev), torch.rand(P, generator=g).to(dev)
    tensors, (accum2, radii2) = prune_optimizer(opt, keep, extra=(accum, radii))
    assert torch.equal(accum2, accum[keep]) and torch.equal(radii2, radii[keep])
    for gr in opt.param_groups:
        p = gr["params"][0]
        assert tensors[gr["name"]] is p and isinstance(p, torch.nn.Parameter) and p.requires_grad
        p0, st0 = before[gr["name"]]
        assert torch.equal(p.detach(), p0[keep])
        st = opt.state[p]
        assert torch.equal(st["exp_avg"], st0["exp_avg"][keep]) and torch.equal(st["exp_avg_sq"], st0["exp_avg_sq"][keep])
        assert int(st["step"]) == 1 and len(opt.state) == len(opt.param_groups)
        p.grad = torch.ones_like(p)
    opt.step()
    assert all(int(opt.state[gr["params"][0]]["step"]) == 2 for gr in opt.param_groups)


def test_reset_opacity_caps_logits_and_restarts_moments(dev):
    """gaussian_model.py:347-352,451-464: opacities above 0.01 are capped, smaller ones kept, the group's Adam moments
    restart at zero under a new Parameter, every other group is untouched, and the optimizer keeps stepping."""
    from eogs2_amd.optim import FusedAdam, reset_opacity

    P = 5_000
    opt = FusedAdam(_groups(P, dev, seed=5), lr=0.0, eps=1e-15)
    for gr in opt.param_groups:
        gr["params"][0].grad = torch.ones_like(gr["params"][0])
    opt.step()
    others = {gr["name"]: gr["params"][0] for gr in opt.param_groups if gr["name"] != "opacity"}
    old = next(gr["params"][0] for gr in opt.param_groups if gr["name"] == "opacity").detach().clone()
    out = reset_opacity(opt)
    new = next(gr["params"][0] for gr in opt.param_groups if gr["name"] == "opacity")
    assert out["opacity"] is new and new.requires_grad and len(opt.state) == len(opt.param_groups)
    expect = torch.min(torch.sigmoid(old), torch.full_like(old, 0.01))
    assert torch.allclose(torch.sigmoid(new.detach()), expect, rtol=1e-5, atol=1e-8)
    assert float(torch.sigmoid(new.detach()).max()) <= 0.01 * (1 + 1e-5)
    st = opt.state[new]
    assert float(st["exp_avg"].abs().max()) == 0.0 and float(st["exp_avg_sq"].abs().max()) == 0.0 and int(st["step"]) == 1
    for gr in opt.param_groups:
        if gr["name"] != "opacity":
            assert gr["params"][0] is others[gr["name"]]
        gr["params"][0].grad = torch.ones_like(gr["params"][0])
    opt.step()
    assert int(opt.state[new]["step"]) == 2


def test_grad_bucket_pack_kernel_equals_cat(dev):
    """The data-parallel bucket's blocks against torch.cat over the same column slices (the partially bucketed colour
    block goes through eogs_pack_columns), and both directions of that C-ABI call against slicing."""
    import ctypes

    from eogs2_amd import _lib
    from eogs2_amd._abi import PackTensor
    from eogs2_amd.parallel import GradBucket

    P = 70_001
    g = torch.Generator().manual_seed(11)
    widths, cols = (3, 5, 1, 3, 4), [slice(0, 3), slice(0, 3), slice(0, 1), slice(0, 3), slice(0, 4)]
    params = [torch.zeros(P, w, device=dev, requires_grad=True) for w in widths]
    for p in params:
        p.grad = torch.randn(p.shape, generator=g).to(dev)
    want = torch.cat([p.grad[:, c] for p, c in zip(params, cols)], dim=1)
    # the bucket: one contiguous [P, k] block per parameter (the colour block is filled by eogs_pack_columns)
    b = GradBucket(params, cols=cols)
    assert b.bytes_per_gaussian == 56
    b.pack()
    o = 0
    for i, c in enumerate(cols):
        n = c.stop - c.start
        assert torch.equal(b.block(i), want[:, o:o + n]), i
        o += n
    b.unpack()  # no process group: the sums are the gradients themselves
    for i, (p, c) in enumerate(zip(params, cols)):
        assert b._is_block(p.grad, i) == (i != 1)
    assert torch.equal(torch.cat([p.grad[:, c] for p, c in zip(params, cols)], dim=1), want)
    # a missing gradient counts as zeros
    params[2].grad = None
    b2 = GradBucket(params, cols=cols)
    b2.pack()
    assert float(b2.block(2).abs().max()) == 0.0 and torch.equal(b2.block(4), want[:, 10:14])
    # unpack direction of the C-ABI
    abi = _lib.get()
    outs = [torch.full((P, w), 7.0, device=dev) for w in widths]
    arr = (PackTensor * 5)()
    for a, o, c in zip(arr, outs, cols):
        a.data, a.width, a.col0, a.ncols = o.data_ptr(), o.shape[1], c.start, c.stop - c.start
    abi.check(abi.pack_columns(P, 5, ctypes.cast(arr, ctypes.c_void_p), ctypes.c_void_p(want.data_ptr()), 14, 1,
                               ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    o = 0
    for out, c, w in zip(outs, cols, widths):
        n = c.stop - c.start
        assert torch.equal(out[:, c], want[:, o:o + n])
        if n < w:
            assert torch.equal(out[:, n:], torch.full((P, w - n), 7.0, device=dev))  # other columns untouched
        o += n


def _reference_densify(opt, mask, split, N, tmp_radii):
    """gaussian_model.py:507-660 in plain PyTorch ops (boolean-mask indexing, torch.cat), on this optimizer."""
    from eogs2_amd.optim import build_rotation

    par = {g["name"]: g["params"][0] for g in opt.param_groups}
    if split:
        stds = torch.exp(par["scaling"])[mask].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=stds.device), std=stds)
        rots = build_rotation(par["rotation"][mask]).repeat(N, 1, 1)
        new = {"xyz": torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + par["xyz"][mask].repeat(N, 1),
               "scaling": torch.log(torch.exp(par["scaling"])[mask].repeat(N, 1) / (0.8 * N)),
               "rotation": par["rotation"][mask].repeat(N, 1), "f_dc": par["f_dc"][mask].repeat(N, 1, 1),
               "f_rest": par["f_rest"][mask].repeat(N, 1, 1), "opacity": par["opacity"][mask].repeat(N, 1)}
        radii = torch.cat((tmp_radii, tmp_radii[mask].repeat(N)))
    else:
        new = {n: par[n][mask] for n in par}
        radii = torch.cat((tmp_radii, tmp_radii[mask]))
    out = {}
    for g in opt.param_groups:  # cat_tensors_to_optimizer
        p, ext = g["params"][0], new[g["name"]]
        st = opt.state[p]
        out[g["name"]] = (torch.cat((p.data, ext)), torch.cat((st["exp_avg"], torch.zeros_like(ext))),
                          torch.cat((st["exp_avg_sq"], torch.zeros_like(ext))))
    if split:  # prune_points(cat(mask, zeros))
        keep = ~torch.cat((mask, torch.zeros(N * int(mask.sum()), dtype=torch.bool, device=mask.device)))
        out = {n: tuple(t[keep] for t in v) for n, v in out.items()}
    return out, radii


@pytest.mark.parametrize("split", [False, True])
def test_densify_clone_and_split_match_reference_ops(dev, split):
    """Clone / split densification (gaussian_model.py:573-660; off by default in the reference) through the compaction
    primitives: same tensors, same moments, same random draw as the reference's op sequence."""
    from eogs2_amd.optim import FusedAdam, densify_and_clone, densify_and_split

    P, N = 20_000, 2
    opt = FusedAdam(_groups(P, dev, seed=5), lr=0.0, eps=1e-15)
    g = torch.Generator().manual_seed(6)
    for gr in opt.param_groups:
        gr["params"][0].grad = torch.randn(gr["params"][0].shape, generator=g).to(dev)
    opt.step()
    mask = (torch.rand(P, generator=g) < 0.15).to(dev)
    radii = torch.rand(P, generator=g).to(dev)
    torch.manual_seed(123)
    want, want_radii = _reference_densify(opt, mask, split, N, radii)
    torch.manual_seed(123)
    if split:
        params, got_radii, keep = densify_and_split(opt, mask, N=N, tmp_radii=radii)
        assert int(keep.sum()) == P - int(mask.sum()) + N * int(mask.sum())
    else:
        params, got_radii = densify_and_clone(opt, mask, tmp_radii=radii)
    assert torch.equal(got_radii, want_radii)
    for gr in opt.param_groups:
        p = gr["params"][0]
        st = opt.state[p]
        w = want[gr["name"]]
        assert params[gr["name"]] is p and p.requires_grad and tuple(p.shape) == tuple(w[0].shape)
        assert torch.equal(p.detach(), w[0]) and torch.equal(st["exp_avg"], w[1]) and torch.equal(st["exp_avg_sq"], w[2])
        p.grad = torch.ones_like(p)
    opt.step()  # training goes on over the new rows
    assert all(int(opt.state[gr["params"][0]]["step"]) == 2 for gr in opt.param_groups)


def test_retired_rows_render_nothing_and_stay_retired(dev):
    """`retire_rows` (the deferred prune): a retired Gaussian is listed nowhere and gets zero gradients, the survivors' render and
    gradients are those of the compacted model bit for bit, FusedAdam steps with momentum in the retired rows leave them retired,
    and `prune_optimizer(alive_rows())` removes exactly them."""
    from eogs2_amd.fused import rasterize_raw
    from eogs2_amd.optim import RETIRED_LOGIT, FusedAdam, alive_rows, prune_optimizer, retire_rows
    from eogs2_amd.synthetic import make_scene, settings_for

    P, H, W = 20000, 160, 192
    sc = make_scene(P, H, W, seed=5, opacity="trained", device=dev)
    rs, alt = settings_for(sc, H, W), sc["viewmatrix"][:, 2].contiguous()
    dL = torch.randn(5, H, W, device=dev) / (H * W)
    names = ("xyz", "f_dc", "opacity", "scaling", "rotation")
    init = dict(xyz=sc["means3D"], f_dc=torch.logit(sc["colors"][:, :3].clamp(0.01, 0.99)).reshape(P, 1, 3),
                opacity=torch.logit(sc["opacities"].clamp(1e-4, 1 - 1e-4)), scaling=sc["scales"].log(), rotation=sc["rotations"])
    opt = FusedAdam([{"params": [torch.nn.Parameter(init[n].clone())], "lr": 1e-2, "name": n} for n in names], lr=0.0, eps=1e-15)

    def params():
        return {g["name"]: g["params"][0] for g in opt.param_groups}

    def fwd_bwd(p):
        for t in p.values():
            t.grad = None
        m2 = torch.zeros_like(p["xyz"], requires_grad=True)
        color, radii, _ = rasterize_raw(p["xyz"], m2, p["f_dc"], p["opacity"], p["scaling"], p["rotation"], alt, rs)
        torch.autograd.backward([color], [dL])
        return color.detach().clone()

    fwd_bwd(params())
    opt.step()  # every row now carries Adam momentum
    keep = torch.rand(P, generator=torch.Generator().manual_seed(3)).to(dev) > 0.3
    retire_rows(opt, keep)
    p = params()
    assert bool((p["opacity"].view(-1)[~keep] == RETIRED_LOGIT).all()) and bool(torch.equal(alive_rows(opt), keep))
    color = fwd_bwd(p)
    for n in names:
        assert float(p[n].grad[~keep].abs().max()) == 0.0, n
    # the compacted model: same image, same gradients in the surviving rows
    compact = {n: torch.nn.Parameter(p[n].detach()[keep].clone()) for n in names}
    color_c = fwd_bwd(compact)
    assert torch.equal(color, color_c)
    for n in names:
        assert torch.equal(p[n].grad[keep], compact[n].grad), n
    for _ in range(5):  # momentum from before the retirement, zero gradients since
        opt.step()
    assert bool(torch.equal(alive_rows(opt), keep))
    new, _ = prune_optimizer(opt, alive_rows(opt))
    assert new["xyz"].shape[0] == int(keep.sum())
    assert torch.isfinite(new["opacity"]).all()


def test_sum_into_equals_the_sequence_of_adds(dev):
    """eogs_sum_into (include/eogs_optim.h): dst += s0; dst += s1; ... for several tensors in one launch — the additions autograd
    makes render by render, bit for bit, odd sizes and unaligned views included; `Branches.run(shared=...)` sums its pieces with it."""
    from eogs2_amd.optim import sum_into_

    g = torch.Generator(device="cpu").manual_seed(3)
    shapes = [(1000, 3), (1000, 1, 3), (1000, 1), (777, 4), (5,), (0, 3)]
    base = [torch.randn(s, generator=g).to(dev) for s in shapes]
    srcs = [[torch.randn(s, generator=g).to(dev) * (10.0 ** j) for s in shapes] for j in range(3)]
    # an unaligned destination / source (a view one float into a larger buffer): the scalar path
    big = torch.randn(4001, generator=g).to(dev)
    base.append(big[1:])
    for j in range(3):
        srcs[j].append(torch.randn(4003, generator=g).to(dev)[3:])
    want = [b.clone() for b in base]
    for s in srcs:
        for w, x in zip(want, s):
            w.add_(x)
    got = [b.clone() for b in base[:-1]] + [big.clone()[1:]]
    sum_into_(got, srcs)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        sum_into_([base[0]], [[base[3]]])  # sizes differ
    sum_into_([], [])  # nothing to do
