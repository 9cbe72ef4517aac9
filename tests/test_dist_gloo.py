"""CPU, world_size 2 over gloo: view-sharded data parallelism. Each rank renders its own view through the
host wrapper (oracle-backed here, test-only), the gradient bucket is all-reduced once, and the result equals
the sum of the per-view gradients computed in a single process."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, H, W = 300, 48, 64
NAMES = ("means3D", "colors", "opacities", "scales", "rotations")
COLS = [slice(0, 3), slice(0, 3), slice(0, 1), slice(0, 3), slice(0, 4)]


def _view_grads(view):
    import oracle
    from eogs2_amd import GaussianRasterizer, _lib
    from eogs2_amd.synthetic import make_camera, make_scene, settings_for

    product_get = _lib.get
    _lib.get = oracle.abi  # test-only checker backend (CPU tensors)
    try:
        sc = make_scene(P, H, W, seed=0, opacity="trained", scale_mult=3.0)
        sc["viewmatrix"] = make_camera(H, W, seed=view)
        params = {k: sc[k].clone().requires_grad_(True) for k in NAMES}
        color, radii, _ = GaussianRasterizer(settings_for(sc, H, W))(
            params["means3D"], torch.zeros(P, 3), params["opacities"], colors_precomp=params["colors"],
            scales=params["scales"], rotations=params["rotations"])
        (color * sc["dL_dcolor"]).sum().backward()
    finally:
        _lib.get = product_get
    return params, radii


def _overlapped_step(view, chunks, params=None):
    """One data-parallel step the way bench.py runs it: begin() arms the bucket, the backward writes into it and starts
    the exchange range by range (eogs_rast_backward_range; the checker library implements the same entry point),
    finish() waits."""
    import oracle
    from eogs2_amd import GaussianRasterizer, _lib
    from eogs2_amd.parallel import GradBucket
    from eogs2_amd.synthetic import make_camera, make_scene, settings_for

    product_get = _lib.get
    _lib.get = oracle.abi  # test-only checker backend (CPU tensors)
    try:
        sc = make_scene(P, H, W, seed=0, opacity="trained", scale_mult=3.0)
        sc["viewmatrix"] = make_camera(H, W, seed=view)
        if params is None:
            params = {k: sc[k].clone().requires_grad_(True) for k in NAMES}
        bucket = GradBucket([params[k] for k in NAMES], cols=COLS, names=NAMES, chunks=chunks)
        bucket.begin()
        color, _, _ = GaussianRasterizer(settings_for(sc, H, W))(
            params["means3D"], torch.zeros(P, 3), params["opacities"], colors_precomp=params["colors"],
            scales=params["scales"], rotations=params["rotations"])
        (color * sc["dL_dcolor"]).sum().backward()
        bucket.finish()
    finally:
        _lib.get = product_get
    return params, bucket


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from eogs2_amd.parallel import GradBucket, all_reduce_densification_stats, shard_views

    assert shard_views(5) == list(range(rank, 5, world))
    params, radii = _view_grads(rank)
    bucket = GradBucket([params[k] for k in NAMES], cols=COLS)
    assert bucket.bytes_per_gaussian == 56
    bucket.all_reduce()
    # the overlapped exchange, whole and in two Gaussian ranges: same sums, and full-width gradients live in the bucket
    ov = {}
    for chunks in (1, 2):
        p2, b2 = _overlapped_step(rank, chunks)
        assert b2.exchanges == chunks
        assert all(b2._placed), "every exchanged column is written by the backward itself (dL_dcolors_lead for colors)"
        for i, k in enumerate(NAMES):
            assert b2._is_block(p2[k].grad, i) == (k != "colors"), k
        ov[chunks] = {k: p2[k].grad.clone() for k in NAMES}
    # gradient accumulation over two steps with zero_grad(set_to_none=False): the gradients are views of the bucket by
    # then, autograd accumulates into them in place, and pack() must still refresh the partially bucketed parameter
    for k in NAMES:
        params[k].grad.zero_()
    params2, _ = _view_grads(rank)
    for k in NAMES:
        params[k].grad += params2[k].grad
    bucket.all_reduce()
    second = {k: params[k].grad.clone() for k in NAMES}
    acc, den, mr = torch.full((P, 1), float(rank + 1)), torch.ones(P, 1), radii.float()
    all_reduce_densification_stats(acc, den, mr)
    torch.save({"grads": second, "ov": ov, "acc": acc, "den": den, "mr": mr, "radii": radii},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_grad_allreduce(tmp_path):
    world, port = 2, 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    singles = [_view_grads(v) for v in range(world)]
    outs = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    for k, c in zip(NAMES, COLS):
        expect = sum(s[0][k].grad for s in singles)
        for r in range(world):
            for got in (outs[r]["grads"][k], outs[r]["ov"][1][k], outs[r]["ov"][2][k]):
                assert torch.allclose(got[:, c], expect[:, c], rtol=1e-6, atol=1e-9), (k, r)
                # columns outside the bucket (altitude / constant feature channels) stay rank-local
                if k == "colors":
                    assert torch.equal(got[:, 3:], singles[r][0][k].grad[:, 3:])
            assert torch.equal(outs[r]["ov"][1][k], outs[r]["ov"][2][k]), k  # ranges change nothing, bit for bit
    for r in range(world):
        assert torch.equal(outs[r]["acc"], torch.full((P, 1), 3.0))
        assert torch.equal(outs[r]["den"], torch.full((P, 1), 2.0))
        assert torch.equal(outs[r]["mr"], torch.maximum(singles[0][1], singles[1][1]).float())
    assert torch.equal(outs[0]["grads"]["means3D"], outs[1]["grads"]["means3D"])


def test_overlapped_path_refuses_other_gradient_sources():
    """The begin()/finish() path is only right when the bucketed gradients come from the one rasterizer backward: a second
    backward while armed raises at once, a loss term that autograd sums into a parameter's gradient raises in finish(),
    and the synchronous all_reduce() takes both cases."""
    import oracle
    from eogs2_amd import GaussianRasterizer, _lib
    from eogs2_amd.parallel import GradBucket
    from eogs2_amd.synthetic import make_camera, make_scene, settings_for

    product_get = _lib.get
    _lib.get = oracle.abi  # test-only checker backend (CPU tensors)
    try:
        sc = make_scene(P, H, W, seed=0, opacity="trained", scale_mult=3.0)
        sc["viewmatrix"] = make_camera(H, W, seed=0)
        params = {k: sc[k].clone().requires_grad_(True) for k in NAMES}
        rast = GaussianRasterizer(settings_for(sc, H, W))

        def render():
            return rast(params["means3D"], torch.zeros(P, 3), params["opacities"], colors_precomp=params["colors"],
                        scales=params["scales"], rotations=params["rotations"])[0]

        bucket = GradBucket([params[k] for k in NAMES], cols=COLS, names=NAMES)
        # (1) the plain single-backward step passes, twice (first step verified by value, second by identity)
        for _ in range(2):
            bucket.begin()
            (render() * sc["dL_dcolor"]).sum().backward()
            bucket.finish()
        ref = {k: params[k].grad.clone() for k in NAMES}
        # (2) a regulariser on the opacities in the same backward: autograd sums it into the gradient
        bucket.begin()
        ((render() * sc["dL_dcolor"]).sum() + params["opacities"].sum()).backward()
        with pytest.raises(RuntimeError, match="another loss term"):
            bucket.finish()
        # (3) a second render + backward while armed
        bucket.begin()
        (render() * sc["dL_dcolor"]).sum().backward()
        with pytest.raises(RuntimeError, match="second gradient"):
            (render() * sc["dL_dcolor"]).sum().backward()
        bucket._armed = False
        from eogs2_amd.rasterizer import set_backward_plan
        set_backward_plan(None)
        # (4) the synchronous path takes any number of sources
        for p in params.values():
            p.grad = None
        for _ in range(2):
            ((render() * sc["dL_dcolor"]).sum() + params["opacities"].sum()).backward()
        bucket.all_reduce()
        assert torch.allclose(params["means3D"].grad, 2 * ref["means3D"], rtol=1e-6, atol=1e-12)
        assert torch.allclose(params["opacities"].grad, 2 * ref["opacities"] + 2, rtol=1e-6, atol=1e-9)
        bucket.close()
    finally:
        _lib.get = product_get


@pytest.mark.parametrize("n", [2, 4])
def test_bench_launcher_starts_its_own_ranks(n):
    """`python3 bench.py --gpus N` with no launcher around it: the parent starts N ranks before any GPU call and
    forwards rank 0's line (--dry-run stops the ranks after a gloo rendezvous + all-reduce, so this runs without a GPU)."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line == {"dry_run": True, "n_gpus": n, "world_size": n, "ranks_seen": n, "launcher": "self"}


def test_chunk_ranges():
    from eogs2_amd.rasterizer import chunk_ranges

    for P in (0, 1, 255, 256, 257, 1000, 1 << 20, (1 << 20) + 3):
        for k in (1, 2, 4, 7, 64):
            r = chunk_ranges(P, k)
            assert r[0][0] == 0 and r[-1][1] == P and len(r) <= max(k, 1)
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert all(p0 % 256 == 0 for p0, _ in r) and all(p1 % 256 == 0 for _, p1 in r[:-1])


def test_bench_launcher_ends_all_ranks_when_one_dies(tmp_path, monkeypatch):
    """A rank that dies must not leave the others waiting inside a rendezvous / collective: the launcher polls its
    children and ends the ones it started (by PID)."""
    import time
    import types

    sys.path.insert(0, ROOT)
    import bench

    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\nsys.exit(7) if os.environ.get('RANK') == '1' else time.sleep(600)\n")
    monkeypatch.setattr(bench, "__file__", str(script))
    t0 = time.time()
    with pytest.raises(SystemExit) as e:
        bench.launch_ranks(types.SimpleNamespace(gpus=2), [])
    assert e.value.code != 0 and time.time() - t0 < 60


def test_view_sharded_schedule():
    """The batch-N protocol of DESIGN.md 7: same number of views before every scheduled event, sqrt(N) learning rates."""
    import math

    sys.path.insert(0, ROOT)
    from eogs2_amd.parallel import view_sharded_schedule

    opt = {"iterations": 10000, "position_lr_init": 1.6e-4, "position_lr_final": 1.6e-6, "position_lr_max_steps": 30000,
           "feature_lr": 0.0025, "densify_from_iter": 500, "densification_interval": 100, "opacity_reset_interval": 3000,
           "iterend_opacity_reset_interval": 999999999, "lambda_dssim": 0.2, "percent_dense": 0.01, "only_prune": True,
           "densify_until_iter": -1}
    assert view_sharded_schedule(opt, 1) == dict(opt, grad_average=False)
    s8 = view_sharded_schedule(opt, 8)
    assert (s8["iterations"], s8["densify_from_iter"], s8["densification_interval"], s8["opacity_reset_interval"],
            s8["position_lr_max_steps"]) == (1250, 62, 12, 375, 3750)
    assert s8["densify_until_iter"] == -1 and s8["lambda_dssim"] == 0.2 and s8["only_prune"] is True and s8["grad_average"]
    assert math.isclose(s8["position_lr_init"], 1.6e-4 * math.sqrt(8)) and math.isclose(s8["feature_lr"], 0.0025 * math.sqrt(8))
    assert s8["iterend_opacity_reset_interval"] == 999999999  # the reference's "never" stays never
    # views seen before each event differ from the reference's by less than one step's worth of views
    for k in ("iterations", "densify_from_iter", "densification_interval", "opacity_reset_interval"):
        assert abs(s8[k] * 8 - opt[k]) < 8, k
    # the reference's own layout (gs_config/train.yaml:87-130): densification knobs one level down, per-loss thresholds
    nested = {"iterations": 10000, "position_lr_init": 1.6e-4, "opacity_reset_interval": 3000,
              "densification_strategy": {"densify_from_iter": 500, "densification_interval": 100, "densify_grad_threshold": 2e-6},
              "iterstart_shadowmapping": 1000, "iterstart_L_new_resample": 1000, "iterstart_L_opacity": -1,
              "iterstart_L_erank": 9999999999, "iterend_L_opacity": 99999999}
    n8 = view_sharded_schedule(nested, 8, strict=True)
    assert n8["densification_strategy"] == {"densify_from_iter": 62, "densification_interval": 12, "densify_grad_threshold": 2e-6}
    assert (n8["iterstart_shadowmapping"], n8["iterstart_L_new_resample"], n8["iterstart_L_opacity"]) == (125, 125, -1)
    assert n8["iterstart_L_erank"] == 9999999999 and n8["iterend_L_opacity"] == 99999999
    assert nested["densification_strategy"]["densify_from_iter"] == 500  # the input is not modified
    with pytest.raises(RuntimeError, match="lack"):
        view_sharded_schedule({"feature_lr": 0.0025}, 8, strict=True)
    with pytest.warns(UserWarning, match="lack"):
        view_sharded_schedule({"feature_lr": 0.0025}, 8)
