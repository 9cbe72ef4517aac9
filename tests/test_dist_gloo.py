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
    acc, den, mr = torch.full((P, 1), float(rank + 1)), torch.ones(P, 1), radii.float()
    all_reduce_densification_stats(acc, den, mr)
    torch.save({"grads": {k: params[k].grad for k in NAMES}, "acc": acc, "den": den, "mr": mr, "radii": radii},
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
            got = outs[r]["grads"][k]
            assert torch.allclose(got[:, c], expect[:, c], rtol=1e-6, atol=1e-9), (k, r)
            # columns outside the bucket (altitude / constant feature channels) stay rank-local
            if k == "colors":
                assert torch.equal(got[:, 3:], singles[r][0][k].grad[:, 3:])
    for r in range(world):
        assert torch.equal(outs[r]["acc"], torch.full((P, 1), 3.0))
        assert torch.equal(outs[r]["den"], torch.full((P, 1), 2.0))
        assert torch.equal(outs[r]["mr"], torch.maximum(singles[0][1], singles[1][1]).float())
    assert torch.equal(outs[0]["grads"]["means3D"], outs[1]["grads"]["means3D"])
