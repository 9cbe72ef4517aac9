"""Where the one-rank cost of the data-parallel step goes (on the GPU box): the bench step (a) plain, (b) with the
bucket armed but no process group (gradient placement + column copies only), (c) with RCCL initialised (one rank).
usage: python tools/dp_probe.py [chunks]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch.distributed as dist
from eogs2_amd import GaussianRasterizer
from eogs2_amd.parallel import GradBucket
from eogs2_amd.synthetic import make_scene, settings_for

chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
P, H, W = 1 << 20, 1024, 1024
sc = make_scene(P, H, W, seed=0, opacity="init", device=dev)
rs = settings_for(sc, H, W)
names = ("means3D", "colors", "opacities", "scales", "rotations")
params = {k: sc[k].clone().requires_grad_(True) for k in names}
m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
bucket = GradBucket([params[k] for k in names], cols=[slice(0, 3), slice(0, 3), slice(0, 1), slice(0, 3), slice(0, 4)],
                    names=names, chunks=chunks)
rast = GaussianRasterizer(rs)

def step(mode):
    m2.grad = None
    if mode: bucket.begin()
    else:
        for p in params.values(): p.grad = None
    color, _, _ = rast(params["means3D"], m2, params["opacities"], colors_precomp=params["colors"], scales=params["scales"], rotations=params["rotations"])
    torch.autograd.backward([color], [sc["dL_dcolor"]])
    if mode: bucket.finish()

def timeit(mode, n=200):
    for _ in range(300): step(mode)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step(mode)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

print("plain            ms", timeit(0))
print("bucket, no dist  ms", timeit(1))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for label, kw in (("bucket + rccl(1), async handle", dict(async_whole=True)), ("bucket + rccl(1), issued in line", dict(async_whole=False))):
    bucket.close()
    bucket = GradBucket([params[k] for k in names], cols=[slice(0, 3), slice(0, 3), slice(0, 1), slice(0, 3), slice(0, 4)],
                        names=names, chunks=chunks, **kw)
    print(label, "ms", timeit(1))
print("plain again      ms", timeit(0))
dist.destroy_process_group()
