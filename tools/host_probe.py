import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from eogs2_amd import GaussianRasterizer
from eogs2_amd.parallel import GradBucket
from eogs2_amd.synthetic import make_scene, settings_for
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
P, H, W = 1 << 20, 1024, 1024
sc = make_scene(P, H, W, seed=0, opacity="init", device=dev)
rs = settings_for(sc, H, W)
names = ("means3D", "colors", "opacities", "scales", "rotations")
params = {k: sc[k].clone().requires_grad_(True) for k in names}
m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
bucket = GradBucket([params[k] for k in names], cols=[slice(0, 3), slice(0, 3), slice(0, 1), slice(0, 3), slice(0, 4)], names=names)
rast = GaussianRasterizer(rs)
T = {}
def acc(k, t0):
    T[k] = T.get(k, 0.0) + time.perf_counter() - t0
def step(mode):
    t0 = time.perf_counter()
    m2.grad = None
    if mode: bucket.begin()
    else:
        for p in params.values(): p.grad = None
    acc("begin", t0); t0 = time.perf_counter()
    color, _, _ = rast(params["means3D"], m2, params["opacities"], colors_precomp=params["colors"], scales=params["scales"], rotations=params["rotations"])
    acc("forward", t0); t0 = time.perf_counter()
    torch.autograd.backward([color], [sc["dL_dcolor"]])
    acc("backward", t0); t0 = time.perf_counter()
    if mode: bucket.finish()
    acc("finish", t0)
for mode in (0, 1, 0, 1):
    for _ in range(200): step(mode)
    torch.cuda.synchronize(); T.clear(); t = time.perf_counter()
    for _ in range(200): step(mode)
    torch.cuda.synchronize(); tot = (time.perf_counter() - t) / 200 * 1e3
    print("mode", mode, "step ms %.3f" % tot, {k: round(v / 200 * 1e6, 1) for k, v in T.items()}, "(host us)")
