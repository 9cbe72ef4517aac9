"""Host time per C-ABI call of a training step, with and without the deferred count readback (on the GPU box).
usage: python tools/spec_probe.py [P] [size] [opacity]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from eogs2_amd import GaussianRasterizer, rasterizer
from eogs2_amd.synthetic import make_scene, settings_for

P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
op = sys.argv[3] if len(sys.argv) > 3 else "trained"
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
sc = make_scene(P, S, S, seed=0, opacity=op, device=dev)
rs = settings_for(sc, S, S)
names = ("means3D", "colors", "opacities", "scales", "rotations")
params = {k: sc[k].clone().requires_grad_(True) for k in names}
m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
rast = GaussianRasterizer(rs)
T, N = {}, {}


class Timed:
    def __init__(self, abi): self._abi = abi
    def __getattr__(self, name):
        f = getattr(self._abi, name)
        if not callable(f) or name in ("check",): return f
        def g(*a):
            t0 = time.perf_counter(); r = f(*a); dt = time.perf_counter() - t0
            T[name] = T.get(name, 0.0) + dt; N[name] = N.get(name, 0) + 1
            return r
        return g


real = rasterizer._backend()
rasterizer._backend = lambda: Timed(real)


def step():
    for p in params.values(): p.grad = None
    m2.grad = None
    color, _, _ = rast(params["means3D"], m2, params["opacities"], colors_precomp=params["colors"], scales=params["scales"], rotations=params["rotations"])
    torch.autograd.backward([color], [sc["dL_dcolor"]])


for spec in (False, True, False, True):
    rasterizer.set_speculation(spec, forget=True)
    for _ in range(300): step()
    torch.cuda.synchronize(); T.clear(); N.clear(); t = time.perf_counter()
    n = 300
    for _ in range(n): step()
    torch.cuda.synchronize(); tot = (time.perf_counter() - t) / n * 1e3
    print("speculate", spec, "step ms %.3f" % tot, {k: round(v / n * 1e6, 1) for k, v in T.items()}, "(host us per step)", flush=True)
