"""Step time (fwd + bwd of one view) eager against a replayed HIP graph, with the kernel sum beside it (on the GPU box).
usage: python tools/graph_probe.py P size [opacity] ..."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from eogs2_amd import GaussianRasterizer, _lib
from eogs2_amd.graph import GraphedStep
from eogs2_amd.synthetic import make_scene, settings_for

dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
abi = _lib.get()
NAMES = ("means3D", "colors", "opacities", "scales", "rotations")


def run(P, S, op):
    sc = make_scene(P, S, S, seed=0, opacity=op, device=dev)
    rast = GaussianRasterizer(settings_for(sc, S, S))
    params = {k: sc[k].clone().requires_grad_(True) for k in NAMES}
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)

    def step():
        for p in params.values(): p.grad = None
        m2.grad = None
        color, _, _ = rast(params["means3D"], m2, params["opacities"], colors_precomp=params["colors"], scales=params["scales"], rotations=params["rotations"])
        torch.autograd.backward([color], [sc["dL_dcolor"]])
        return color

    def timeit(f, n=300):
        for _ in range(200): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

    eager = timeit(step)
    abi.check(abi.profile_enable(1)); abi.check(abi.profile_reset())
    for _ in range(50): step()
    torch.cuda.synchronize()
    ksum = sum(ms / n for ms, n in abi.profile().values() if n)
    abi.check(abi.profile_enable(0))
    g = GraphedStep(step, warmup=2)
    checked = timeit(g)            # replay + count check (one wait per step)
    unchecked = timeit(g.replay)   # replay only
    assert g.fits() and g.recaptures == 0
    print(f"{P} x {S}^2 {op}: kernels {ksum:.3f} ms | eager {eager:.3f} | graph + check {checked:.3f} | graph replay only {unchecked:.3f}", flush=True)


args = sys.argv[1:] or ["100000", "512", "init"]
i = 0
while i < len(args):
    run(int(args[i]), int(args[i + 1]), args[i + 2] if i + 2 < len(args) and not args[i + 2].isdigit() else "trained")
    i += 3 if i + 2 < len(args) and not args[i + 2].isdigit() else 2
