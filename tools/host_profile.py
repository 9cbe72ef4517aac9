"""Host-side profile of the drop-in step at a small scene (100 k / 512^2: the device needs 0.17 ms, the host more):
cProfile over 300 steps, top functions by cumulative time. On the GPU box: python tools/host_profile.py"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eogs2_amd import GaussianRasterizer
from eogs2_amd.synthetic import make_scene, settings_for
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
P, H, W = int(os.environ.get("HP_P", 100000)), int(os.environ.get("HP_S", 512)), int(os.environ.get("HP_S", 512))
sc = make_scene(P, H, W, seed=0, opacity="init", device=dev)
rast = GaussianRasterizer(settings_for(sc, H, W))
names = ("means3D", "colors", "opacities", "scales", "rotations")
params = {k: sc[k].clone().requires_grad_(True) for k in names}
m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
def step():
    for p in params.values(): p.grad = None
    m2.grad = None
    color, _, _ = rast(params["means3D"], m2, params["opacities"], colors_precomp=params["colors"], scales=params["scales"], rotations=params["rotations"])
    torch.autograd.backward([color], [sc["dL_dcolor"]])
for _ in range(100): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize(); print("step ms", (time.perf_counter() - t) / 300 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
