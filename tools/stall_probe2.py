"""Tuning aid (GPU box): are the periodic stalls of the eager loop (tools/stall_probe.py) stalls of the DEVICE?  The same step as a
replayed HIP graph, queued back to back without any host wait (3000 replays, one synchronize at the end): if the device stopped for 4-5 ms
ten times a second, 2 s of saturated queue would take ~4 % longer than replays x the replay time measured in short windows.
usage: python tools/stall_probe2.py"""
import json, os, statistics, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from eogs2_amd import GaussianRasterizer
from eogs2_amd.graph import GraphedStep
from eogs2_amd.synthetic import make_scene, settings_for

P, S = 1 << 20, 1024
dev = torch.device("cuda:0")
sc = make_scene(P, S, S, seed=0, opacity="init", device=dev)
rast = GaussianRasterizer(settings_for(sc, S, S))
params = {k: sc[k].clone().requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
m2 = torch.zeros(P, 3, device=dev, requires_grad=True)


def step():
    for p in params.values():
        p.grad = None
    m2.grad = None
    c, _, _ = rast(params["means3D"], m2, params["opacities"], colors_precomp=params["colors"], scales=params["scales"], rotations=params["rotations"])
    torch.autograd.backward([c], [sc["dL_dcolor"]])
    return c.detach()


for _ in range(300):
    step()
torch.cuda.synchronize()
# eager: per-step intervals (is this a box with stalls?)
ts = []
for _ in range(1500):
    t0 = time.perf_counter(); step(); ts.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
eager = {"median": round(statistics.median(ts), 4), "mean": round(sum(ts) / len(ts), 4), "over_2ms": [(i, round(t, 2)) for i, t in enumerate(ts) if t > 2.0][:12]}
gs = GraphedStep(step, warmup=2)
for _ in range(20):
    gs()
torch.cuda.synchronize()
# short windows of 20 replays with a check each (the usual use): the replay time
w = []
for _ in range(15):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        gs()
    torch.cuda.synchronize(); w.append((time.perf_counter() - t0) / 20 * 1e3)
# saturated queue: replays only, no host wait in between
N = 3000
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N):
    gs.replay()
t_queue = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(json.dumps({"eager": eager, "replay_window_ms": {"median": round(statistics.median(w), 4), "min": round(min(w), 4), "max": round(max(w), 4)},
                  "saturated": {"replays": N, "ms_per_replay": round(t_all / N * 1e3, 4), "host_queueing_s": round(t_queue, 3), "total_s": round(t_all, 3)}}))
