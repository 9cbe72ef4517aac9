"""Tuning aid (GPU box): are the host hiccups of the eager step Python's cyclic GC?  Times every collection (gc.callbacks) during
600 eager steps at the headline, lists the steps whose host time exceeds twice the median, then repeats with gc.freeze() + gc.disable().
usage: python tools/gc_probe.py"""
import gc, os, sys, time, json, statistics, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench  # the same module weight (imports) as the driver's process  # noqa: F401
from eogs2_amd import GaussianRasterizer
from eogs2_amd.synthetic import make_scene, settings_for

P, S = 1 << 20, 1024
dev = torch.device("cuda:0")
sc = make_scene(P, S, S, seed=0, opacity="init", device=dev)
rast = GaussianRasterizer(settings_for(sc, S, S))
params = {k: sc[k].clone().requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
coll = []
_t = [0.0]
def cb(phase, info):
    if phase == "start":
        _t[0] = time.perf_counter()
    else:
        coll.append((info["generation"], (time.perf_counter() - _t[0]) * 1e3))
gc.callbacks.append(cb)

def step():
    for p in params.values():
        p.grad = None
    m2.grad = None
    c, _, _ = rast(params["means3D"], m2, params["opacities"], colors_precomp=params["colors"], scales=params["scales"], rotations=params["rotations"])
    torch.autograd.backward([c], [sc["dL_dcolor"]])

def run(tag, n=600):
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    coll.clear()
    ts = []
    t00 = time.perf_counter()
    for _ in range(n):
        t0 = time.perf_counter()
        step()
        ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t00) / n * 1e3
    med = statistics.median(ts)
    slow = [(i, round(t, 2)) for i, t in enumerate(ts) if t > 2 * med]
    # worst 20-step window (what a --steps 20 record would show)
    w20 = max(sum(ts[i:i + 20]) / 20 for i in range(0, n - 20))
    print(json.dumps({"tag": tag, "ms_per_step": round(tot, 4), "median_host_ms": round(med, 4), "worst_20_step_window_ms": round(w20, 4),
                      "steps_over_2x_median": slow[:20], "gc_collections": {g: [len([1 for x in coll if x[0] == g]), round(sum(x[1] for x in coll if x[0] == g), 2), round(max([x[1] for x in coll if x[0] == g] or [0]), 2)] for g in (0, 1, 2)},
                      "gc_objects": len(gc.get_objects())}), flush=True)

run("gc on")
run("gc on (2)")
gc.collect(); gc.freeze()
run("gc frozen")
gc.disable()
run("gc frozen+disabled")
run("gc frozen+disabled (2)")
