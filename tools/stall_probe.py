"""Tuning aid (GPU box): what stalls the eager step's host thread?  2000 eager steps at the headline; per-step host intervals, the
process's context switches, and the cgroup's CPU throttling counters before and after.
usage: python tools/stall_probe.py"""
import glob, json, os, resource, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from eogs2_amd import GaussianRasterizer
from eogs2_amd.synthetic import make_scene, settings_for


def cpu_stat():
    out = {}
    for f in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat"):
        if os.path.exists(f):
            for ln in open(f):
                k, v = ln.split()
                out[k] = int(v)
            out["file"] = f
            break
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        if os.path.exists(f):
            out["quota"] = open(f).read().strip()
    return out


def ctx():
    r = resource.getrusage(resource.RUSAGE_SELF)
    return {"vol": r.ru_nvcsw, "invol": r.ru_nivcsw, "utime": r.ru_utime, "stime": r.ru_stime}


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


for _ in range(200):
    step()
torch.cuda.synchronize()
print("threads in process:", len(os.listdir("/proc/self/task")), "cpus allowed:", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
print("loadavg", open("/proc/loadavg").read().strip())
s0, c0 = cpu_stat(), ctx()
ts = []
t00 = time.perf_counter()
for _ in range(2000):
    t0 = time.perf_counter()
    step()
    ts.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
wall = time.perf_counter() - t00
s1, c1 = cpu_stat(), ctx()
ts_sorted = sorted(ts)
print(json.dumps({"ms_per_step": round(wall / 2000 * 1e3, 4), "median": round(ts_sorted[1000], 4), "p99": round(ts_sorted[1980], 4), "max": round(ts_sorted[-1], 3),
                  "stalls_over_2ms": [(i, round(t, 2)) for i, t in enumerate(ts) if t > 2.0][:20],
                  "ctx_switches": {k: round(c1[k] - c0[k], 3) for k in c0},
                  "cgroup": {k: (s1[k] - s0[k] if isinstance(s1.get(k), int) else s1.get(k)) for k in s1}}))
