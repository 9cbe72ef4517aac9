"""Tuning aid (GPU box): one regime of bench.regime_scan with the per-kernel-group breakdown, to tell a host effect from a kernel effect.
usage: python tools/regime_probe.py <P> <size> <opacity> [steps]"""
import os, sys, time, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from eogs2_amd import GaussianRasterizer, _lib
from eogs2_amd.synthetic import make_scene, settings_for

P, S = int(sys.argv[1]), int(sys.argv[2])
try:
    op = float(sys.argv[3])
except ValueError:
    op = sys.argv[3]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda:0")
abi = _lib.get()
sc = make_scene(P, S, S, seed=0, opacity=op, device=dev)
rast = GaussianRasterizer(settings_for(sc, S, S))
params = {k: sc[k].clone().requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
m2 = torch.zeros(P, 3, device=dev, requires_grad=True)

def step():
    for p in params.values():
        p.grad = None
    m2.grad = None
    c, _, _ = rast(params["means3D"], m2, params["opacities"], colors_precomp=params["colors"], scales=params["scales"], rotations=params["rotations"])
    torch.autograd.backward([c], [sc["dL_dcolor"]])
    return c

for rep in range(3):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    abi.profile_select(0xFFFFFFFF); abi.profile_reset(); abi.profile_enable(1)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    abi.profile_enable(0)
    prof = {k: round(v[0] / steps, 4) for k, v in abi.profile().items() if v[1]}
    from eogs2_amd.rasterizer import last_exact_token
    tok = int(last_exact_token(dev))
    nblocks = ((S + 31) // 32) ** 2
    print(json.dumps({"rep": rep, "ms_per_step": round(ms, 4), "kernel_sum": round(sum(prof.values()), 4),
                      "entries_per_block": round(((tok >> 32) & 0x07FFFFFF) / nblocks, 1), "block_lists": (tok >> 62) & 1, "kernels": prof}), flush=True)
