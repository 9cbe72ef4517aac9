"""Tuning aid (GPU box): fwd + bwd of one view through the raw-parameter front end (eogs2_amd.fused.rasterize_raw), with the library's
per-kernel-group timers. usage: python tools/raw_step_probe.py [P] [size] [opacity] [steps]"""
import json, os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eogs2_amd import _lib
from eogs2_amd.fused import rasterize_raw
from eogs2_amd.synthetic import make_scene, settings_for

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
op = sys.argv[3] if len(sys.argv) > 3 else "init"
try:
    op = float(op)
except ValueError:
    pass
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
dev = torch.device("cuda:0")
abi = _lib.get()
sc = make_scene(P, S, S, seed=0, opacity=op, device=dev)
rs = settings_for(sc, S, S)
C0 = 0.28209479177387814
o = sc["opacities"].double()
leaves = [sc["means3D"].clone(), ((sc["colors"][:, :3] - 0.5) / C0).contiguous(), torch.log(o / (1 - o)).float(), torch.log(sc["scales"]), sc["rotations"] * 1.5]
for v in leaves:
    v.requires_grad_(True)
alt = sc["viewmatrix"][:, 2].contiguous()
m2 = torch.zeros(P, 3, device=dev, requires_grad=True)


def step():
    for p in leaves:
        p.grad = None
    m2.grad = None
    c, _, _ = rasterize_raw(leaves[0], m2, leaves[1], leaves[2], leaves[3], leaves[4], alt, rs)
    torch.autograd.backward([c], [sc["dL_dcolor"]])


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
    print(json.dumps({"rep": rep, "ms_per_step": round(ms, 4), "kernels": prof}), flush=True)
