"""GPU box: where does the raw-parameter render of 1 M Gaussians at 2048^2 (tests/test_gpu_altonly.py case 65) leave the oracle?
HIP raw / HIP activated inputs / oracle raw / oracle activated, forward only, altitude channel."""
import os, sys, time
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from util import run_raw, raw_params_from_scene
from eogs2_amd import _lib
from eogs2_amd.synthetic import make_scene

P, H, W, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20, 2048, 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 65
dev = torch.device("cuda:0")
scene = make_scene(P, H, W, seed=seed, opacity="init", scale_mult=1.0)
raw, alt = raw_params_from_scene(scene, seed=seed)
dL = torch.zeros(5, H, W); dL[3] = torch.randn(H, W, generator=torch.Generator().manual_seed(5)) / (H * W)
sc = dict(scene, dL_dcolor=dL)
to = lambda d: {k: v.to(dev) for k, v in d.items()}
res = {}
res["hip raw"] = run_raw(to(raw), alt.to(dev), to(sc), H, W, False, fused=True)
res["hip act"] = run_raw(to(raw), alt.to(dev), to(sc), H, W, False, fused=False)
real = _lib.get
oabi = oracle.abi(); _lib.get = lambda: oabi
res["ora raw"] = run_raw(raw, alt, sc, H, W, False, fused=True)
res["ora act"] = run_raw(raw, alt, sc, H, W, False, fused=False)
_lib.get = real
ref = res["ora act"]["out_color"][3]
scale = float(ref.abs().max())
for k, v in res.items():
    a = v["out_color"][3].cpu()
    d = (a - ref).abs() / scale
    print(f"{k}: alt max err {float(d.max()):.3e}  pixels beyond 1e-4: {int((d > 1e-4).sum())}  radii != ora act: {int((v['out_radii'].cpu() != res['ora act']['out_radii']).sum())}"
          f"  g_xyz max rel {float((v['g_xyz'].cpu() - res['ora act']['g_xyz']).abs().max() / res['ora act']['g_xyz'].abs().max()):.3e}")
d = (res["hip raw"]["out_color"][3].cpu() - ref).abs() / scale
idx = torch.nonzero(d > 1e-4)
print("bad pixels: rows", int(idx[:, 0].min()) if len(idx) else -1, int(idx[:, 0].max()) if len(idx) else -1, "cols", int(idx[:, 1].min()) if len(idx) else -1, int(idx[:, 1].max()) if len(idx) else -1)
if len(idx):
    h = torch.histc(idx[:, 0].float(), bins=16, min=0, max=H); print("rows hist", h.int().tolist())
    h = torch.histc(idx[:, 1].float(), bins=16, min=0, max=W); print("cols hist", h.int().tolist())
