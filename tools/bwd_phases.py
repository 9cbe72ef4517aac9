"""Where a wave of render_bwd_quad_kernel spends its residency (on the GPU box, from the repo root):
    python3 tools/bwd_phases.py [--opacity init]
Builds the library with -DEOGS_BWD_PHASES (s_memtime samples at the phase boundaries, summed over all waves), runs the
headline fwd+bwd a few times and prints each phase's share of the summed wave time; then rebuilds the normal library.
s_memtime counts a constant 100 MHz clock; only the shares are meaningful."""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--opacity", default="init")
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    from eogs2_amd import build
    build.build(force=True, extra=["-DEOGS_BWD_PHASES"], verbose=False)
    try:
        code = f"""
import ctypes, sys, torch
sys.path.insert(0, {ROOT!r})
from eogs2_amd import GaussianRasterizer, _lib
from eogs2_amd.synthetic import make_camera, make_scene, settings_for
dev = torch.device('cuda:0')
P, H, W = 1 << 20, 1024, 1024
op = {a.opacity!r}
try: op = float(op)
except ValueError: pass
sc = make_scene(P, H, W, seed=0, opacity=op, device=dev)
sc['viewmatrix'] = make_camera(H, W, seed=0, device=dev)
rast = GaussianRasterizer(settings_for(sc, H, W))
params = {{k: sc[k].clone().requires_grad_(True) for k in ('means3D', 'colors', 'opacities', 'scales', 'rotations')}}
means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
def step():
    for p in params.values(): p.grad = None
    color, radii, _ = rast(params['means3D'], means2D, params['opacities'], colors_precomp=params['colors'],
                           scales=params['scales'], rotations=params['rotations'])
    torch.autograd.backward([color], [sc['dL_dcolor']])
_lib.get()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
for _ in range(3): step()
torch.cuda.synchronize(); assert lib.eogs_debug_bwd_phases(buf, 1) == 0
for _ in range({a.steps}): step()
torch.cuda.synchronize(); assert lib.eogs_debug_bwd_phases(buf, 0) == 0
v = list(buf); tot = sum(v[:5])
names = ['chunk setup (gather, quad masks, sub-lists)', 'trips', 'transposition + staging', 'owner pull', 'record formation + stores']
for n, x in zip(names, v[:5]): print(f'{{n:46s}} {{100.0 * x / tot:5.1f}} %')
print('waves', v[7], ' mean residency', tot / max(v[7], 1) * 10e-3, 'us (100 MHz counter)')
"""
        subprocess.check_call([sys.executable, "-c", code])
    finally:
        build.build(force=True, verbose=False)


if __name__ == "__main__":
    main()
