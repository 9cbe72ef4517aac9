"""Where a workgroup of block_lists_kernel spends its time (on the GPU box, from the repo root):
    python3 tools/bl_phases.py [--opacity init] [--size 1024]
Builds the library with -DEOGS_BL_PHASES (thread 0's clock at the phase boundaries, summed over the workgroups), runs the
fwd+bwd a few times and prints each phase's share and mean duration; then rebuilds the normal library."""
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
    ap.add_argument("--size", type=int, default=1024)
    a = ap.parse_args()
    from eogs2_amd import build
    build.build(force=True, extra=["-DEOGS_BL_PHASES"], verbose=False)
    try:
        code = f"""
import ctypes, sys, torch
sys.path.insert(0, {ROOT!r})
from eogs2_amd import GaussianRasterizer, _lib
from eogs2_amd.synthetic import make_camera, make_scene, settings_for
dev = torch.device('cuda:0')
P, H, W = 1 << 20, {a.size}, {a.size}
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
torch.cuda.synchronize(); assert lib.eogs_debug_bl_phases(buf, 1) == 0
for _ in range({a.steps}): step()
torch.cuda.synchronize(); assert lib.eogs_debug_bl_phases(buf, 0) == 0
v = list(buf); tot = sum(v[:6])
names = ['prologue (fits, start of the block: sums over the blocks before)', 'entry loads issued', 'radix passes', 'per-tile counts, ranges, descriptors', 'live flags cleared', 'split into the tile lists']
for n, x in zip(names, v[:6]): print(f'{{n:66s}} {{100.0 * x / tot:5.1f}} %  {{x / max(v[7], 1) * 10e-3:7.2f}} us per workgroup')
print('workgroups', v[7], ' mean lifetime of thread 0', tot / max(v[7], 1) * 10e-3, 'us (100 MHz counter)')
"""
        subprocess.check_call([sys.executable, "-c", code])
    finally:
        build.build(force=True, verbose=False)


if __name__ == "__main__":
    main()
