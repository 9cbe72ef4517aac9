"""Where the wave slots of the render kernels are empty (on the GPU box, from the repo root):
    python3 tools/wave_trace.py [--opacity init] [--size 1024] [--envs "EOGS_TILE_SCHED=0|EOGS_TILE_SCHED=1"]
Builds the library with -DEOGS_WAVE_TRACE (every tile's wave stores its start and end on the constant-rate clock, the shader-clock
ticks between them and the SIMD it ran on), runs the headline fwd+bwd in one child process per environment, and prints per
kernel: launch span, resident waves per SIMD over the span (mean, and per tenth of the span), the ramp until 90 % of the slots
are filled, the tail after the first slot went idle for good, the idle time between consecutive waves of one SIMD slot, and the
shader clock the waves saw. Then rebuilds the normal library."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import ctypes, json, sys, numpy as np, torch
sys.path.insert(0, ROOT)
from eogs2_amd import GaussianRasterizer, _lib
from eogs2_amd.synthetic import make_camera, make_scene, settings_for
dev = torch.device('cuda:0')
P, H, W = PP, SZ, SZ
op = OPAC
sc = make_scene(P, H, W, seed=0, opacity=op, device=dev)
sc['viewmatrix'] = make_camera(H, W, seed=0, device=dev)
rast = GaussianRasterizer(settings_for(sc, H, W))
params = {k: sc[k].clone().requires_grad_(True) for k in ('means3D', 'colors', 'opacities', 'scales', 'rotations')}
means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
def step():
    for p in params.values(): p.grad = None
    color, radii, _ = rast(params['means3D'], means2D, params['opacities'], colors_precomp=params['colors'],
                           scales=params['scales'], rotations=params['rotations'])
    torch.autograd.backward([color], [sc['dL_dcolor']])
_lib.get()
lib = ctypes.CDLL(_lib.LIB_PATH)
for _ in range(5): step()
torch.cuda.synchronize()
tiles = ((H + 7) // 8) * ((W + 7) // 8)
res = {}
for d, name in ((0, 'render_fwd_quad'), (1, 'render_bwd_quad')):
    a = np.zeros((tiles, 4), dtype=np.uint64)
    assert lib.eogs_debug_wave_trace(ctypes.c_void_p(a.ctypes.data), d, tiles) == 0
    t0, t1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
    ok = t1 > 0
    t0, t1, clk, hw = t0[ok], t1[ok], a[ok, 2].astype(np.float64), a[ok, 3]
    base = t0.min(); span = int(t1.max() - base)
    s, e = t0 - base, t1 - base
    # SIMD slot identity: XCC (bits 32..35), SE (13..15), SH (12), CU (8..11), SIMD (4..5)
    hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64); xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
    simd = (xcc << 16) | (((hwid >> 13) & 7) << 12) | (((hwid >> 12) & 1) << 11) | (((hwid >> 8) & 15) << 4) | ((hwid >> 4) & 3)
    nsimd = len(np.unique(simd))
    wave_id = hwid & 15
    # resident waves over time (100 MHz ticks)
    ev = np.zeros(span + 2, dtype=np.int64); np.add.at(ev, s, 1); np.add.at(ev, e, -1)
    resident = np.cumsum(ev)[:span]
    per_simd = resident / float(nsimd)
    tenths = [float(per_simd[int(span * i / 10):max(int(span * (i + 1) / 10), int(span * i / 10) + 1)].mean()) for i in range(10)]
    peak = int(resident.max())
    ramp = int(np.argmax(resident >= 0.9 * peak))
    # idle time between consecutive waves of one (SIMD, wave slot)
    slot = simd * 16 + wave_id
    order = np.lexsort((s, slot)); ss, ee, sl = s[order], e[order], slot[order]
    same = sl[1:] == sl[:-1]
    gaps = (ss[1:] - ee[:-1])[same]
    dur = (e - s).astype(np.float64)
    res[name] = dict(waves=int(ok.sum()), simds_seen=nsimd, span_us=span / 100.0, mean_wave_us=float(dur.mean()) / 100.0,
                     p5_wave_us=float(np.percentile(dur, 5)) / 100.0, p95_wave_us=float(np.percentile(dur, 95)) / 100.0,
                     max_wave_us=float(dur.max()) / 100.0,
                     sum_wave_us_per_simd=float(dur.sum()) / 100.0 / nsimd, mean_resident_per_simd=float(per_simd.mean()),
                     resident_per_simd_by_tenth=[round(x, 2) for x in tenths], peak_resident=peak,
                     ramp_to_90pct_us=ramp / 100.0,
                     last_start_us=float(s.max()) / 100.0, first_end_us=float(e.min()) / 100.0,
                     slot_gap_us=dict(n=int(gaps.size), mean=float(gaps.mean()) / 100.0 if gaps.size else 0.0,
                                      p50=float(np.percentile(gaps, 50)) / 100.0 if gaps.size else 0.0,
                                      p95=float(np.percentile(gaps, 95)) / 100.0 if gaps.size else 0.0),
                     distinct_slots=int(len(np.unique(slot))),
                     shader_clock_GHz=float((clk / np.maximum(dur, 1.0)).mean()) * 0.1,
                     xcc_last_end_us=[float(e[xcc == x].max()) / 100.0 if (xcc == x).any() else 0.0 for x in range(8)],
                     xcc_wave_us_per_simd=[float(dur[xcc == x].sum()) / 100.0 / max(len(np.unique(simd[xcc == x])), 1) for x in range(8)],
                     waves_per_simd_hist=np.bincount(np.bincount(np.unique(simd, return_inverse=True)[1])).tolist())
print('WAVE_TRACE ' + json.dumps(res))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--opacity", default="init")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--gaussians", type=int, default=1 << 20)
    ap.add_argument("--envs", default="EOGS_TILE_SCHED=0|EOGS_TILE_SCHED=1", help="'|'-separated sets of VAR=value (space-separated inside a set)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "wave_trace.json"))
    a = ap.parse_args()
    from eogs2_amd import build
    try:
        op = repr(float(a.opacity))
    except ValueError:
        op = repr(a.opacity)
    build.build(force=True, extra=["-DEOGS_WAVE_TRACE"], verbose=False)
    out = {}
    try:
        code = CHILD.replace("ROOT", repr(ROOT)).replace("PP", str(a.gaussians)).replace("SZ", str(a.size)).replace("OPAC", op)
        for mode in a.envs.split("|"):
            env = dict(os.environ, **dict(kv.split("=", 1) for kv in mode.split()))
            r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                print(f"mode {mode} failed:", r.stderr[-2000:])
                continue
            line = [l for l in r.stdout.splitlines() if l.startswith("WAVE_TRACE ")][-1]
            out[mode] = json.loads(line[len("WAVE_TRACE "):])
            for k, v in out[mode].items():
                print(f"[{mode}] {k}: " + json.dumps(v))
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)
    finally:
        build.build(force=True, verbose=False)


if __name__ == "__main__":
    main()
