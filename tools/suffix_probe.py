"""CPU: how far the HIP path's formulation of dL/dalpha (front to back, the sum behind a Gaussian taken as rendered total
minus running prefix) moves the ORACLE's gradients when the oracle is switched to it (eogs_oracle_suffix_by_subtraction) —
algebraically the same value as the reference's back-to-front recursion, a different fp32 error. DESIGN.md 5.
    python tools/suffix_probe.py 1258 216 1302"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from parity_cases import oracle_run, sweep_case  # noqa: E402

lib = oracle.abi().cdll
for seed in map(int, sys.argv[1:] or ["1258"]):
    case, label = sweep_case(seed)
    base = oracle_run(case)
    alts = {}
    for mode in (1, 2, 3):  # 1: total from the rendered image (the HIP path); 2: total = the running sum's own end value (two walks)
        lib.eogs_oracle_suffix_by_subtraction(mode)
        try:
            alts[mode] = oracle_run(case)
        finally:
            lib.eogs_oracle_suffix_by_subtraction(0)
    alt = alts[1]
    print(seed, label, case["means3D"].shape[0], "Gaussians", f'{case["H"]}x{case["W"]}')
    for k in ("g_rotations", "g_scales", "g_means3D", "g_opacities", "g_colors", "g_viewmatrix"):
        sc = np.abs(base[k]).max()
        d = np.abs(alt[k].astype(np.float64) - base[k]) / max(sc, 1e-30)
        d2 = np.abs(alts[2][k].astype(np.float64) - base[k]) / max(sc, 1e-30)
        d3 = np.abs(alts[3][k].astype(np.float64) - base[k]) / max(sc, 1e-30)
        print(f"   {k:12s} max |front-to-back - back-to-front| = {d.max():.2e} of the tensor scale ({int((d > 1e-4).sum())} elements beyond 1e-4);"
              f" with a self-consistent total {d2.max():.2e}; back to front with the projected recursion {d3.max():.2e}")
