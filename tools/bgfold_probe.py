"""GPU: one sweep seed through the library as it stands — how many elements of each gradient lie beyond 1e-4 of the oracle, and how
far the HIP values and the oracle's sit from the float64 arbiter on those elements.   python tools/bgfold_probe.py 7232 [...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from parity_cases import oracle_run, quantity_scale, sweep_case  # noqa: E402
from util import run_case  # noqa: E402

dev = torch.device("cuda:0")
for seed in map(int, sys.argv[1:] or ["7232"]):
    case, label = sweep_case(seed)
    got = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    base = oracle_run(case)
    f64 = oracle_run(case, backend=oracle.abi_f64)
    print(seed, label, "bg", np.asarray(case["bg"]).tolist())
    for k in ("g_rotations", "g_scales", "g_means3D", "g_opacities", "g_colors"):
        sc = quantity_scale(torch.as_tensor(np.asarray(base[k], dtype=np.float64))).numpy()
        h = np.asarray(got[k].cpu() if hasattr(got[k], "cpu") else got[k], dtype=np.float64)
        e = np.abs(h - base[k]) / sc
        bad = e > 1e-4
        dh = np.abs(h - f64[k]) / sc
        do = np.abs(np.asarray(base[k], dtype=np.float64) - f64[k]) / sc
        print(f"   {k:12s} beyond 1e-4: {int(bad.sum()):4d} of {bad.size}   on those: |HIP-f64| median {np.median(dh[bad]) if bad.any() else 0:.2e} max {dh[bad].max() if bad.any() else 0:.2e}"
              f"   |oracle-f64| median {np.median(do[bad]) if bad.any() else 0:.2e} max {do[bad].max() if bad.any() else 0:.2e}   HIP closer on {int((dh[bad] <= do[bad]).sum())}"
              f"   overall rms |HIP-f64| {np.sqrt((dh**2).mean()):.3e} |oracle-f64| {np.sqrt((do**2).mean()):.3e}")
