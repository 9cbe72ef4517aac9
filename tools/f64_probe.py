"""CPU: how far the valid fp32 evaluations of the reference's algorithm (the oracle, its FMA-contracted build, its fp32-accumulating
mode) sit from the arbiter build (oracle/librast_oracle_f64.so: same decisions, double arithmetic) on sweep seeds.
    python tools/f64_probe.py 1259 4275 216"""
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from parity_cases import fp32_variants, oracle_run, quantity_scale, sweep_case  # noqa: E402

for seed in map(int, sys.argv[1:] or ["1259"]):
    case, label = sweep_case(seed)
    base = oracle_run(case)
    f64 = oracle_run(case, backend=oracle.abi_f64)
    var = fp32_variants(case)
    print(seed, label, case["means3D"].shape[0], "Gaussians", f'{case["H"]}x{case["W"]}')
    for k in ("g_rotations", "g_scales", "g_means3D", "g_opacities", "g_colors", "g_means2D", "g_viewmatrix"):
        sc = quantity_scale(f64[k]).numpy()
        d0 = np.abs(base[k].astype(np.float64) - f64[k]) / sc
        dv = {n: np.abs(v[k].astype(np.float64) - f64[k]) / sc for n, v in var.items()}
        print(f"   {k:12s} |oracle - f64| max {d0.max():.2e}   " + "   ".join(f"|{n} - f64| {d.max():.2e}" for n, d in dv.items()))
