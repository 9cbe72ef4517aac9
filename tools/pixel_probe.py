"""GPU: where the forward image of a sweep seed differs from the oracle's — pixel, channel, both values — and which Gaussians the
oracle blends there with alpha / transmittance close to a threshold.   python tools/pixel_probe.py 9241 [...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from parity_cases import oracle_run, quantity_scale, sweep_case  # noqa: E402
from util import run_case  # noqa: E402

dev = torch.device("cuda:0")
for seed in map(int, sys.argv[1:] or ["9241"]):
    case, label = sweep_case(seed)
    got = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
    ref = oracle_run(case)
    H, W = case["H"], case["W"]
    print(seed, label, f"{H}x{W}", case["means3D"].shape[0], "Gaussians", {k: case[k] for k in case if np.isscalar(case[k]) or isinstance(case[k], (bool, int, float))})
    h = np.asarray(got["out_color"].cpu(), dtype=np.float64)
    r = np.asarray(ref["out_color"], dtype=np.float64)
    sc = quantity_scale(torch.as_tensor(r)).numpy()
    e = np.abs(h - r) / sc
    bad = np.argwhere(e > 1e-4)
    print("   radii equal:", bool(np.array_equal(np.asarray(got["out_radii"].cpu()), ref["out_radii"])), " elements beyond 1e-4:", len(bad))
    for c, y, x in bad[:12]:
        print(f"   channel {c} pixel ({x},{y})  tile8 ({x // 8},{y // 8})  HIP {h[c, y, x]:.7g}  oracle {r[c, y, x]:.7g}  err {e[c, y, x]:.3e}")
    # the Gaussians the reference would consider at the first bad pixel: alpha, and the running transmittance
    if len(bad):
        _, y, x = bad[0]
        import math
        m3, op = case["means3D"], case["opacities"].reshape(-1)
        print("   (per-pair detail needs the oracle's trace: see tests/parity_cases.py Attribution)")
