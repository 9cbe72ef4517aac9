"""Conditioning probe (GPU box): for sweep seeds whose g_rotations differ between the HIP path and the C oracle, compare
both with a float64 evaluation of the same chain (oracle/torch_dense.py run on float64 inputs).
    python tools/cond_probe.py 1185 1329 1359"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from oracle.torch_dense import render_dense  # noqa: E402
from util import run_case  # noqa: E402

from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib  # noqa: E402
from eogs2_amd.synthetic import make_scene  # noqa: E402


def scene_of(seed):  # tests/test_gpu_parity.py::test_randomised_sweep_against_oracle
    g = torch.Generator().manual_seed(seed)
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    P, H, W = r(1, 3000), r(9, 200), r(9, 260)
    opacity = ["init", "trained", 0.3, 0.02, 0.95][r(0, 4)]
    scale_mult = [0.5, 1.0, 2.5, 6.0, 15.0][r(0, 4)]
    aniso = [0.0, 0.3, 0.7, 1.2][r(0, 3)]
    aa, dgrad = bool(r(0, 1)), bool(r(0, 1))
    sc = make_scene(P, H, W, seed=seed, opacity=opacity, scale_mult=scale_mult, anisotropy=aniso)
    if r(0, 1):
        q = torch.randn(P, 4, generator=g)
        sc["rotations"] = (q / q.norm(dim=1, keepdim=True)).contiguous()
    case = {k: v.numpy() for k, v in sc.items()}
    case.update(H=H, W=W, antialiasing=aa)
    if dgrad:
        case["dL_dinvdepth"] = (torch.randn(1, H, W, generator=g) / (H * W) * 100).numpy()
    return sc, case, dict(P=P, H=H, W=W, opacity=opacity, scale_mult=scale_mult, aniso=aniso, aa=aa, dgrad=dgrad)


for seed in map(int, sys.argv[1:]):
    sc, case, info = scene_of(seed)
    hip = run_case(case, torch.device("cuda:0"), GaussianRasterizer, GaussianRasterizationSettings)
    real = _lib.get
    _lib.get = oracle.abi
    try:
        ref = run_case(case, torch.device("cpu"), GaussianRasterizer, GaussianRasterizationSettings)
    finally:
        _lib.get = real
    d = lambda k: torch.from_numpy(case[k]).double()
    leaves = {k: d(k).requires_grad_(True) for k in ("means3D", "opacities", "colors", "scales", "rotations")}
    c, _, inv = render_dense(leaves["means3D"], leaves["opacities"], leaves["colors"], d("bg"), d("viewmatrix"), info["H"], info["W"],
                             scales=leaves["scales"], rotations=leaves["rotations"], antialiasing=info["aa"], block=32)
    loss = (c * d("dL_dcolor")).sum()
    if "dL_dinvdepth" in case:
        loss = loss + (inv * d("dL_dinvdepth")).sum()
    loss.backward()
    print(seed, info)
    for name, key in (("g_rotations", "rotations"), ("g_scales", "scales"), ("g_means3D", "means3D")):
        f64 = leaves[key].grad
        scale = float(f64.abs().max())
        e_hip = float((hip[name].double().cpu() - f64).abs().max()) / scale
        e_ora = float((ref[name].double() - f64).abs().max()) / scale
        e_ho = float((hip[name].double().cpu() - ref[name].double()).abs().max()) / scale
        print(f"   {name:12s} |HIP-f64| {e_hip:.2e}   |oracle-f64| {e_ora:.2e}   |HIP-oracle| {e_ho:.2e}   (of max |f64| {scale:.2e})")
