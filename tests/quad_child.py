"""Helper process of tests/test_gpu_quad.py: renders one seeded scene forward + backward through the HIP library with the
tuning switches of its environment (they are read once per process) and writes outputs and gradients to an .npz."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from util import run_case  # noqa: E402

from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from eogs2_amd.synthetic import make_scene  # noqa: E402


def main(out_path, P, H, W, opacity, invdepth):
    try:
        opacity = float(opacity)
    except ValueError:
        pass
    sc = make_scene(int(P), int(H), int(W), seed=3, opacity=opacity)
    case = {k: v.numpy() for k, v in sc.items()}
    case.update(H=int(H), W=int(W), antialiasing=False)
    if int(invdepth):
        g = torch.Generator().manual_seed(9)
        case["dL_dinvdepth"] = (torch.randn((1, int(H), int(W)), generator=g) / (int(H) * int(W))).numpy()
    got = run_case(case, torch.device("cuda:0"), GaussianRasterizer, GaussianRasterizationSettings)
    np.savez(out_path, **{k: v.detach().cpu().numpy() for k, v in got.items() if not k.startswith('_')})


if __name__ == "__main__":
    main(*sys.argv[1:])
