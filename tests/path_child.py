"""Helper process of tests/test_gpu_paths.py: with the tuning switches of its environment (read once per process), runs
every case file of a directory through the HIP library, compares it with the oracle results stored in the file
(tests/parity_cases.py compare: 1e-4 + attributed threshold flips), and records which kernels the library chose
(eogs_rast_path_info). Writes one JSON: {case: {"ok", "error", "path": [list block px, fwd kernel, bwd kernel], "flips"}}."""
import glob
import json
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from parity_cases import compare  # noqa: E402
from util import run_case  # noqa: E402

from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib  # noqa: E402


def main(case_dir, out_json):
    abi = _lib.get()
    assert abi.backend == "hip-gfx950"
    dev = torch.device("cuda:0")
    res = {}
    for f in sorted(glob.glob(os.path.join(case_dir, "*.npz"))):
        name = os.path.basename(f)[:-4]
        z = np.load(f)
        case = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
        ref = {k[4:]: z[k] for k in z.files if k.startswith("ref_")}
        label = str(z["label"])
        try:
            got = run_case(case, dev, GaussianRasterizer, GaussianRasterizationSettings)
            P = case["means3D"].shape[0]
            path = list(abi.path_info(P, got.get("_num_rendered", 0))) if P else [0, -1, -1]
            flips = compare(got, ref, label, case, cache=os.path.join(case_dir, "cache", name))
            res[name] = {"ok": True, "path": path, "flips": flips, "listed": bool(int(got.get("_num_rendered", 0)) & 0x7FFFFFFF)}
        except Exception as e:  # noqa: BLE001 - reported to the parent test
            res[name] = {"ok": False, "error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()[-1500:]}
    json.dump(res, open(out_json, "w"))


if __name__ == "__main__":
    main(*sys.argv[1:])
