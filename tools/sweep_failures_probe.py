import os, sys, torch
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests"))
from parity_cases import sweep_case, oracle_run, compare
from util import run_case
from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
dev=torch.device("cuda:0")
bad=0
for sd in [1002,1012,1050,1090,1126,1340,1342,1379,1391,2115,2219,2224,2311,2362]:
    case,label=sweep_case(sd)
    got=run_case(case,dev,GaussianRasterizer,GaussianRasterizationSettings)
    path=_lib.get().path_info(case["means3D"].shape[0], got["_num_rendered"])
    try:
        compare(got, oracle_run(case), label, case); print(sd, label, path, "ok", flush=True)
    except AssertionError as e:
        bad+=1; print(sd, label, path, "FAIL", str(e)[:200], flush=True)
print("failed", bad)
