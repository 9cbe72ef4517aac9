"""Tuning aid (GPU box): bench.photometric_loss_bench alone (fused L1 + SSIM loss, 3 x 1024^2): kernel ms and wall ms."""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from eogs2_amd import _lib
r = bench.photometric_loss_bench(_lib.get(), torch.device("cuda:0"), 1024, 1024)
print(json.dumps({"fused_ms": round(r["fused_ms"], 4), "kernels_ms": {k: round(v, 4) for k, v in r["kernels_ms"].items()}, "value_fused": r["value_fused"], "value_torch_ops": r["value_torch_ops"]}))
