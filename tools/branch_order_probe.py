"""Tuning aid (GPU box): the bench's 3-render iteration as a graph with parallel branches, for a given order of queueing the renders
(EOGS_BRANCH_ORDER=1,0,2: the sun camera first). usage: EOGS_BRANCH_ORDER=0,1,2 python tools/branch_order_probe.py"""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from eogs2_amd.synthetic import make_scene
dev = torch.device("cuda:0")
P, H, W = 1 << 20, 1024, 1024
sc = make_scene(P, H, W, seed=0, opacity="init", device=dev)
for alt in (False, True):
    r = bench.train_iteration(sc, P, H, W, dev, fused=True, graphed=True, parallel=True, iters=20, sun_altitude_only=alt)
    print(os.environ.get("EOGS_BRANCH_ORDER", "0,1,2"), "alt_only" if alt else "full", round(r["ms_per_iter"], 4))
