"""GPU (MI355X, one card): BASELINE.json's configuration 5 in miniature — >= 300 synthetic training iterations at 300 k
Gaussians / 800^2 through per-iteration prune, opacity reset and FusedAdam, then TSDF integration of four views
(tests/config5_child.py; train_pan.py:663-732, tsdf.py:459-498) — on one rank, and as a two-rank data-parallel
rehearsal (the ranks share the card and exchange over gloo) whose replicas must take identical prune decisions and end
in identical states."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "config5_child.py")


def _run(world, args, port=0):
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    procs = []
    for r in range(world):
        env = dict(base, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, CHILD, *args], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = []
    for p in procs:  # exactly the processes started above
        try:
            o, e = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads([ln for ln in o.strip().splitlines() if ln.startswith("{")][-1]))
    return outs


def _check(line, iters):
    assert line["finite"] and line["iters"] == iters
    assert line["loss_last"] < 0.7 * line["loss_first"], line
    assert line["prunes"] >= 1 and line["resets"] == iters // 100 and 0 < line["gaussians_end"] < line["gaussians_start"], line
    assert line["tsdf_finite"] and line["tsdf_touched_frac"] > 0.05 and line["tsdf_surface_frac"] > 0.005, line


def test_config5_single_rank():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    (line,) = _run(1, ["--gaussians", "300000", "--size", "800", "--iters", "300"])
    _check(line, 300)


def test_config5_two_rank_replicas_stay_identical():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a, b = _run(2, ["--gaussians", "300000", "--size", "800", "--iters", "300"], port=29700 + os.getpid() % 200)
    _check(a, 300)
    _check(b, 300)
    # replica-identical prune decisions (every keep mask, in order) and final parameters, bit for bit
    assert a["prune_digest"] == b["prune_digest"] and a["state_digest"] == b["state_digest"], (a, b)
    assert a["gaussians_end"] == b["gaussians_end"] and a["prunes"] == b["prunes"]
