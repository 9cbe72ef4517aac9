"""Round 4: is the full-chain example run-to-run reproducible, and is the deferred prune the prune? Prints (first loss, last loss,
survivors) of repeated runs of examples/train_synthetic.py; optional extra arguments are passed on."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import train_synthetic  # noqa: E402

base = ["--gaussians", "30000", "--size", "192", "--iters", "400", "--quiet"] + sys.argv[1:]
runs = ([], [], ["--defer-prune", "3"], ["--defer-prune", "3"], ["--no-prune"], ["--no-prune"])
if os.environ.get("PROBE_PLAIN_ONLY"):
    runs = ([], [], [])
for extra in runs:
    print(extra, train_synthetic.main(base + extra), flush=True)
