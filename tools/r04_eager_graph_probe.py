import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import train_synthetic
for cfg in ([], ["--sun-altitude-only", "--random-camera"], ["--require-radii"]):
    base = ["--gaussians", "30000", "--size", "192", "--iters", "120", "--quiet"] + cfg
    e = train_synthetic.main(base); g = train_synthetic.main(base + ["--graph"]); p = train_synthetic.main(base + ["--graph", "--parallel-renders"]) if "--random-camera" in cfg else None
    print(cfg, "\n  eager", e, "\n  graph", g, "\n  par  ", p, flush=True)
