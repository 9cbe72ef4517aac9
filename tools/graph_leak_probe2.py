"""Round 4: device memory around every re-recording of the example's iteration (see graph_leak_probe.py). Wraps GraphedStep.__init__
and logs hipMemGetInfo + the allocator's own figures before the old step is dropped, after it is dropped (gc + empty_cache) and
after the new recording.   python tools/graph_leak_probe2.py [extra example arguments]"""
import gc
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import eogs2_amd.graph as G  # noqa: E402
import train_synthetic  # noqa: E402


def snap(tag):
    torch.cuda.synchronize()
    f, t = torch.cuda.mem_get_info()
    st = torch.cuda.memory_stats()
    print(f"    {tag:26s} device in use {(t - f) / 2**20:8.0f} MiB | torch reserved {torch.cuda.memory_reserved() / 2**20:7.0f} allocated "
          f"{torch.cuda.memory_allocated() / 2**20:7.0f} | segments {st.get('segment.all.current', 0)}", flush=True)


if os.environ.get("PROBE_OLD_INIT"):  # GraphedStep as it was before it kept a pool: no pool argument, no anchor graph
    def _old_init(self, fn, warmup=2, idempotent=True):
        self.fn, self.idempotent = fn, idempotent
        self.replays = self.recaptures = 0
        self._mirror = torch.empty((64 * (self.MAX_MIRRORED + 1),), dtype=torch.uint8, pin_memory=True)
        self._pool = None
        self._warm_up(warmup)
        self._capture()

    G.GraphedStep.__init__ = _old_init
orig = G.GraphedStep.__init__
count = [0]


def patched(self, *a, **k):
    count[0] += 1
    print(f"  recording {count[0]}", flush=True)
    snap("before (old step dropped)")
    gc.collect()
    torch.cuda.empty_cache()
    snap("after gc + empty_cache")
    orig(self, *a, **k)
    snap("after the new recording")


G.GraphedStep.__init__ = patched
orig_again = G.GraphedStep.record_again


def patched_again(self, *a, **k):
    count[0] += 1
    print(f"  recording {count[0]} (record_again)", flush=True)
    snap("before")
    orig_again(self, *a, **k)
    snap("after the new recording")


G.GraphedStep.record_again = patched_again
base = ["--gaussians", "200000", "--size", "512", "--iters", "158", "--sun-altitude-only", "--random-camera", "--graph",
        "--prune-every", "1", "--quiet"]
if os.environ.get("PROBE_NEW_STEP"):  # the behaviour before record_again: a new GraphedStep (a new pool) per recording
    G.GraphedStep.record_again = lambda self, warmup=1: (_ for _ in ()).throw(RuntimeError("unused"))
    src = open(train_synthetic.__file__).read().replace("elif stale:\n                step.record_again()", "elif stale:\n                step = None; step = GraphedStep(fwd_bwd, warmup=1)")
    ns = {"__name__": "train_synthetic_new_step", "__file__": train_synthetic.__file__}
    exec(compile(src, train_synthetic.__file__, "exec"), ns)
    train_synthetic = type("M", (), {"main": staticmethod(ns["main"])})
if os.environ.get("PROBE_NO_POOL"):  # every capture in a pool of its own, as before the step kept one (the anchor graph too)
    _graph = torch.cuda.graph

    class _NoPool(_graph):
        def __init__(self, g, pool=None, **k):
            super().__init__(g, **k)

    torch.cuda.graph = _NoPool
drop = [a[5:] for a in sys.argv[1:] if a.startswith("drop=")]
base = [b for b in base if b not in drop]
train_synthetic.main(base + [a for a in sys.argv[1:] if not a.startswith("drop=")])
