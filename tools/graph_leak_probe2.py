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
train_synthetic.main(base + sys.argv[1:])
