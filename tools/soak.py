"""Round 4: soak run of the full-chain example (examples/train_synthetic.py) — thousands of iterations with the prune firing
(every prune re-materialises the parameters and, with --graph, records the iteration's HIP graph again), eager and as a replayed
graph with its renders as parallel branches. Reports the loss, the surviving Gaussians and the device memory after each leg:
growth from leg to leg would be a leak (scratch buffers kept per stream, private graph pools that are not released).

    python tools/soak.py [--gaussians 200000] [--size 512] [--iters 3000]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))


def mem(tag):
    torch.cuda.synchronize()
    import gc

    gc.collect()
    torch.cuda.empty_cache()
    f, t = torch.cuda.mem_get_info()  # the device's own figure: what the HIP runtime holds besides the allocator's segments shows here only
    print(f"  [{tag}] allocated {torch.cuda.memory_allocated() / 2**20:8.1f} MiB   reserved {torch.cuda.memory_reserved() / 2**20:8.1f} MiB   "
          f"peak allocated {torch.cuda.max_memory_allocated() / 2**20:8.1f} MiB   device in use {(t - f) / 2**20:8.1f} MiB", flush=True)
    torch.cuda.reset_peak_memory_stats()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=200_000)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--iters", type=int, default=3000)
    a = ap.parse_args()
    import train_synthetic

    base = ["--gaussians", str(a.gaussians), "--size", str(a.size), "--iters", str(a.iters), "--quiet", "--sun-altitude-only",
            "--random-camera"]
    mem("start")
    for name, extra in (("eager", []), ("graph", ["--graph"]), ("graph, parallel renders", ["--graph", "--parallel-renders"]),
                        ("graph, parallel renders (again)", ["--graph", "--parallel-renders"])):
        t0 = time.perf_counter()
        first, last, n = train_synthetic.main(base + extra)
        dt = time.perf_counter() - t0
        print(f"{name}: loss {first:.5f} -> {last:.5f}, {n} of {a.gaussians} Gaussians left, {dt:.1f} s "
              f"({train_synthetic.main.last_ms_per_iter:.3f} ms/iter over the last half)", flush=True)
        mem(name)


if __name__ == "__main__":
    main()
