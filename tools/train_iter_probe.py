"""The fused synthetic training iteration of bench.py alone (3 renders fwd+bwd + loss + FusedAdam, 1 M Gaussians), for
profiling: `rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/train_iter_probe.py` and then
`python3 tools/train_iter_probe.py --gaps OUT` prints, per iteration, the device-busy time against the wall time between
the first and the last kernel (what the host leaves idle)."""
import glob, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--gaps":
    import csv
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # iterations end with the Adam kernel
    ends = [i for i, r in enumerate(rows) if "adam" in r[2].lower()]
    for a, b in list(zip(ends[:-1], ends[1:]))[-8:]:
        seg = rows[a + 1:b + 1]
        busy = sum(e - s for s, e, _ in seg)
        wall = seg[-1][1] - seg[0][0]
        gaps = sorted(((seg[i + 1][0] - seg[i][1], seg[i][2][:40], seg[i + 1][2][:40]) for i in range(len(seg) - 1)), reverse=True)[:6]
        print(f"iteration: wall {wall/1e3:8.1f} us  busy {busy/1e3:8.1f} us  idle {100*(1-busy/wall):4.1f} %  kernels {len(seg)}")
        for g, k0, k1 in gaps:
            print(f"      gap {g/1e3:7.1f} us  after {k0}  before {k1}")
    # device time per kernel over the last iteration
    a, b = ends[-2], ends[-1]
    tot = {}
    for s_, e_, k in rows[a + 1:b + 1]:
        k = k.split("(")[0][-48:]
        tot.setdefault(k, [0, 0])
        tot[k][0] += e_ - s_
        tot[k][1] += 1
    for k, (ns, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
        print(f"   {ns/1e3:8.1f} us  x{c:<3d} {k}")
    sys.exit(0)

import torch
import bench
from eogs2_amd.synthetic import make_scene
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
P, H, W = 1 << 20, 1024, 1024
sc = make_scene(P, H, W, seed=0, opacity="init", device=dev)
print(bench.train_iteration(sc, P, H, W, dev, fused=True, iters=20))
