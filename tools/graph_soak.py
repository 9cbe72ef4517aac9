import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from test_gpu_graph import _Step
from eogs2_amd.graph import GraphedStep
dev = torch.device("cuda:0")
st = _Step(50000, 256, 256, dev, seed=1)
g = GraphedStep(st, warmup=1)
base = None
for i in range(30):
    # alternate small / large footprints: every switch to "large" outgrows the graph recorded for "small" only the first time
    st.load(100 + i, scale_mult=(0.4 if i % 2 == 0 else 2.5))
    out = g()
    ref = st()
    assert all(torch.equal(a, b) for a, b in zip(out[:3], ref[:3])), i
    g._capture()  # force a re-record each round: the pool of the old graph must be released
    torch.cuda.synchronize()
    m = torch.cuda.memory_allocated() >> 20
    r = torch.cuda.memory_reserved() >> 20
    if i in (3, 29): print("round", i, "allocated MiB", m, "reserved MiB", r, "recaptures", g.recaptures, flush=True)
    if i == 3: base = (m, r)
assert m <= base[0] * 1.2 + 64 and r <= base[1] * 1.5 + 256, (base, m, r)
print("soak ok")
