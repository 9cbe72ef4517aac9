"""GPU box: are a sweep seed's outputs and gradients the same BITS under the library's run-time switches?
    python tools/seed_bits.py 1259
One child process per environment (the switches are read once per process); exits non-zero when a child fails or prints no
result — an error is never taken as a baseline."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import torch
    from parity_cases import sweep_case
    from util import run_case

    from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer

    case, name = sweep_case(int(sys.argv[2]))
    got = run_case(case, torch.device("cuda:0"), GaussianRasterizer, GaussianRasterizationSettings)
    out = {}
    for k, v in got.items():
        if hasattr(v, "detach"):
            a = v.detach().cpu().numpy()
            a = a + 0.0 if a.dtype.kind == "f" else a  # -0.0 -> +0.0
            out[k] = hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
    assert out, "the case produced no tensors"
    print("RESULT " + json.dumps(out))
    sys.exit(0)

seed = sys.argv[1] if len(sys.argv) > 1 else "1259"
ALL_OFF = {"EOGS_GB_WIDE": "0", "EOGS_FWD_MASKS": "0", "EOGS_NOFLAG": "0", "EOGS_PLAIN_TRIPS": "0"}
base, rc = None, 0
for env in ({}, {"EOGS_GB_WIDE": "0"}, {"EOGS_FWD_MASKS": "0"}, {"EOGS_NOFLAG": "0"}, {"EOGS_PLAIN_TRIPS": "0"}, ALL_OFF):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", seed], env=dict(os.environ, **env), capture_output=True, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    if r.returncode != 0 or not lines:
        print(env, "CHILD FAILED (exit %d): %s" % (r.returncode, (r.stderr or r.stdout)[-400:]))
        rc = 1
        continue
    d = json.loads(lines[-1][7:])
    if base is None:
        base = d
    print(env, "SAME" if d == base else {k: (base.get(k), v) for k, v in d.items() if base.get(k) != v})
    rc = rc or int(d != base)
sys.exit(rc)
