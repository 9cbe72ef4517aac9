import hashlib, os, subprocess, sys, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch, numpy as np
    from parity_cases import sweep_case, run_case
    from eogs2_amd import GaussianRasterizationSettings, GaussianRasterizer
    case, name = sweep_case(int(sys.argv[2]))
    got = run_case(case, torch.device('cuda:0'), GaussianRasterizer, GaussianRasterizationSettings)
    out = {}
    for k, v in got.items():
        if hasattr(v, 'detach'):
            a = v.detach().cpu().numpy()
            a = a + 0.0 if a.dtype.kind == 'f' else a  # -0.0 -> +0.0
            out[k] = hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
    print(json.dumps(out))
    sys.exit(0)
seed = sys.argv[1] if len(sys.argv) > 1 else '1259'
base = None
for env in ({}, {'EOGS_GB_WIDE': '0'}, {'EOGS_FWD_MASKS': '0'}, {'EOGS_NOFLAG': '0'}, {'EOGS_PLAIN_TRIPS': '0'}, {'EOGS_GB_WIDE': '0', 'EOGS_FWD_MASKS': '0', 'EOGS_NOFLAG': '0', 'EOGS_PLAIN_TRIPS': '0'}):
    r = subprocess.run([sys.executable, __file__, 'child', seed], env=dict(os.environ, **env), capture_output=True, text=True)
    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
    d = json.loads(line) if line.startswith('{') else {'err': line}
    if base is None: base = d
    print(env, 'SAME' if d == base else {k: (base.get(k), v) for k, v in d.items() if base.get(k) != v})
