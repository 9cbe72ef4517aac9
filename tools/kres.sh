#!/bin/bash
# Register / scratch / LDS use of every kernel of one source file, as the compiler reports it (no GPU needed).
#   tools/kres.sh eogs2_amd/csrc/render.hip [extra -D flags]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
f=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I "$ROOT/include" "$@" -c "$f" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys, subprocess
cur = None
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        if cur: print(cur)
        name = t.split(":", 1)[1].strip()
        try: name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        except Exception: pass
        cur = name[-60:].ljust(62)
    elif any(t.startswith(k) for k in ("VGPRs:", "ScratchSize", "Occupancy", "LDS Size", "TotalSGPRs:")):
        cur += " " + t.replace(" [bytes/lane]", "").replace(" [waves/SIMD]", "").replace(" [bytes/block]", "").replace("ScratchSize", "scratch").replace("Occupancy", "occ").replace("LDS Size", "lds").replace("TotalSGPRs", "sgpr")
if cur: print(cur)
'
