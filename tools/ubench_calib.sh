#!/bin/bash
# Calibrates the units of SQ_WAVE_CYCLES / SQ_BUSY_CYCLES on kernels of known residency (tools/ubench.hip: every wave of a
# launch is resident for the whole kernel, 8 or 4 waves per SIMD). On the GPU box, from the repo root.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O="$R/gpurun_out/ubench_calib"
mkdir -p "$O"
hipcc -O2 --offload-arch=gfx950 -w -o /tmp/ubench "$R/tools/ubench.hip"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- /tmp/ubench > "$O/ubench.txt" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d "$O/pmc" -- /tmp/ubench > "$O/pmc.log" 2>&1
find "$O" -name "*agent_info*" -delete
python3 - "$O" <<'P'
import csv, glob, sys, collections
O = sys.argv[1]
dur = {}
for f in glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = float(r["AverageNs"])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if k not in dur: continue
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    cyc = dur[k] * 2.4
    print(f"{k[:34]:36s} {dur[k]/1e3:9.1f} us  WAVE_CYCLES/(cycles*1024)={c.get('SQ_WAVE_CYCLES',0)/(cyc*1024):6.3f}  BUSY/cycles={c.get('SQ_BUSY_CYCLES',0)/cyc:7.3f}  "
          f"ACTIVE_VALU/INSTS={c.get('SQ_ACTIVE_INST_VALU',0)/max(c.get('SQ_INSTS_VALU',1),1):5.2f}  waves={c.get('SQ_WAVES',0):.0f}")
P
