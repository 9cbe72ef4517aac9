#!/bin/bash
# Per-kernel averages of one bench workload (GPU box): tools/kstats.sh <tag> <bench args...>; environment switches are inherited.
# Prints the kernels above 1 % of the kernel time and keeps the list in gpurun_out/kstats_<tag>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; tag=$1; shift
O=$R/gpurun_out/kstats_$tag; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --no-cpu-baseline --no-train-iter --no-live-traffic --steps 60 "$@" > $O/run.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - $f > $R/gpurun_out/kstats_$tag.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs'])):
    if float(r['Percentage'])<1.0: continue
    print("%9.1f us  x%-5s %5.1f %%  %s"%(float(r['AverageNs'])/1e3, r['Calls'], float(r['Percentage']), r['Name'][:70]))
PY
find $O -name "*kernel_trace.csv" -delete
cat $R/gpurun_out/kstats_$tag.txt
