#!/bin/bash
# round 4: kernel-level profile of the full-chain iteration (examples/train_synthetic.py at 1 M / 1024^2, three renders)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04/example_prof; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/examples/train_synthetic.py --gaussians 1048576 --size 1024 --iters 24 --no-prune --sun-altitude-only --random-camera $EX_ARGS > $O/run.log 2>&1
tail -2 $O/run.log
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - $f 24 <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
n=int(sys.argv[2])
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel time per iteration (all %d iterations incl. the first): %.3f ms"%(n, tot/n/1e6))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:28]:
    print("  %8.1f us/iter  x%-5s avg %8.1f us  %s"%(float(r['TotalDurationNs'])/n/1e3, r['Calls'], float(r['AverageNs'])/1e3, r['Name'][:90]))
PY
find $O -name "*kernel_trace.csv" -delete
