set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04/dist1_prof; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --force-dist --steps 200 --warmup 5 --no-cpu-baseline --no-train-iter --no-live-traffic --ar-chunks 1 --ar-algo all_reduce > $O/run.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:26]:
    print("  x%-6s avg %8.1f us  %s"%(r['Calls'], float(r['AverageNs'])/1e3, r['Name'][:100]))
PY
find $O -name "*kernel_trace.csv" -delete
