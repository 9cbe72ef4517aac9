set -e
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for cfg in "2 5 4" "2 1 1" "1 5 4"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04/rs_$tag -- python3 $R/tools/resample_probe.py $cfg > $R/gpurun_out/r04/rs_$tag.log 2>&1
  grep "ms per fwd" $R/gpurun_out/r04/rs_$tag.log
  f=$(find $R/gpurun_out/r04/rs_$tag -name "*kernel_stats.csv" | head -1)
  python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'resample' in r['Name']: print('   ', r['Name'][:60], r['Calls'], 'avg_us=%.1f'%(float(r['AverageNs'])/1e3))
PY
  find $R/gpurun_out/r04/rs_$tag -name "*kernel_trace.csv" -delete
done
