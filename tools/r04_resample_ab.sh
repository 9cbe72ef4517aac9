#!/bin/bash
# round 4: the resample backward's two tile kernels (EOGS_RESAMPLE_BWD=1: LDS float atomics, 2: bucketed gather) per configuration and
# altitude field, kernel averages from rocprofv3
set -e
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r04; cd /tmp; export TMPDIR=/tmp
for mode in smooth rand; do for form in ${FORMS:-1 2}; do for cfg in "2 5 4" "2 1 1" "1 5 4"; do
  tag=$(echo $cfg | tr ' ' '_')_${mode}_f$form
  EOGS_RESAMPLE_BWD=$form rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04/rs3_$tag -- python3 $R/tools/resample_probe.py $cfg 1024 $mode > $R/gpurun_out/r04/rs3_$tag.log 2>&1
  f=$(find $R/gpurun_out/r04/rs3_$tag -name "*kernel_stats.csv" | head -1)
  python3 - $f "$cfg $mode form $form" <<'PY'
import csv,sys
out=[]
for r in csv.DictReader(open(sys.argv[1])):
    if 'resample' in r['Name']: out.append('%s %.1f'%(r['Name'].split('resample_')[1].split('(')[0][:28], float(r['AverageNs'])/1e3))
print(sys.argv[2], '|', ' | '.join(sorted(out)))
PY
  rm -rf $R/gpurun_out/r04/rs3_$tag
done; done; done
