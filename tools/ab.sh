#!/bin/bash
# A/B of a tuning switch on the GPU box: tools/ab.sh VAR "v1 v2 ..." -> kernels_ms per value, init and trained opacities
VAR=$1; VALS=$2; OUT=gpurun_out/ab_${VAR}.txt; : > $OUT
for v in $VALS; do
  for op in init trained; do
    env $VAR=$v python bench.py --no-cpu-baseline --no-train-iter --opacity $op --steps 60 > gpurun_out/ab_tmp.json 2>/dev/null || exit 1
    python - "$VAR=$v $op" >> $OUT <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
print(sys.argv[1], 'ms_per_step=%.4f'%d['ms_per_step'], {k:round(v,4) for k,v in d['kernels_ms'].items()})
PY
  done
done
cat $OUT
