#!/bin/bash
# tools/ab2.sh VAR "v1 v2" "bench args 1|bench args 2|..." -> one line per (value, args)
VAR=$1; VALS=$2; IFS='|' read -ra ARGS <<< "$3"; OUT=gpurun_out/ab2_${VAR}.txt; : > $OUT
for a in "${ARGS[@]}"; do
  for v in $VALS; do
    env $VAR=$v python bench.py --no-cpu-baseline --no-train-iter --steps 60 $a > gpurun_out/ab_tmp.json 2>/dev/null || exit 1
    python - "$VAR=$v [$a]" >> $OUT <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
c=d['config']
print(sys.argv[1], 'tiles/G=%.2f'%(c['num_rendered']/c['gaussians']), 'blk', c['list_block_px'], 'ms=%.4f'%d['ms_per_step'], {k:round(v,4) for k,v in d['kernels_ms'].items() if k.startswith('render')})
PY
  done
done
cat $OUT
