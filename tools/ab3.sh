#!/bin/bash
# Tuning aid (on the GPU box, from the repo root): A/B of environment switches.
# tools/ab3.sh "ENV1=a ENV2=b|ENV1=c" "bench args 1|bench args 2" -> one line per (env set, args) with all kernel groups
IFS='|' read -ra ENVS <<< "$1"; IFS='|' read -ra ARGS <<< "$2"; OUT=gpurun_out/ab3.txt; : > $OUT
for a in "${ARGS[@]}"; do
  for e in "${ENVS[@]}"; do
    env $e python bench.py --no-cpu-baseline --no-train-iter --no-live-traffic --steps 60 $a > gpurun_out/ab_tmp.json 2>/dev/null || exit 1
    python - "[$e] [$a]" >> $OUT <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
c=d['config']
print(sys.argv[1], 'tiles/G=%.2f'%(c['num_rendered']/c['gaussians']), 'blk', c['list_block_px'], 'ms=%.4f'%d['ms_per_step'], {k:round(v,4) for k,v in d['kernels_ms'].items()})
PY
  done
done
cat $OUT
