#!/bin/bash
# A/B of an environment switch: tools/ab_env.sh "VAR=0 VAR=1 VAR=0 VAR=1" "bench args 1|bench args 2"
IFS=' ' read -ra ENVS <<< "$1"; IFS='|' read -ra ARGS <<< "$2"; OUT=gpurun_out/ab_env.txt; : > $OUT
for a in "${ARGS[@]}"; do
  for e in "${ENVS[@]}"; do
    env $e python bench.py --no-cpu-baseline --no-train-iter --no-live-traffic --steps 60 $a > gpurun_out/ab_tmp.json 2>/dev/null || exit 1
    python - "[$e] [$a]" >> $OUT <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
c=d['config']
print(sys.argv[1], 'tiles/G=%.2f'%(c['num_rendered']/c['gaussians']), 'ms=%.4f'%d['ms_per_step'], {k:round(v,4) for k,v in d['kernels_ms'].items()})
PY
  done
done
cat $OUT
