#!/bin/bash
# Tuning aid (GPU box, from the repo root): A/B of whole library builds (compile-time variants under eogs2_amd/variants/*.so, built in
# the container with extra -D flags). tools/ab_lib.sh "base fw7 base fw7" "bench args 1|bench args 2" -> one line per (variant, args)
IFS=' ' read -ra LIBS <<< "$1"; IFS='|' read -ra ARGS <<< "$2"; OUT=gpurun_out/ab_lib.txt; : > $OUT
cp eogs2_amd/libeogs_rast_hip.so /tmp/eogs_keep.so
for a in "${ARGS[@]}"; do
  for l in "${LIBS[@]}"; do
    cp eogs2_amd/variants/$l.so eogs2_amd/libeogs_rast_hip.so
    python bench.py --no-cpu-baseline --no-train-iter --no-live-traffic --steps 60 $a > gpurun_out/ab_tmp.json 2>/dev/null || { cp /tmp/eogs_keep.so eogs2_amd/libeogs_rast_hip.so; exit 1; }
    python - "[$l] [$a]" >> $OUT <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
c=d['config']
print(sys.argv[1], 'tiles/G=%.2f'%(c['num_rendered']/c['gaussians']), 'ms=%.4f'%d['ms_per_step'], {k:round(v,4) for k,v in d['kernels_ms'].items()})
PY
  done
done
cp /tmp/eogs_keep.so eogs2_amd/libeogs_rast_hip.so
cat $OUT
