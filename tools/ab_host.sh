#!/bin/bash
# Tuning aid (GPU box): host sensitivity of the eager step, library variants side by side: step time over kernel sum, 8 runs of 100 steps each
IFS=' ' read -ra LIBS <<< "$1"; OUT=gpurun_out/ab_host.txt; : > $OUT
cp eogs2_amd/libeogs_rast_hip.so /tmp/eogs_keep.so
for rep in 1 2 3 4; do
  for l in "${LIBS[@]}"; do
    cp eogs2_amd/variants/$l.so eogs2_amd/libeogs_rast_hip.so
    python bench.py --no-cpu-baseline --no-train-iter --no-live-traffic --steps 100 $2 > gpurun_out/ab_tmp.json 2>/dev/null || { cp /tmp/eogs_keep.so eogs2_amd/libeogs_rast_hip.so; exit 1; }
    python - "[$l]" >> $OUT <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
h=d['host']
print(sys.argv[1], 'ms=%.4f'%d['ms_per_step'], 'ksum=%.4f'%h['kernel_sum_ms'], 'ratio=%.3f'%h['step_over_kernel_sum'], h.get('timed_step_intervals_ms'))
PY
  done
done
cp /tmp/eogs_keep.so eogs2_amd/libeogs_rast_hip.so
cat $OUT
