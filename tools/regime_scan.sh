#!/bin/bash
# Regime scan (on the GPU box, from the repo root): fwd+bwd ms of the library's default choice of list granularity / render
# kernel against the two forced alternatives, over Gaussian count, image size and opacity. Output: gpurun_out/regime_scan.txt
OUT=gpurun_out/regime_scan.txt; mkdir -p gpurun_out; : > $OUT
run() {  # $1 = env assignments, $2 = bench args
  env $1 python bench.py --no-cpu-baseline --no-train-iter --no-live-traffic --steps 60 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('%.3f blk %d tiles/G %.2f' % (d['ms_per_step'], c['list_block_px'], c['num_rendered']/c['gaussians']))"
}
for args in "--opacity init" "--opacity 0.03" "--opacity 0.1" "--opacity 0.2" "--opacity 0.5" "--opacity trained" "--size 2048" "--size 2048 --opacity trained" "--gaussians 2097152 --opacity 0.1" "--gaussians 2097152 --opacity 0.3" "--gaussians 2097152 --opacity trained" "--gaussians 4194304 --opacity trained" "--gaussians 300000 --size 800" "--gaussians 300000 --size 800 --opacity trained" "--gaussians 300000 --size 1600" "--gaussians 300000 --size 800 --opacity surface" "--opacity surface" "--gaussians 2000000 --opacity surface" "--size 2048 --opacity surface"; do
  d=$(run "EOGS_NOP=1" "$args"); t=$(run "EOGS_BLOCK_SWITCH=1000 EOGS_DEPTH_SWITCH=0" "$args"); b=$(run "EOGS_BLOCK_SWITCH=0.5 EOGS_DEPTH_SWITCH=0.001" "$args")
  printf "%-52s default %-28s per-tile %-28s block %s\n" "$args" "$d" "$t" "$b" >> $OUT
done
cat $OUT
