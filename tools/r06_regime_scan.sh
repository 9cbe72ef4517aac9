#!/bin/bash
# round 6: default list granularity against both forced alternatives in the regimes where rounds 5's scan saw the default lose
# (trained opacities at 300 k / 800^2 and 4 M) and in the trained-scene-shaped ones (synthetic.py kind="surface").
# Output: gpurun_out/r06_regime_scan.txt
OUT=gpurun_out/r06_regime_scan.txt; mkdir -p gpurun_out; : > $OUT
run() {  # $1 = env assignments, $2 = bench args
  env $1 python bench.py --no-cpu-baseline --no-train-iter --no-live-traffic --steps 60 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; k=d['kernels_ms']
print('%.3f blk %d tiles/G %.2f bl %.3f fwd %.3f bwd %.3f gb %.3f' % (d['ms_per_step'], c['list_block_px'], c['num_rendered']/c['gaussians'], k['depth_sort'], k['render_fwd'], k['render_bwd'], k['gaussian_bwd']))"
}
for args in "--opacity trained" "--size 2048 --opacity trained" "--gaussians 2097152 --opacity trained" "--gaussians 4194304 --opacity trained" "--gaussians 300000 --size 800 --opacity trained" "--gaussians 300000 --size 800 --opacity surface" "--opacity surface" "--gaussians 2000000 --opacity surface" "--size 2048 --opacity surface"; do
  d=$(run "EOGS_NOP=1" "$args"); t=$(run "EOGS_BLOCK_SWITCH=1000 EOGS_DEPTH_SWITCH=0" "$args"); b=$(run "EOGS_BLOCK_SWITCH=0.5 EOGS_DEPTH_SWITCH=0.001" "$args")
  printf "%-52s\n    default  %s\n    per-tile %s\n    block    %s\n" "$args" "$d" "$t" "$b" >> $OUT
done
cat $OUT
