#!/bin/bash
# round 4: forward occupancy under the tile schedule (compile-time EOGS_FW), the two sweep seeds beyond SENS_RTOL
set -o pipefail
mkdir -p gpurun_out/r04
for fw in 6 7 5 8 6 7; do
  python -m eogs2_amd.build --force -DEOGS_FW=$fw > /dev/null 2>&1 || exit 1
  for args in "" "--opacity trained" "--opacity 0.1" "--size 2048"; do
    python bench.py --no-cpu-baseline --no-train-iter --steps 60 $args > gpurun_out/ab_tmp.json 2>/dev/null || exit 1
    python - "[EOGS_FW=$fw] [$args]" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
print(sys.argv[1], 'ms=%.4f'%d['ms_per_step'], {k:round(v,4) for k,v in d['kernels_ms'].items()})
PY
  done
done | tee gpurun_out/r04/ab_fw.txt
python -m eogs2_amd.build --force > /dev/null 2>&1
for sd in 1259-1259 4275-4275; do
EOGS_SENS_RTOL=0.13 EOGS_SWEEP_SEEDS=$sd timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomised_sweep 2>&1 | tail -1
done
