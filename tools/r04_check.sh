#!/bin/bash
# round 4: record layout A/B (compile-time), new tests
set -o pipefail
mkdir -p gpurun_out/r04
for soa in 0 1 0 1; do
  python -m eogs2_amd.build --force -DEOGS_REC_SOA=$soa > /dev/null 2>&1 || exit 1
  for args in "" "--opacity trained" "--opacity 0.1" "--size 2048"; do
    python bench.py --no-cpu-baseline --no-train-iter --steps 60 $args > gpurun_out/ab_tmp.json 2>/dev/null || exit 1
    python - "[REC_SOA=$soa] [$args]" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
print(sys.argv[1], 'ms=%.4f'%d['ms_per_step'], {k:round(v,4) for k,v in d['kernels_ms'].items()})
PY
  done
done | tee gpurun_out/r04/ab_rec_soa.txt
python -m eogs2_amd.build --force > /dev/null 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_altonly.py tests/test_gpu_graph.py tests/test_gpu_parity.py tests/test_gpu_quad.py -m gpu -q > gpurun_out/r04/new_tests.log 2>&1
tail -12 gpurun_out/r04/new_tests.log | cut -c1-220
