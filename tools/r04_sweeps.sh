#!/bin/bash
# round 4: the four 400-seed parity sweeps under the per-quantity rule, statistics kept
set -o pipefail
mkdir -p gpurun_out/r04
for r in 1000-1399 2000-2399 3000-3399 4000-4399; do
  rm -f gpurun_out/r04/stats_$r.jsonl
  EOGS_SWEEP_SEEDS=$r EOGS_PARITY_STATS=$PWD/gpurun_out/r04/stats_$r.jsonl timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomised_sweep > gpurun_out/r04/sweep_$r.log 2>&1
  echo "== seeds $r: $(tail -1 gpurun_out/r04/sweep_$r.log)"
  grep -E "^(FAILED|E  +Assertion)" gpurun_out/r04/sweep_$r.log | head -20
done
