#!/bin/bash
# round 5: the five 400-seed parity sweeps with the back-to-front switch at $1 (default 0.02) instead of the library's default
set -o pipefail
SW=${1:-0.02}; mkdir -p gpurun_out/r05b
for r in 1000-1399 2000-2399 3000-3399 4000-4399 5000-5399; do
  EOGS_BTF_SWITCH=$SW EOGS_SWEEP_SEEDS=$r timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomised_sweep > gpurun_out/r05b/sweep_$r.log 2>&1
  echo "== EOGS_BTF_SWITCH=$SW seeds $r: $(tail -1 gpurun_out/r05b/sweep_$r.log)"
  grep -E "^(FAILED|E  +Assertion)" gpurun_out/r05b/sweep_$r.log | head -20
done
