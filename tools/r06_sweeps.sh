#!/bin/bash
# round 6: the six 400-seed parity sweeps of rounds 4-5 (profiles/r05_sweeps.txt) on the round-6 library: every backward kernel walks
# back to front (no EOGS_BTF_SWITCH), and ill-conditioned elements are decided by the float64 arbiter (tests/parity_cases.py
# ARB_FACTOR) instead of the tuned bound SENS_RTOL.    tools/r06_sweeps.sh [ranges ...]
set -o pipefail
mkdir -p gpurun_out/r06
for r in ${@:-1000-1399 2000-2399 3000-3399 4000-4399 5000-5399 6000-6399}; do
  rm -f gpurun_out/r06/stats_$r.jsonl
  EOGS_SWEEP_SEEDS=$r EOGS_PARITY_STATS=$PWD/gpurun_out/r06/stats_$r.jsonl timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k randomised_sweep > gpurun_out/r06/sweep_$r.log 2>&1
  echo "== seeds $r: $(tail -1 gpurun_out/r06/sweep_$r.log)"
  grep -E "^(FAILED|E  +Assertion)" gpurun_out/r06/sweep_$r.log | head -20
  grep -hE "arbiter REJECTS|decided by the arbiter" gpurun_out/r06/sweep_$r.log | sort | uniq -c | sort -rn | head -5
done
