#!/bin/bash
# round 5: round 4's four 400-seed parity sweeps on the final round-5 library (plain chunks, ds_read_b128 layouts, flag-free
# records, DPP scans: every change is bit-neutral — tests/test_gpu_quad.py::test_round5_fast_paths_change_no_bit — so the counts must be
# round 4's: profiles/r04_sweeps.txt)
set -o pipefail
mkdir -p gpurun_out/r05
for r in 1000-1399 2000-2399 3000-3399 4000-4399; do
  rm -f gpurun_out/r05/stats_$r.jsonl
  EOGS_SWEEP_SEEDS=$r EOGS_PARITY_STATS=$PWD/gpurun_out/r05/stats_$r.jsonl timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomised_sweep > gpurun_out/r05/sweep_$r.log 2>&1
  echo "== seeds $r: $(tail -1 gpurun_out/r05/sweep_$r.log)"
  grep -E "^(FAILED|E  +Assertion)" gpurun_out/r05/sweep_$r.log | head -20
done
