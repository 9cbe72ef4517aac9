#!/bin/bash
# round 4: persistent-wave A/B on one box (parity subset under each mode, wave trace, bench A/B)
set -e -o pipefail
mkdir -p gpurun_out/r04
MODES=${MODES:-"3 4 5 6"}
for m in $MODES; do
  EOGS_PERSIST=$m timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or seeded or deterministic" > gpurun_out/r04/parity_persist$m.log 2>&1 || { tail -30 gpurun_out/r04/parity_persist$m.log; exit 1; }
  tail -1 gpurun_out/r04/parity_persist$m.log
done
TM=$(echo "0 $MODES" | tr ' ' ',')
timeout -k 10 600 python tools/wave_trace.py --modes $TM --out gpurun_out/r04/wave_trace_init_b.json > gpurun_out/r04/wave_trace_init_b.log 2>&1 || { tail -30 gpurun_out/r04/wave_trace_init_b.log; exit 1; }
cat gpurun_out/r04/wave_trace_init_b.log
E="EOGS_PERSIST=0"; for m in $MODES; do E="$E|EOGS_PERSIST=$m"; done
bash tools/ab3.sh "$E|EOGS_PERSIST=0" "|--opacity trained|--opacity 0.1" > gpurun_out/r04/ab_persist_b.txt 2>&1
cat gpurun_out/r04/ab_persist_b.txt
