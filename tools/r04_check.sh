#!/bin/bash
# round 4: schedule v31f (edge blocks first where tiles saturate) against the band mapping and against itself without the edge class (0x40)
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or seeded or deterministic or two_pass or long_block" > gpurun_out/r04/parity_v31f.log 2>&1 || { tail -30 gpurun_out/r04/parity_v31f.log; exit 1; }
tail -1 gpurun_out/r04/parity_v31f.log
bash tools/ab3.sh "EOGS_TILE_SCHED=0|EOGS_SCHED_FLAGS=0|EOGS_SCHED_FLAGS=0x40|EOGS_TILE_SCHED=0|EOGS_SCHED_FLAGS=0" "--opacity trained|--gaussians 2000000 --opacity trained|--opacity 0.3|--gaussians 300000 --size 800 --opacity trained|--size 2048 --opacity trained||--opacity 0.1" > gpurun_out/r04/ab_sched_v31f.txt 2>&1 || { tail -5 gpurun_out/r04/ab_sched_v31f.txt; exit 1; }
cat gpurun_out/r04/ab_sched_v31f.txt
