#!/bin/bash
# usage: bash tools/prof_lds.sh <tag>   LDS / stall / MFMA counters of the render kernels (on the GPU box, from the repo root)
set -e
TAG=${1:-lds}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O="$R/gpurun_out/prof_$TAG"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-iter --no-live-traffic $BENCH_ARGS"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$O/pmc_a" -- $B > "$O/pmc_a.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES SQ_INSTS_SALU --output-format csv -d "$O/pmc_b" -- $B > "$O/pmc_b.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES --output-format csv -d "$O/pmc_c" -- $B > "$O/pmc_c.log" 2>&1 || echo "pmc_c failed (counter names)" >> "$O/pmc_c.log"
find "$O" -name "*agent_info*" -delete
python3 "$R/tools/prof_summary.py" "$O" "$O/summary" > "$O/summary.txt" 2>&1
find "$O" -name "*counter_collection.csv" -delete
cat "$O/summary.txt"
