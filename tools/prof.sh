#!/bin/bash
# usage: [BENCH_ARGS="--size 2048 --opacity trained"] [STATS_ONLY=1] bash tools/prof.sh <tag>   (on the GPU box, from the repo root)
# rocprofv3 kernel stats + PMC passes (separate runs: FETCH_SIZE and WRITE_SIZE do not fit one pass) of the bench command;
# the program itself follows `--` (no env / bash -c hop: the profiler's preload initialises the GPU first).
set -e
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O="$R/gpurun_out/prof_$TAG"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-iter --no-live-traffic $BENCH_ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- $B > "$O/stats.log" 2>&1
[ -n "$KEEP_TRACE" ] || find "$O/stats" -name "*kernel_trace.csv" -delete   # tens of MB; gpurun_out is capped at 64 MiB
if [ -z "$STATS_ONLY" ]; then
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$O/pmc_sq" -- $B > "$O/pmc_sq.log" 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES --output-format csv -d "$O/pmc_sq2" -- $B > "$O/pmc_sq2.log" 2>&1 || \
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES --output-format csv -d "$O/pmc_sq2" -- $B > "$O/pmc_sq2.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- $B > "$O/pmc_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- $B > "$O/pmc_write.log" 2>&1
fi
find "$O" -name "*agent_info*" -delete
# condensed copy (what gets committed under profiles/): kernel_stats.csv + pmc_mean_per_dispatch.json + meta.json
python3 "$R/tools/prof_summary.py" "$O" "$O/summary" > "$O/summary.txt" 2>&1
find "$O" -name "*counter_collection.csv" -delete
tail -5 "$O/summary.txt"
