# usage: [BENCH_ARGS="--size 2048 --opacity trained"] [STATS_ONLY=1] bash tools/prof.sh <tag>   (on the GPU box, from the repo root)
TAG=${1:-run}
mkdir -p gpurun_out/prof_$TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-iter $BENCH_ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1
[ -n "$KEEP_TRACE" ] || find $O/stats -name "*kernel_trace.csv" -delete   # tens of MB; gpurun_out is capped at 64 MiB
if [ -n "$STATS_ONLY" ]; then find $O -name "*agent_info*" -delete; exit 0; fi
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc_sq2 -- $B > $O/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
find $O -name "*agent_info*" -delete
find $O -name "*.csv" | head -20
tail -3 $O/pmc_sq2.log
