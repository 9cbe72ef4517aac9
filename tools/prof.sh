set -x
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stats -- $B > $R/gpurun_out/prof/stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/prof/pmc_sq -- $B > $R/gpurun_out/prof/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_fetch -- $B > $R/gpurun_out/prof/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_write -- $B > $R/gpurun_out/prof/pmc_write.log 2>&1
cd $R/gpurun_out/prof && find . -name "*.csv" | head -30 && du -sh .
# keep only compact files
find . -name "*agent_info*" -delete
