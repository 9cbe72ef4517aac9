#!/bin/bash
# round 4: tile schedule (interleaved units) A/B on one box, then the new tests, then the bench line
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_altonly.py tests/test_gpu_graph.py tests/test_gpu_speculation.py -m gpu -q -x > gpurun_out/r04/new_tests.log 2>&1
tail -25 gpurun_out/r04/new_tests.log
bash tools/ab3.sh "EOGS_TILE_SCHED=0|EOGS_SCHED_UNIT=1|EOGS_SCHED_UNIT=2|EOGS_SCHED_UNIT=4|EOGS_SCHED_UNIT=8|EOGS_SCHED_UNIT=32|EOGS_TILE_SCHED=0" "|--opacity trained|--opacity 0.1|--gaussians 300000 --size 800 --opacity trained|--size 2048 --opacity trained|--size 2048" > gpurun_out/r04/ab_sched_units.txt 2>&1 || { tail -5 gpurun_out/r04/ab_sched_units.txt; exit 1; }
cat gpurun_out/r04/ab_sched_units.txt
for op in init trained; do
  timeout -k 10 600 python tools/wave_trace.py --opacity $op --envs "EOGS_TILE_SCHED=0|EOGS_SCHED_UNIT=4" --out gpurun_out/r04/wave_trace_units_$op.json > gpurun_out/r04/wave_trace_units_$op.log 2>&1 || { tail -30 gpurun_out/r04/wave_trace_units_$op.log; exit 1; }
  python - $op <<'PY'
import json,sys
d=json.load(open('gpurun_out/r04/wave_trace_units_%s.json'%sys.argv[1]))
for mode,v in d.items():
    for k,r in v.items():
        print(sys.argv[1],mode,k,'span',r['span_us'],'mean_res',round(r['mean_resident_per_simd'],2),'tenths',r['resident_per_simd_by_tenth'],'gap',round(r['slot_gap_us']['mean'],2))
        print('    xcc_last_end',[round(x) for x in r['xcc_last_end_us']],'xcc_work',[round(x) for x in r['xcc_wave_us_per_simd']])
PY
done
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_full.json 2> gpurun_out/r04/bench_full.err
python tools/show_bench.py gpurun_out/r04/bench_full.json 2>/dev/null | tail -40 || tail -c 3000 gpurun_out/r04/bench_full.json
