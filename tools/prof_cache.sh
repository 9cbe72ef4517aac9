#!/bin/bash
# usage: [BENCH_ARGS="..."] bash tools/prof_cache.sh <tag>   (on the GPU box, from the repo root)
# L1 (TCP) / L2 (TCC) request counters of the bench command's kernels: how many line requests a kernel sends to L2 per byte it needs.
TAG=${1:-cache}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O="$R/gpurun_out/prof_$TAG"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$O/counters_all.txt" 2>&1 || true
grep -o "TC[PC]_[A-Z0-9_]*" "$O/counters_all.txt" | sort -u > "$O/counters_tc.txt"
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-iter --no-live-traffic $BENCH_ARGS"
pass() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$O/pmc_$n" -- $B > "$O/pmc_$n.log" 2>&1 || echo "pass $n failed"; }
pass tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum
pass tcp2 TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
find "$O" -name "*agent_info*" -delete
python3 - "$O" <<'P'
import csv, glob, sys, collections, json
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open(O + "/cache_mean_per_dispatch.json", "w"), indent=1)
for k, d in out.items():
    print(k, {c: round(v) for c, v in d.items()})
P
find "$O" -name "*counter_collection.csv" -delete
