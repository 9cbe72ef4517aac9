"""Condenses a gpurun_out/prof_<tag> directory (tools/prof.sh) into profiles/<name>/ : kernel_stats.csv +
pmc_mean_per_dispatch.json (mean counter value per dispatch, per kernel)."""
import collections, csv, glob, json, os, re, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")[:60]
for f in glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, "kernel_stats.csv"))
out = {}
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
json.dump(out, open(os.path.join(dst, "pmc_mean_per_dispatch.json"), "w"), indent=1, sort_keys=True)
# which kernels these numbers belong to: bench.py only replays counter-derived figures (HBM traffic, VALU instructions)
# from a profile whose kernel sources are the ones it is running
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eogs2_amd.build import source_hash  # noqa: E402

json.dump({"kernel_source_sha256": source_hash(), "bench_args": os.environ.get("BENCH_ARGS", "")}, open(os.path.join(dst, "meta.json"), "w"))
for k in [k for k in out if k.startswith("render_")]:
    if k in out:
        print(k, {c: f"{v:.3g}" for c, v in sorted(out[k].items())})
