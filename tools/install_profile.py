"""Copies the summaries of tools/prof.sh + tools/prof_lds.sh runs (gpurun_out/prof_<tag>/summary, gpurun_out/prof_<ldstag>/summary) into
profiles/<name>/ — kernel_stats.csv, meta.json and ONE pmc_mean_per_dispatch.json that holds both runs' counters — and says whether the
profile is of the sources in the tree:   python3 tools/install_profile.py <tag> <ldstag> <name>"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eogs2_amd.build import source_hash  # noqa: E402

tag, lds, name = sys.argv[1:4]
src, l, d = f"{ROOT}/gpurun_out/prof_{tag}/summary", f"{ROOT}/gpurun_out/prof_{lds}/summary", f"{ROOT}/profiles/{name}"
os.makedirs(d, exist_ok=True)
for f in ("kernel_stats.csv", "meta.json"):
    shutil.copy(f"{src}/{f}", f"{d}/{f}")
a, b = json.load(open(f"{src}/pmc_mean_per_dispatch.json")), json.load(open(f"{l}/pmc_mean_per_dispatch.json"))
for k, v in b.items():
    for c, x in v.items():
        a.setdefault(k, {}).setdefault(c, x)
json.dump(a, open(f"{d}/pmc_mean_per_dispatch.json", "w"), indent=1, sort_keys=True)
print(name, "is of the tree's kernel sources:", json.load(open(f"{d}/meta.json")).get("kernel_source_sha256") == source_hash())
