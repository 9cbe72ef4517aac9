import json,sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(l["ms_per_step"], l["host"]); print(l.get("graphed_step"))
print("train_iter", l["train_iter"]["ms_per_iter"], "fused", l["train_iter_fused"]["ms_per_iter"], "graphed", {k:l.get("train_iter_fused_graphed",{}).get(k) for k in ("ms_per_iter","graph")})
