#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4n; mkdir -p $O
timeout 900 python bench.py --no-cpu --min-seconds 0 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","knobs")}, d["roofline"]["frac"], d["roofline"]["kernel_ms"])
for k,v in d["extras"].items():
    if isinstance(v,dict):
        print(k, {kk:(vv if not isinstance(vv,(dict,list)) else '...') for kk,vv in v.items() if kk in ("value","ms_per_step","error","seconds","speedup_whole_iteration")})
        if k=="headline_scan_sparse": print(json.dumps(v)[:1500])
        if k.startswith("cvrp_local"): print(json.dumps(v)[:900])
PY
tail -3 $O/bench.err
