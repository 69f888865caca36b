#!/bin/bash
# Round 6, GPU session AL: the default bench (driver's command line) with the final tree: line + extras.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06al
mkdir -p $OUT
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.log 2> $OUT/bench_default.err; echo "bench rc=$?"
tail -1 $OUT/bench_default.log > $OUT/bench_default_line.json
cp gpurun_out/bench_extras.json $OUT/bench_extras.json 2>/dev/null
python - <<'PY'
import json,os
R=os.environ.get("GRAFT_REPO_ROOT",".")
O=os.path.join(R,"gpurun_out/r06al")
j=json.load(open(os.path.join(O,"bench_default_line.json")))
print({k:j[k] for k in ("value","ms_per_step")}, "frac", j["roofline"]["frac"], "kernel_ms", j["roofline"].get("kernel_ms"), "sustained", j["sustained"]["value"], "cpu", j["cpu_baseline"]["value"])
e=json.load(open(os.path.join(O,"bench_extras.json")))
ex=e.get("extras",e)
for k in ("headline_compact_tours","headline_two_streams","headline_b1","headline_b8","train_step_tsp100_b20_a30","train_step_tsp500_b8_a50","gnn_tsp500_k50_b64","c3_tsp500_nls_a256_b64"):
    v=ex.get(k,{}); print(k, v.get("value"), v.get("ms_per_step"), v.get("error"), (v.get("compact_tours") or {}).get("value"))
PY
