#!/bin/bash
# Round 6, GPU session R: smoke(), the whole GPU suite and the default bench after the training-path changes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06r
mkdir -p $OUT
cd $R
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log | cut -c1-300
timeout 1200 python -m pytest tests -m gpu -q --timeout 240 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log | cut -c1-300
timeout 900 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err; echo "bench rc=$?"
tail -1 $OUT/bench_default.log > $OUT/bench_default_line.json
cp gpurun_out/bench_extras.json $OUT/bench_extras.json 2>/dev/null
python - <<'PY'
import json,os
R=os.environ.get("GRAFT_REPO_ROOT",".")
j=json.load(open(os.path.join(R,"gpurun_out/r06r/bench_default_line.json")))
print({k:j[k] for k in ("value","ms_per_step","vs_baseline")}, j["roofline"]["frac"], j["cpu_baseline"]["value"])
try:
    e=json.load(open(os.path.join(R,"gpurun_out/r06r/bench_extras.json")))
    ex=e.get("extras",e)
    for k in ("train_step_tsp100_b20_a30","train_step_tsp500_b8_a50"):
        print(k, json.dumps(ex.get(k))[:700])
except Exception as ex_: print("extras:", ex_)
PY
