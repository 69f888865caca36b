#!/bin/bash
# Round 6, GPU session BO (final evidence, the final tree of round 6): smoke(), the whole GPU suite, the default bench on the driver's command line with its
# extras, kernel statistics of the same command (rocprofv3 --kernel-trace --stats) and of the captured training step.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06bo
mkdir -p $OUT
cd $R
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log | cut -c1-300
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.log 2> $OUT/bench_default.err; echo "bench rc=$?"
tail -1 $OUT/bench_default.log > $OUT/bench_default_line.json
cp gpurun_out/bench_extras.json $OUT/bench_extras.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_headline -o p -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras > $OUT/stats_headline.log 2>&1)
cp /tmp/stats_headline/p_kernel_stats.csv $OUT/kernel_stats_headline.csv 2>/dev/null || cp $(find /tmp/stats_headline -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_headline.csv
TRAIN_MODES=graph timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_train -o p -- python $R/tools/time_train_step.py 20 --shape 100 > $OUT/stats_train.log 2>&1
cp $(find /tmp/stats_train -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_train_graph_tsp100.csv
cd $R
python - <<'PY'
import json,os,csv
R=os.environ.get("GRAFT_REPO_ROOT",".")
O=os.path.join(R,"gpurun_out/r06bo")
j=json.load(open(os.path.join(O,"bench_default_line.json")))
print({k:j[k] for k in ("value","ms_per_step","vs_baseline")}, "frac", j["roofline"]["frac"], "kernel_ms", j["roofline"].get("kernel_ms"), "cpu", j["cpu_baseline"]["value"])
rows=list(csv.DictReader(open(os.path.join(O,"kernel_stats_headline.csv"))))
for r in rows[:4]: print(r["Name"][:70], r["Calls"], r["AverageNs"])
e=json.load(open(os.path.join(O,"bench_extras.json")))
ex=e.get("extras",e)
for k in ("headline_compact_tours","headline_two_streams","headline_b1","headline_b8","train_step_tsp100_b20_a30"):
    v=ex.get(k,{}); print(k, v.get("value"), v.get("ms_per_step"), v.get("error"))
PY
