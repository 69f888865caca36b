#!/bin/bash
# Round 6, GPU session H: the whole suite, then the default bench line (CPU leg, extras, live counter passes) as the driver runs it.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06h
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 240 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
( time python bench.py ) > $OUT/bench_default.log 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.log | head -c 3800; echo
grep -E "^\[bench\]|real" $OUT/bench_default.err | tail -12
cp bench_extras.json $OUT/ 2>/dev/null
python -c "
import json; j=json.load(open('bench_extras.json')); e=j.get('extras',{})
for k in ('headline_b1','headline_b8','gnn_forward_b64_n500_k50','c3_tsp500_nls_a256_b64','c4_cvrp100_a512_b256','c5_share_tsp1000_a2048_b64','c2_tsp100_a512_b256'):
    v=e.get(k); print(k, (v.get('value'), v.get('ms_per_step')) if isinstance(v,dict) else v)
print(j.get('best_cost_gap',{}).get('samplers')); print(j['roofline'].get('traffic'), j['roofline'].get('traffic_source'))
"
ls $OUT
