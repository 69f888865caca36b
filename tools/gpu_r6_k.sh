#!/bin/bash
# Round 6, GPU session K: the evidence of the round with the final library -- the suite, the default bench line, kernel stats and
# counter passes of the headline's kernels (tools/profile_r6.sh headline / gnn), counters.json / hbm_traffic.json.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06k
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 240 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default.log 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.log > $OUT/bench_default_line.json
head -c 600 $OUT/bench_default_line.json; echo
grep -E "^\[bench\]|real" $OUT/bench_default.err | tail -8
cp bench_extras.json $OUT/bench_extras.json 2>/dev/null
bash tools/profile_r6.sh r06k/prof headline > $OUT/profile_headline.log 2>&1
bash tools/profile_r6.sh r06k/prof gnn > $OUT/profile_gnn.log 2>&1
python tools/make_counters.py gpurun_out/r06k/prof > $OUT/make_counters.log 2>&1
cp profiles/counters.json profiles/hbm_traffic.json $OUT/ 2>/dev/null
tail -12 $OUT/make_counters.log
ls $OUT $OUT/prof | head -60
