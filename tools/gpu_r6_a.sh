#!/bin/bash
# Round 6, GPU session A: the whole -m gpu suite (new: the head-row sampler at the bench shape, every ant; chi-square of the HIP
# draws), the full-power best-cost gap study (CPU side in the background while the tests run), kernel stats at B = 1, the
# force-dist check, then the default bench line.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06a
mkdir -p $OUT
cd $R
nproc > $OUT/nproc.txt
(python tools/best_cost_gap.py 64 20 3 $OUT/best_cost_gap.json > $OUT/best_cost_gap.log 2>&1) &
GAP=$!
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b1 -o p -- python tools/b1_modes.py 200 > $OUT/b1_modes.log 2>&1)
cp $OUT/stats_b1/p_kernel_stats.csv $OUT/kernel_stats_b1.csv 2>/dev/null; rm -rf $OUT/stats_b1
cd $R
timeout 900 python tools/check_force_dist.py 20 > $OUT/force_dist.log 2>&1
echo "force-dist rc=$?" >> $OUT/force_dist.log
wait $GAP
cat $OUT/best_cost_gap.log | tail -3
( time python bench.py ) > $OUT/bench_default.log 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.log | head -c 3000
tail -5 $OUT/bench_default.err
cp bench_extras.json $OUT/ 2>/dev/null
ls $OUT
