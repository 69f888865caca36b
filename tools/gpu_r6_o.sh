#!/bin/bash
# Round 6, GPU session O: kernel statistics of the captured training step (TSP-100 x 20 instances; TSP-500 x 8).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06o
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for shape in 100 500; do
  TRAIN_MODES=graph timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_g$shape -o g$shape --output-format csv -- python $R/tools/time_train_step.py 20 --shape $shape > $OUT/prof_graph$shape.log 2>&1
  f=$(find /tmp/prof_g$shape -name "*kernel_stats.csv" | head -1)
  echo "== $shape $f"; cp "$f" $OUT/kernel_stats_train_graph_$shape.csv; head -14 "$f" | cut -c1-160
done
