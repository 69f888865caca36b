#!/bin/bash
# Round 6, GPU session V: what the contended flush of the weight-gradient accumulators costs (measurement build: one atomic per lane).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06v
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in libdeepaco_hip.so libdeepaco_hip_ablate.so; do
  DACO_LIB_PATH=$R/deepaco_amd/lib/$lib TRAIN_MODES=graph timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$lib -o t --output-format csv -- python $R/tools/time_train_step.py 20 --shape 100 > $OUT/prof_$lib.log 2>&1
  f=$(find /tmp/prof_$lib -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/kernel_stats_$lib.csv
  echo "== $lib"; grep workload $OUT/prof_$lib.log | cut -c1-300; head -8 "$f" | cut -c1-50,200-330
done
