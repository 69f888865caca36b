#!/bin/bash
# Round 6, GPU session BD: bwd_stats with eight rows per trip, sample_backward with one memory round trip per step:
# parity (gradient / net / CVRP / sibling tests), the training step against the previous build on one box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06bd
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_07_net.py tests/test_gpu_04_grad.py tests/test_gpu_02_cvrp.py tests/test_gpu_05_siblings.py -m gpu -q --timeout 300 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log | cut -c1-300
for i in 1 2; do
  echo "== new" | tee -a $OUT/train_step.txt
  TRAIN_MODES=graph timeout 300 python tools/time_train_step.py 40 2>/dev/null | tee -a $OUT/train_step.txt | cut -c1-200
  echo "== previous build (before this session's training-kernel changes)" | tee -a $OUT/train_step.txt
  DACO_LIB_PATH=$R/deepaco_amd/lib/libdeepaco_hip_prev.so TRAIN_MODES=graph timeout 300 python tools/time_train_step.py 40 2>/dev/null | tee -a $OUT/train_step.txt | cut -c1-200
done
cd /tmp && export TMPDIR=/tmp
TRAIN_MODES=graph timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o g --output-format csv -- python $R/tools/time_train_step.py 20 --shape 100 > $OUT/prof.log 2>&1
f=$(find /tmp/prof_g -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats_train_graph_tsp100.csv
TRAIN_MODES=graph timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o g --output-format csv -- python $R/tools/time_train_step.py 20 --shape 500 > $OUT/prof500.log 2>&1
f=$(find /tmp/prof_h -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats_train_graph_tsp500.csv
