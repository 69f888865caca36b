#!/bin/bash
# Round 6, GPU session BC: the GNN training kernels issue a tile's global loads together (node_lin_bwd, edge_bwd, edge_pre, node_pre,
# head_bwd): parity (net / gradient tests), the training step against the previous build on one box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06bc
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_07_net.py tests/test_gpu_04_grad.py -m gpu -q --timeout 300 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log | cut -c1-300
for i in 1 2; do
  echo "== new" | tee -a $OUT/train_step.txt
  TRAIN_MODES=graph timeout 300 python tools/time_train_step.py 40 2>/dev/null | tee -a $OUT/train_step.txt | cut -c1-400
  echo "== previous build" | tee -a $OUT/train_step.txt
  DACO_LIB_PATH=$R/deepaco_amd/lib/libdeepaco_hip_prev.so TRAIN_MODES=graph timeout 300 python tools/time_train_step.py 40 2>/dev/null | tee -a $OUT/train_step.txt | cut -c1-400
done
echo "== new, TSP-500" | tee -a $OUT/train_step.txt
TRAIN_MODES=graph timeout 300 python tools/time_train_step.py 20 --shape 500 2>/dev/null | tee -a $OUT/train_step.txt | cut -c1-400
echo "== previous build, TSP-500" | tee -a $OUT/train_step.txt
DACO_LIB_PATH=$R/deepaco_amd/lib/libdeepaco_hip_prev.so TRAIN_MODES=graph timeout 300 python tools/time_train_step.py 20 --shape 500 2>/dev/null | tee -a $OUT/train_step.txt | cut -c1-400
cd /tmp && export TMPDIR=/tmp
TRAIN_MODES=graph timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o g --output-format csv -- python $R/tools/time_train_step.py 20 --shape 100 > $OUT/prof.log 2>&1
f=$(find /tmp/prof_g -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats_train_graph_tsp100.csv; head -12 "$f" | cut -c1-60,200-330
