#!/bin/bash
# Round 6, GPU session AO: one gradient flush per workgroup in the GNN backward kernels: parity, the training step's three modes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06ao
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_07_net.py tests/test_gpu_04_grad.py -m gpu -q --timeout 300 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log | cut -c1-300
timeout 300 python tools/time_train_step.py 40 2>/dev/null | tee $OUT/train_step_modes.txt | cut -c1-900
cd /tmp && export TMPDIR=/tmp
for shape in 100 500; do
  TRAIN_MODES=graph timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_g$shape -o g$shape --output-format csv -- python $R/tools/time_train_step.py 20 --shape $shape > $OUT/prof_graph$shape.log 2>&1
  f=$(find /tmp/prof_g$shape -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/kernel_stats_train_graph_$shape.csv
  echo "== $shape"; head -9 "$f" | cut -c1-40,200-330
done
