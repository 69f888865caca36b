#!/bin/bash
# Round 6, GPU session AN: edge_bwd with the lane's per-channel statistics read once per tile: parity, kernel time at both training shapes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06an
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_07_net.py -m gpu -q --timeout 300 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
for shape in 100 500; do
  TRAIN_MODES=graph timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_g$shape -o g$shape --output-format csv -- python $R/tools/time_train_step.py 20 --shape $shape > $OUT/prof_graph$shape.log 2>&1
  f=$(find /tmp/prof_g$shape -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/kernel_stats_train_graph_$shape.csv
  echo "== $shape"; grep workload $OUT/prof_graph$shape.log | cut -c1-260; grep -E "edge_bwd|nls_kernel|bwd_stats|node_lin_bwd" "$f" | cut -c1-40,180-330
done
