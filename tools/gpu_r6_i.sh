#!/bin/bash
# Round 6, GPU session I: the output head inside the last GNN layer's launch -- tests of the network, forward time with and
# without it (alternating), kernel stats of the forward.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06i
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_07_net.py -q --timeout 240 > $OUT/pytest_net.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_net.log; tail -4 $OUT/pytest_net.log
for i in 1 2 3; do
  timeout 100 python tools/time_gnn_batch.py 2>/dev/null | tail -1 >> $OUT/gnn_head_fused.txt
  DACO_GNN_HEAD_FUSED=0 timeout 100 python tools/time_gnn_batch.py 2>/dev/null | tail -1 >> $OUT/gnn_head_separate.txt
done
echo fused; cat $OUT/gnn_head_fused.txt; echo separate; cat $OUT/gnn_head_separate.txt
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_gnn -o p -- python tools/time_gnn_batch.py > $OUT/stats_gnn.log 2>&1)
cp $OUT/stats_gnn/p_kernel_stats.csv $OUT/kernel_stats_gnn.csv 2>/dev/null; rm -rf $OUT/stats_gnn
head -7 $OUT/kernel_stats_gnn.csv | cut -c1-150
ls $OUT
