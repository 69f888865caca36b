#!/bin/bash
# Round 6, GPU session Q: the NLS thread rule after the sweep, the training step's three modes, gather / atomics in the GNN backward.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06q
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_07_net.py tests/test_gpu_03_two_opt.py -m gpu -q --timeout 240 > $OUT/pytest_net_nls.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_net_nls.log
tail -4 $OUT/pytest_net_nls.log | cut -c1-400
timeout 300 python tools/sweep_nls_threads.py 5 > $OUT/sweep_nls_threads.txt 2>&1; cut -c1-300 $OUT/sweep_nls_threads.txt
timeout 300 python tools/time_train_step.py 30 > $OUT/train_step_modes.txt 2>&1; cut -c1-1000 $OUT/train_step_modes.txt
for g in 0 1; do
  DACO_GNN_TRAIN_GATHER=$g TRAIN_MODES=graph timeout 200 python tools/time_train_step.py 40 --shape 100 > $OUT/train_gather$g.txt 2>&1; echo "GATHER=$g"; cut -c1-400 $OUT/train_gather$g.txt
done
