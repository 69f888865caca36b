#!/bin/bash
# Round 6, GPU session BF: the gather form of the backward from 16 k edges: parity, the training step.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06bf
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_07_net.py tests/test_gpu_04_grad.py -m gpu -q --timeout 300 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log | cut -c1-300
TRAIN_MODES=graph timeout 300 python tools/time_train_step.py 40 2>/dev/null | tee -a $OUT/train_step.txt | cut -c1-200
