#!/bin/bash
# Round 6, GPU session BN: gather_bwd with eight edges' rows in flight, knn_graph_kernel with a row's coordinates read up front:
# parity (net / gradient / pipeline tests), the training steps against the previous build on one box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06bn
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_07_net.py tests/test_gpu_04_grad.py tests/test_gpu_14_surface.py -m gpu -q --timeout 300 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -2 $OUT/pytest.log | cut -c1-300
for i in 1 2; do
  for v in new prev; do
    L=$R/deepaco_amd/lib/libdeepaco_hip.so; [ $v = prev ] && L=$R/deepaco_amd/lib/libdeepaco_hip_prev.so
    echo "== $v" | tee -a $OUT/train_step.txt
    DACO_LIB_PATH=$L TRAIN_MODES=graph timeout 300 python tools/time_train_step.py 40 2>/dev/null | tee -a $OUT/train_step.txt | cut -c1-170
  done
done
