#!/bin/bash
# Round 6, GPU session AQ: nls_kernel holds four words per entry across the gather (the two edge lengths are read afterwards):
# 192 threads x 4 entries per round fit 80 registers.  Parity of both forms, config 3 at 3 / 4 entries per thread, 192 / 256 threads.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06aq
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_03_two_opt.py tests/test_gpu_15_full_batch.py -m gpu -q --timeout 600 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -2 $OUT/pytest.log | cut -c1-300
DACO_NLS_GROUP=4 timeout 900 python -m pytest tests/test_gpu_03_two_opt.py tests/test_gpu_15_full_batch.py -m gpu -q --timeout 600 -x -k "nls or config3" > $OUT/pytest_g4.log 2>&1
echo "pytest g4 rc=$?" >> $OUT/pytest_g4.log
tail -2 $OUT/pytest_g4.log | cut -c1-300
for i in 1 2; do
  timeout 400 python tools/bench_nls_fused.py 64 3 g3,g4_192,t256,g4 2>&1 | grep variant | tee -a $OUT/bench_nls_c3.txt | cut -c1-200
done
DACO_NLS_GROUP=4 timeout 200 python tools/bench_nls_fused.py 64 2 prof 2>&1 | tail -2 | tee -a $OUT/bench_nls_c3.txt | cut -c1-600
