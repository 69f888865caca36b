#!/bin/bash
# Round 6, GPU session AV: nls_kernel with every step of an evaluation round as a loop of its own (the entries' LDS round trips overlap) and each entry finding its own list:
# waited for (no rank read under a branch): parity, then against the previous build (DACO_LIB_PATH) on one box, alternating.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06av
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_03_two_opt.py tests/test_gpu_15_full_batch.py tests/test_gpu_07_net.py tests/test_gpu_00_tsp.py -m gpu -q --timeout 600 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -2 $OUT/pytest.log | cut -c1-300
DACO_NLS_GROUP=3 timeout 900 python -m pytest tests/test_gpu_03_two_opt.py -m gpu -q --timeout 600 -x -k "nls" > $OUT/pytest_g3.log 2>&1
echo "pytest g3 rc=$?" >> $OUT/pytest_g3.log
tail -2 $OUT/pytest_g3.log | cut -c1-300
for i in 1 2; do
  echo "== new" | tee -a $OUT/bench_nls_c3.txt
  timeout 400 python tools/bench_nls_fused.py 64 3 g4_192,g3 2>&1 | grep variant | tee -a $OUT/bench_nls_c3.txt | cut -c1-200
  echo "== previous build" | tee -a $OUT/bench_nls_c3.txt
  DACO_LIB_PATH=$R/deepaco_amd/lib/libdeepaco_hip_prev.so timeout 400 python tools/bench_nls_fused.py 64 3 g4_192,g3 2>&1 | grep variant | tee -a $OUT/bench_nls_c3.txt | cut -c1-200
done
echo "== new, shapes" | tee $OUT/shapes.txt
timeout 600 python tools/ab_nls_owner_bits.py 5 DACO_NLS_OWNER_BITS 1 2>&1 | grep instances | tee -a $OUT/shapes.txt | cut -c1-200
echo "== previous build, shapes" | tee -a $OUT/shapes.txt
DACO_LIB_PATH=$R/deepaco_amd/lib/libdeepaco_hip_prev.so timeout 600 python tools/ab_nls_owner_bits.py 5 DACO_NLS_OWNER_BITS 1 2>&1 | grep instances | tee -a $OUT/shapes.txt | cut -c1-200
