#!/bin/bash
# Round 6, GPU session AW: two variants of nls_kernel against the committed build on one box, alternating (DACO_LIB_PATH):
# v1 = every entry finds its own list from the bitmap (power-of-two entries per thread); v2 = the dirty test's LDS reads up front.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06aw
mkdir -p $OUT
cd $R
for v in v1 v2; do
  DACO_LIB_PATH=$R/deepaco_amd/lib/libdeepaco_hip_$v.so timeout 600 python -m pytest tests/test_gpu_03_two_opt.py tests/test_gpu_15_full_batch.py -m gpu -q --timeout 600 -x > $OUT/pytest_$v.log 2>&1
  echo "pytest $v rc=$?" | tee -a $OUT/pytest_$v.log
done
for i in 1 2; do
  for v in head v1 v2; do
    echo "== $v" | tee -a $OUT/bench_nls_c3.txt
    L=$R/deepaco_amd/lib/libdeepaco_hip_$v.so; [ $v = head ] && L=$R/deepaco_amd/lib/libdeepaco_hip.so
    DACO_LIB_PATH=$L timeout 400 python tools/bench_nls_fused.py 64 3 g4_192 2>&1 | grep variant | tee -a $OUT/bench_nls_c3.txt | cut -c1-140
  done
done
for v in head v1 v2; do
  echo "== $v, shapes" | tee -a $OUT/shapes.txt
  L=$R/deepaco_amd/lib/libdeepaco_hip_$v.so; [ $v = head ] && L=$R/deepaco_amd/lib/libdeepaco_hip.so
  DACO_LIB_PATH=$L timeout 600 python tools/ab_nls_owner_bits.py 5 DACO_NLS_OWNER_BITS 1 2>&1 | grep instances | tee -a $OUT/shapes.txt | cut -c1-200
done
