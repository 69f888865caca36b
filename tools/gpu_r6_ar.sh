#!/bin/bash
# Round 6, GPU session AR: nls_kernel with four entries per thread and round as the default: parity, the callers' shapes 4 against 3
# (2 where the launch had two), config 3 in the three forms on one box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06ar
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_03_two_opt.py tests/test_gpu_15_full_batch.py tests/test_gpu_07_net.py tests/test_gpu_00_tsp.py -m gpu -q --timeout 600 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -2 $OUT/pytest.log | cut -c1-300
timeout 600 python tools/ab_nls_owner_bits.py 7 DACO_NLS_GROUP 4,3 2>&1 | grep instances | tee $OUT/ab_nls_group.txt | cut -c1-300
for i in 1 2; do
  timeout 400 python tools/bench_nls_fused.py 64 3 g4_192,g3,g3s 2>&1 | grep variant | tee -a $OUT/bench_nls_c3.txt | cut -c1-200
done
