#!/bin/bash
# Round 6, GPU session AP: nls_kernel finds the list of a thread's first entry from a bitmap of list starts (no binary search):
# parity (every test that runs the NLS), A/B against the search at the callers' shapes and at config 3.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06ap
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_03_two_opt.py tests/test_gpu_15_full_batch.py tests/test_gpu_07_net.py -m gpu -q --timeout 600 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log | cut -c1-300
timeout 600 python tools/ab_nls_owner_bits.py 7 2>&1 | tee $OUT/ab_nls_owner_bits.txt | cut -c1-300
for i in 1 2; do
  timeout 300 python tools/bench_nls_fused.py 64 3 g3,g3s 2>&1 | grep variant | tee -a $OUT/bench_nls_c3.txt | cut -c1-300
done
timeout 200 python tools/bench_nls_fused.py 64 2 prof 2>&1 | tail -3 | tee -a $OUT/bench_nls_c3.txt | cut -c1-600
