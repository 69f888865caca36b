#!/bin/bash
# Round 6, GPU session BK: the LDS-heads variant's table copy eight rows at a time: parity of the few-ants tests, one instance
# (TSP-500 x 512 and x 50 ants) against the previous build on one box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06bk
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_11_scan_sparse.py tests/test_gpu_00_tsp.py -m gpu -q --timeout 600 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -2 $OUT/pytest.log | cut -c1-300
for i in 1 2 3; do
  for v in new prev; do
    L=$R/deepaco_amd/lib/libdeepaco_hip.so; [ $v = prev ] && L=$R/deepaco_amd/lib/libdeepaco_hip_prev.so
    echo "== $v" | tee -a $OUT/b1_modes.txt
    DACO_LIB_PATH=$L timeout 300 python tools/b1_modes.py 200 2>/dev/null | grep '"n": 500' | tee -a $OUT/b1_modes.txt | cut -c1-200
  done
done
