#!/bin/bash
# Round 6, GPU session BL: hgs_ls_kernel's latency mode copies its matrix sixteen entries per lane at a time: parity, the
# reference's 8-ant call against the previous build on one box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06bl
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_13_hgs_ls.py tests/test_gpu_16_cvrp_pipeline.py -m gpu -q --timeout 600 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -2 $OUT/pytest.log | cut -c1-300
for i in 1 2 3; do
  for v in new prev; do
    L=$R/deepaco_amd/lib/libdeepaco_hip.so; [ $v = prev ] && L=$R/deepaco_amd/lib/libdeepaco_hip_prev.so
    echo "== $v" | tee -a $OUT/hgs_latency.txt
    DACO_LIB_PATH=$L timeout 300 python tools/bench_hgs_ls.py --ants 8 --batch 1 2>/dev/null | grep "^hgs" | tee -a $OUT/hgs_latency.txt | cut -c1-200
  done
done
