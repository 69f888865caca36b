#!/bin/bash
# Round 6, GPU session BJ: the construction kernel's tour lengths with four chunks' gathers in flight, the LDS-heads variant's table
# copy eight rows at a time: parity, the headline and the one-instance iteration against the previous build on one box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06bj
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_00_tsp.py tests/test_gpu_11_scan_sparse.py tests/test_gpu_15_full_batch.py tests/test_gpu_12_streams.py -m gpu -q --timeout 600 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log | cut -c1-300
for i in 1 2 3; do
  for v in new prev; do
    L=$R/deepaco_amd/lib/libdeepaco_hip.so; [ $v = prev ] && L=$R/deepaco_amd/lib/libdeepaco_hip_prev.so
    DACO_LIB_PATH=$L timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" | tee -a $OUT/ab_bench.txt
  done
done
for i in 1 2; do
  for v in new prev; do
    L=$R/deepaco_amd/lib/libdeepaco_hip.so; [ $v = prev ] && L=$R/deepaco_amd/lib/libdeepaco_hip_prev.so
    echo "== $v" | tee -a $OUT/b1_modes.txt
    DACO_LIB_PATH=$L timeout 300 python tools/b1_modes.py 200 2>/dev/null | tail -4 | tee -a $OUT/b1_modes.txt | cut -c1-300
  done
done
