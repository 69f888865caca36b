#!/bin/bash
# Round 6, GPU session Z: sampler 'auto' picks an LDS-compatible head on a learned heuristic: tests, the inference pipeline again.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06z
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_11_scan_sparse.py tests/test_gpu_07_net.py tests/test_gpu_00_tsp.py tests/test_gpu_14_surface.py -m gpu -q --timeout 300 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log | cut -c1-300
for cfg in "1 50" "1 512" "16 20" "64 512"; do
  timeout 200 python tools/time_infer_pipeline.py $cfg 2>/dev/null | tee -a $OUT/infer_pipeline.txt
done
