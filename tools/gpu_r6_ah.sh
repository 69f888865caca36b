#!/bin/bash
# Round 6, GPU session AH: less host time per library call (device context only when needed, loop invariants, the loop's own flag words).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06ah
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_00_tsp.py tests/test_gpu_02_cvrp.py tests/test_gpu_14_surface.py tests/test_gpu_16_cvrp_pipeline.py tests/test_gpu_05_siblings.py tests/test_gpu_06_parallel.py -m gpu -q --timeout 300 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log | cut -c1-300
timeout 300 python tools/host_overhead_small.py 2>/dev/null | tee $OUT/host_overhead.txt
