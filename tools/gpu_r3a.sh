#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
O=gpurun_out/r3i
timeout 900 python -m pytest tests/test_gpu_00_tsp.py tests/test_gpu_02_cvrp.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
for S in lds regs; do DACO_SCAN_SEARCH=$S timeout 300 python tools/measure_configs.py c2 c4 2>&1 | grep '^{' | cut -c1-200; done
