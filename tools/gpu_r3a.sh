#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3t
O=gpurun_out/r3t
timeout 900 python -m pytest tests/test_gpu_00_tsp.py tests/test_gpu_02_cvrp.py -x -q -m gpu > $O/tests.txt 2>&1
tail -4 $O/tests.txt
for L in 8 16; do DACO_SCAN_LAYOUT=$L timeout 200 python tools/time_layouts.py 200 2>&1 | tail -1; done
for L in 4 8 16; do DACO_SCAN_LAYOUT=$L timeout 200 python tools/time_layouts.py 50 512 256 2>&1 | tail -1; done
for L in 4 8 16; do DACO_SCAN_LAYOUT=$L timeout 200 python tools/time_layouts.py 128 512 256 2>&1 | tail -1; done
for L in 4 8 16; do DACO_SCAN_LAYOUT=$L timeout 200 python tools/time_layouts.py 20 512 512 2>&1 | tail -1; done
timeout 300 python tests/soak_parity.py 800 777 > $O/soak.txt 2>&1
tail -3 $O/soak.txt
