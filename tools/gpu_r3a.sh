#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3n
O=gpurun_out/r3n
timeout 900 python -m pytest tests/test_gpu_00_tsp.py tests/test_gpu_02_cvrp.py tests/test_gpu_05_siblings.py tests/test_gpu_06_parallel.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
timeout 300 python tests/soak_parity.py 800 77 > $O/soak.txt 2>&1; tail -3 $O/soak.txt
timeout 200 python tools/measure_configs.py c2 c4 2>&1 | grep '^{' | cut -c1-200
timeout 200 python bench.py --no-cpu --no-extras --min-seconds 0 2>/dev/null | grep '^{' | cut -c1-220
