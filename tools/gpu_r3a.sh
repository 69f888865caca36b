#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3o
timeout 600 python -m pytest tests/test_gpu_00_tsp.py tests/test_gpu_02_cvrp.py -x -q -m gpu 2>&1 | tail -1
timeout 200 python tools/measure_configs.py c2 c4 2>&1 | grep '^{' | cut -c1-200
timeout 200 python bench.py --no-cpu --no-extras --min-seconds 0 2>/dev/null | grep '^{' | cut -c1-220
