#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python tools/measure_configs.py c2 c4 2>/dev/null | grep '^{' | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_00_tsp.py tests/test_gpu_02_cvrp.py -x -q -m gpu 2>&1 | tail -2
