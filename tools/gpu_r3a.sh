#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3z
O=gpurun_out/r3z
timeout 900 python -m pytest tests/test_gpu_02_cvrp.py -x -q -m gpu > $O/tests02.txt 2>&1
tail -2 $O/tests02.txt
timeout 300 python tools/measure_configs.py c4 2>&1 | grep '^{' | cut -c1-200
timeout 300 python tools/measure_configs.py c4 2>&1 | grep '^{' | cut -c1-200
