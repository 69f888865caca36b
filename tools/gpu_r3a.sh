#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3v
O=gpurun_out/r3v
timeout 900 python -m pytest tests/test_gpu_02_cvrp.py tests/test_gpu_09_cvrp_ls.py -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout 300 python tools/measure_configs.py c4 2>&1 | grep '^{'
timeout 300 python tools/measure_configs.py c4 2>&1 | grep '^{'
timeout 300 python tests/soak_parity.py 600 99 > $O/soak.txt 2>&1
tail -3 $O/soak.txt
