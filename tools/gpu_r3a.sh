#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
timeout 1700 python -m pytest tests -q -m gpu -x tests/test_gpu_00_tsp.py tests/test_gpu_06_parallel.py tests/test_gpu_07_net.py 2>&1 | tail -15 > gpurun_out/r3h/pytest.log
cat gpurun_out/r3h/pytest.log
