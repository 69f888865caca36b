#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
timeout 1200 python -m pytest tests/test_gpu_09_cvrp_ls.py -x -q -s 2>&1 | tail -25 > gpurun_out/r3j/pytest.log
cat gpurun_out/r3j/pytest.log
