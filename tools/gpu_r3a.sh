#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout 600 python -m pytest tests/test_gpu_02_cvrp.py -x -q -m gpu 2>&1 | tail -15
