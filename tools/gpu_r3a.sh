#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
timeout 1700 python -m pytest tests -q -m gpu --deselect tests/test_gpu_08_soak.py 2>&1 | tail -15 > gpurun_out/r3d/pytest_gpu.log
cat gpurun_out/r3d/pytest_gpu.log
