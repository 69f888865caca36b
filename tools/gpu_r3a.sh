#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests/test_gpu_03_two_opt.py -x -q 2>&1 | tail -15 > gpurun_out/r3b/pytest_03.log
timeout 900 python tools/bench_nls_fused.py 64 4 prof,g2,g1,g4,g2,g1,g4,t512 > gpurun_out/r3b/bench_nls.log 2>&1
cat gpurun_out/r3b/pytest_03.log gpurun_out/r3b/bench_nls.log 
