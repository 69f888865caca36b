#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
timeout 900 python -m pytest tests/test_gpu_03_two_opt.py -x -q 2>&1 | tail -5 > gpurun_out/r3g/pytest_03.log
timeout 900 python tools/bench_nls_fused.py 64 4 prof,g2,g4,g1,g2,g4 > gpurun_out/r3g/bench_nls.log 2>&1
cat gpurun_out/r3g/pytest_03.log gpurun_out/r3g/bench_nls.log
