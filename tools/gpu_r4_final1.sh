#!/bin/bash
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/final_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1
bash tools/profile_r4.sh r04b sparse > gpurun_out/final_profile_sparse.log 2>&1
