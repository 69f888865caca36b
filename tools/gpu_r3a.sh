#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3z
O=gpurun_out/r3z
timeout 600 python -m pytest tests/test_gpu_03_two_opt.py -x -q -m gpu > $O/tests3.txt 2>&1
tail -2 $O/tests3.txt
timeout 300 python tools/soak_two_opt.py 100 777 > $O/soak_two_opt.txt 2>&1
tail -1 $O/soak_two_opt.txt | cut -c1-500
timeout 300 python tools/bench_nls_fused.py 64 3 g3 > $O/nls3.txt 2>&1
grep variant $O/nls3.txt
