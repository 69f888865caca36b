#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3y
O=gpurun_out/r3y
timeout 600 python -m pytest tests/test_gpu_02_cvrp.py -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt
for n in 500 300; do
for E in 0 1; do DACO_CVRP_SCAN32=$E timeout 200 python tools/time_layouts.py $n 256 32 2>&1 | tail -1; done
done
timeout 200 python tests/soak_parity.py 400 5 > $O/soak.txt 2>&1; tail -3 $O/soak.txt
