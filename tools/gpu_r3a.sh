#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3x
O=gpurun_out/r3x
timeout 300 python -m pytest tests/test_gpu_09_cvrp_ls.py -x -q -m gpu -s > $O/tests.txt 2>&1
grep -E "mean cost|passed|failed|Error" $O/tests.txt | tail -8
timeout 200 python tools/soak_cvrp_ls.py 100 31337 > $O/soak_cvrp_ls.txt 2>&1
tail -2 $O/soak_cvrp_ls.txt | cut -c1-400
timeout 200 python tools/measure_cvrp_ls.py > $O/cvrp_ls.json 2>&1
tail -1 $O/cvrp_ls.json | cut -c1-600
