#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_09_cvrp_ls.py -q -m gpu -s -k "many_instances or reaches" 2>&1 | grep -E "n = |passed|failed|FAILED" > $O/pytest.log; cat $O/pytest.log
