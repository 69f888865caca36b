#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_09_cvrp_ls.py -x -q 2>&1 | tail -2
timeout 300 python tools/measure_cvrp_ls.py 2>/dev/null | cut -c1-330
