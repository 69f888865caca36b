#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_00_tsp.py tests/test_gpu_02_cvrp.py tests/test_gpu_10_layout_knob.py tests/test_gpu_09_cvrp_ls.py -x -q -m gpu 2>&1 | tail -3
