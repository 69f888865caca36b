#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_11_scan_sparse.py -x -q -m gpu 2>&1 | tail -25 > $O/pytest.log; cat $O/pytest.log
