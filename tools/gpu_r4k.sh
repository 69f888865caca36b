#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_11_scan_sparse.py -x -q -m gpu 2>&1 | tail -5 > $O/pytest.log; cat $O/pytest.log
python tools/run_headline_kernel.py 8 64 512 500 scan_sparse | tee $O/sparse_time.json
python tools/run_headline_kernel.py 8 48 512 500 scan_sparse | tee -a $O/sparse_time.json
