#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4l; mkdir -p $O
python tools/run_headline_kernel.py 8 64 512 500 scan_sparse | tee $O/sparse_time.json
python tools/run_headline_kernel.py 8 64 512 500 scan | tee -a $O/sparse_time.json
python tools/run_headline_kernel.py 6 64 2048 1000 scan_sparse | tee -a $O/sparse_time.json
python tools/run_headline_kernel.py 6 64 2048 1000 scan | tee -a $O/sparse_time.json
