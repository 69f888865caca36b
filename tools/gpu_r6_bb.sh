#!/bin/bash
# Round 6, GPU session BB: soaks on the final tree -- the head / tail row sampler against its restatement, the CVRP local searches.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06bb
mkdir -p $OUT
cd $R
timeout 400 python tools/soak_scan_sparse.py 150 20261001 2>&1 | tail -1 | tee $OUT/soak_scan_sparse.txt | cut -c1-400
timeout 400 python tools/soak_hgs_ls.py 120 20261001 2>&1 | tail -1 | tee $OUT/soak_hgs_ls.txt | cut -c1-400
