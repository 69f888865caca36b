#!/bin/bash
# Round 6, GPU session AD: the network's forward for ONE graph: the per-layer kernel (default below 200 k edges) against the fused
# layer kernel forced (DACO_GNN_SPLIT_MIN_EDGES=1) over its nodes per wave.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06ad
mkdir -p $OUT
cd $R
echo "# default kernel selection" | tee $OUT/gnn_single.txt
timeout 300 python tools/time_gnn_single.py 2>/dev/null | tee -a $OUT/gnn_single.txt
echo "# DACO_GNN_SPLIT_MIN_EDGES=1 (fused layer kernel at every size)" | tee -a $OUT/gnn_single.txt
DACO_GNN_SPLIT_MIN_EDGES=1 timeout 300 python tools/time_gnn_single.py 2>/dev/null | tee -a $OUT/gnn_single.txt
