#!/bin/bash
# Round 6, GPU session W: the fused GNN layer kernel compiled for 4 / 3 / 2 workgroups per CU (more registers per wavefront).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06w
mkdir -p $OUT
cd $R
for rep in 1 2; do
for lib in libdeepaco_hip.so libdeepaco_hip_occ3.so libdeepaco_hip_occ2.so; do
  echo "$lib: $(DACO_LIB_PATH=$R/deepaco_amd/lib/$lib timeout 200 python tools/time_gnn_batch.py 2>/dev/null | tail -1)" | tee -a $OUT/gnn_occupancy.txt
done
done
