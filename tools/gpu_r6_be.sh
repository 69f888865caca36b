#!/bin/bash
# Round 6, GPU session BE: the backward's scatter as atomics (default below 100 k edges) against the CSR gather form at the TSP-100 training shape.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06be
mkdir -p $OUT
cd $R
for i in 1 2; do
  echo "== default" | tee -a $OUT/train_step.txt
  TRAIN_MODES=graph timeout 300 python tools/time_train_step.py 40 --shape 100 2>/dev/null | tee -a $OUT/train_step.txt | cut -c1-200
  echo "== DACO_GNN_TRAIN_GATHER=1" | tee -a $OUT/train_step.txt
  DACO_GNN_TRAIN_GATHER=1 TRAIN_MODES=graph timeout 300 python tools/time_train_step.py 40 --shape 100 2>/dev/null | tee -a $OUT/train_step.txt | cut -c1-200
done
