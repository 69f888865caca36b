#!/bin/bash
# Round 6, GPU session Y: the inference pipeline stage by stage at the batch sizes a caller forms (64 x 512 ants, 1 x 50, 16 x 20).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06y
mkdir -p $OUT
cd $R
for cfg in "64 512" "1 50" "16 20" "1 512"; do
  timeout 200 python tools/time_infer_pipeline.py $cfg 2>/dev/null | tee -a $OUT/infer_pipeline.txt
done
