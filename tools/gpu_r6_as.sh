#!/bin/bash
# Round 6, GPU session AS: the train-mode forward of the HIP kernels against the reference fixtures' heu_train (what tolerance holds).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06as
mkdir -p $OUT
cd $R
timeout 300 python tools/net_train_mode_error.py 2>&1 | grep fixture | tee $OUT/net_train_mode_error.txt | cut -c1-300
