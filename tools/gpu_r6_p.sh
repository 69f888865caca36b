#!/bin/bash
# Round 6, GPU session P: threads per tour of the fused NLS kernel at the tour counts its callers form.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06p
mkdir -p $OUT
cd $R
timeout 600 python tools/sweep_nls_threads.py 5 > $OUT/sweep_nls_threads.txt 2>&1; cat $OUT/sweep_nls_threads.txt | cut -c1-400
