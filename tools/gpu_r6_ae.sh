#!/bin/bash
# Round 6, GPU session AE: the fused NLS kernel's per-phase cycle shares (DACO_NLS_PROFILE=1) at the latency-bound shapes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06ae
mkdir -p $OUT
cd $R
DACO_NLS_PROFILE=1 timeout 300 python tools/sweep_nls_threads.py 1 2>&1 | grep -E "nls profile|^\{" | awk 'NR<=60' | cut -c1-330 | tee $OUT/nls_profile.txt
