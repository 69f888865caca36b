#!/bin/bash
# Round 6, GPU session AG: host share of an ACO iteration at the reference's small sizes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06ag
mkdir -p $OUT
cd $R
timeout 300 python tools/host_overhead_small.py 2>/dev/null | tee $OUT/host_overhead.txt
