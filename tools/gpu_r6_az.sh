#!/bin/bash
# Round 6, GPU session AZ: soak of the dirty-list / fused NLS kernel after this session's changes (random sizes, matrix kinds,
# thread and group shapes) against the dense incremental kernel and the pass-by-pass driver.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06az
mkdir -p $OUT
cd $R
timeout 500 python tools/soak_two_opt.py 200 20261001 2>&1 | tail -2 | tee $OUT/soak_two_opt.txt | cut -c1-400
timeout 500 python tools/soak_two_opt.py 200 777 2>&1 | tail -2 | tee -a $OUT/soak_two_opt.txt | cut -c1-400
