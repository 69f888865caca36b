#!/bin/bash
# Round 6, GPU session AB: where a solution's time goes in the latency mode (measurement build: cycles per phase in the stats).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06ab
mkdir -p $OUT
cd $R
DACO_LIB_PATH=$R/deepaco_amd/lib/libdeepaco_hip_hgsprof.so timeout 200 python tools/bench_hgs_ls.py --ants 8 --batch 1 --reps 3 --no-short 2>&1 | grep -v amdgpu | tee $OUT/hgs_profile.txt
