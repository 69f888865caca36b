#!/bin/bash
# Round 6, GPU session BP: counters of the update with the head rows after the chunk-loader change.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile_r6.sh r06bp deposit 2>&1 | tail -3
awk '/^## .*deposit_rows_kernel<true, 1, true>/{f=1} f{print} /parked on waitcnt/{if(f)exit}' gpurun_out/r06bp/pmc_deposit_heads.txt | head -40
