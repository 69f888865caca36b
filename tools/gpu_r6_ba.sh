#!/bin/bash
# Round 6, GPU session BA: counters and kernel statistics of nls_kernel at config 3 after this session's changes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile_r6.sh r06ba nls 2>&1 | tail -5
grep -A32 "nls_kernel" gpurun_out/r06ba/pmc_nls.txt | head -40
head -5 gpurun_out/r06ba/kernel_stats_nls.csv | cut -c1-200
