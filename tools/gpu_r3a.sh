#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3o
mkdir -p $O
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python bench.py --no-cpu --no-extras --min-seconds 0 > $O/bench_profiled.log 2>&1)
cp $O/stats/p_kernel_stats.csv $O/kernel_stats_bench_default.csv
find $O -name "*.db" -delete
grep "^{" $O/bench_profiled.log | cut -c1-600
head -5 $O/kernel_stats_bench_default.csv
