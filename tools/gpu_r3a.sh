#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
O=gpurun_out/r3h
R=$GRAFT_REPO_ROOT
(cd $R && python bench.py > $O/bench_default.json 2> $O/bench_default.log)
cd /tmp; export TMPDIR=/tmp
(cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o p -- python bench.py --no-cpu --min-seconds 0 > $R/$O/bench_profiled.log 2>&1)
cp $R/$O/stats/p_kernel_stats.csv $R/$O/kernel_stats_bench_default.csv
grep -h '^{' $R/$O/bench_default.json | cut -c1-260
