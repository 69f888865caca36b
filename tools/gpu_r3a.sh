#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
O=gpurun_out/r3g
timeout 900 python -m pytest tests/test_gpu_00_tsp.py tests/test_gpu_01_pick.py tests/test_gpu_05_siblings.py tests/test_gpu_02_cvrp.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras --min-seconds 0 > $GRAFT_REPO_ROOT/$O/bench.log 2>&1
grep -h '^{' $GRAFT_REPO_ROOT/$O/bench.log | cut -c1-200
head -8 $GRAFT_REPO_ROOT/$O/stats/p_kernel_stats.csv | cut -c1-60,200-300
