#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
O=gpurun_out/r3g
timeout 900 python -m pytest tests/test_gpu_00_tsp.py tests/test_gpu_05_siblings.py tests/test_gpu_02_cvrp.py tests/test_gpu_06_parallel.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
timeout 300 python tests/soak_parity.py 600 31 > $O/soak.txt 2>&1; tail -3 $O/soak.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats2 -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras --min-seconds 0 > $GRAFT_REPO_ROOT/$O/bench2.log 2>&1
grep -h '^{' $GRAFT_REPO_ROOT/$O/bench2.log | cut -c1-200
