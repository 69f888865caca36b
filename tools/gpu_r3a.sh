#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3s
O=gpurun_out/r3s
timeout 900 python -m pytest tests/test_gpu_00_tsp.py tests/test_gpu_02_cvrp.py -x -q -m gpu > $O/tests.txt 2>&1
tail -5 $O/tests.txt
timeout 300 python tests/soak_parity.py 600 4242 > $O/soak.txt 2>&1
tail -3 $O/soak.txt
cd /tmp; export TMPDIR=/tmp
for L in 8 16; do
  DACO_SCAN_LAYOUT=$L timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_l$L -o p -- python $GRAFT_REPO_ROOT/tools/measure_configs.py c2 c4 > $GRAFT_REPO_ROOT/$O/cfg_l$L.txt 2>&1
  grep -h '^{' $GRAFT_REPO_ROOT/$O/cfg_l$L.txt
  f=$(find $GRAFT_REPO_ROOT/$O/prof_l$L -name '*kernel_stats.csv' | head -1)
  head -6 "$f" | cut -d, -f1-5
done
