#!/bin/bash
# Round 6, GPU session N: the captured training step after the memset nodes became kernels; where a slow replay spends its time.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06n
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_07_net.py tests/test_gpu_03_two_opt.py -m gpu -q --timeout 240 > $OUT/pytest_net_nls.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_net_nls.log
tail -8 $OUT/pytest_net_nls.log | cut -c1-400
timeout 300 python tools/time_train_step.py 30 > $OUT/train_default.txt 2>&1; cut -c1-900 $OUT/train_default.txt
for nt in 256 1024; do
  DACO_NLS_THREADS=$nt TRAIN_MODES=eager_flat,graph timeout 200 python tools/time_train_step.py 30 --shape 100 > $OUT/train_nt$nt.txt 2>&1; echo "NT=$nt"; cut -c1-700 $OUT/train_nt$nt.txt
done
cd /tmp && export TMPDIR=/tmp
TRAIN_MODES=graph timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_graph500 -o g500 -- python $R/tools/time_train_step.py 5 --shape 500 > $OUT/prof_graph500.log 2>&1
TRAIN_MODES=graph timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_graph100 -o g100 -- python $R/tools/time_train_step.py 20 --shape 100 > $OUT/prof_graph100.log 2>&1
cd $R
for d in prof_graph500 prof_graph100; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); echo "== $d $f"; head -12 "$f" | cut -c1-200
  # keep only the stats (the traces are large)
  find $OUT/$d -name "*kernel_trace.csv" -size +20M -delete
done
tail -3 $OUT/prof_graph500.log | cut -c1-600
