#!/bin/bash
# Round 6, GPU session M: the flat parameter block / captured training step (tests + timing), the NLS with one or two wavefronts per tour.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06m
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_07_net.py tests/test_gpu_03_two_opt.py -m gpu -q --timeout 240 -x > $OUT/pytest_net_nls.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_net_nls.log
tail -15 $OUT/pytest_net_nls.log
timeout 300 python tools/time_train_step.py 30 > $OUT/train_default.txt 2>&1; cat $OUT/train_default.txt | cut -c1-700
for nt in 64 128 256 1024; do
  DACO_NLS_THREADS=$nt timeout 200 python tools/time_train_step.py 30 --shape 100 > $OUT/train_nt$nt.txt 2>&1; echo "NT=$nt"; cut -c1-700 $OUT/train_nt$nt.txt
done
