#!/bin/bash
# Round 6, GPU session AF: the NLS kernel's owner table for launches of few tours: parity, timing with and without.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06af
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_03_two_opt.py tests/test_gpu_15_full_batch.py -m gpu -q --timeout 300 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log | cut -c1-300
echo "# owner table (default)" | tee $OUT/sweep_nls_owner.txt
timeout 300 python tools/sweep_nls_threads.py 5 2>/dev/null | head -5 | cut -c1-260 | tee -a $OUT/sweep_nls_owner.txt
echo "# DACO_NLS_OWNER=0 (binary search)" | tee -a $OUT/sweep_nls_owner.txt
DACO_NLS_OWNER=0 timeout 300 python tools/sweep_nls_threads.py 5 2>/dev/null | head -5 | cut -c1-260 | tee -a $OUT/sweep_nls_owner.txt
for o in 1 0; do echo "DACO_NLS_OWNER=$o"; DACO_NLS_OWNER=$o TRAIN_MODES=graph timeout 200 python tools/time_train_step.py 40 2>/dev/null | cut -c1-330 | tee -a $OUT/train_owner$o.txt; done
