#!/bin/bash
# Round 6, GPU session AY: the NLS tests with the 192 x 4 shape among the thread shapes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06ay
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_03_two_opt.py -m gpu -q --timeout 600 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log | cut -c1-300
