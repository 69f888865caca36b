#!/bin/bash
# Round 6, GPU session AC: the latency mode's applied moves on register mirrors: parity (both forms), soak, the 8-ant call.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06ac
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_13_hgs_ls.py tests/test_gpu_16_cvrp_pipeline.py -m gpu -q --timeout 300 -x > $OUT/pytest_hgs.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_hgs.log
tail -6 $OUT/pytest_hgs.log | cut -c1-300
DACO_HGS_LATENCY=1 timeout 200 python tools/soak_hgs_ls.py 60 21 > $OUT/soak_hgs_latency.txt 2>&1; tail -1 $OUT/soak_hgs_latency.txt | cut -c1-300
for m in 0 1; do
  echo "== DACO_HGS_LATENCY=$m, 8 ants x 1 instance" | tee -a $OUT/hgs_latency.txt
  DACO_HGS_LATENCY=$m timeout 200 python tools/bench_hgs_ls.py --ants 8 --batch 1 --reps 5 2>&1 | grep -v amdgpu.ids | tee -a $OUT/hgs_latency.txt
done
for a in 16 64; do
  for m in 0 1; do
    echo "== DACO_HGS_LATENCY=$m, $a ants x 1 instance" | tee -a $OUT/hgs_latency.txt
    DACO_HGS_LATENCY=$m timeout 200 python tools/bench_hgs_ls.py --ants $a --batch 1 --reps 3 --no-short 2>&1 | grep "^hgs" | tail -1 | tee -a $OUT/hgs_latency.txt
  done
done
