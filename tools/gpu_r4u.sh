#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_11_scan_sparse.py -x -q 2>&1 | tail -3 > gpurun_out/r4u_tests.log
for E in 0 1 2 3 4 5 7 8 10 15; do
  for B in 8 64; do
    DACO_SPARSE_EXP=$E SPARSE_PLAIN=1 timeout 120 python tools/run_headline_kernel.py 8 $B 512 500 scan_sparse 2>/dev/null | sed "s/^{/{\"exp\": $E, /"
  done
done > gpurun_out/r4u_exp.jsonl 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sp -o sp -- python $GRAFT_REPO_ROOT/bench.py --sampler scan_sparse --no-cpu --no-extras --steps 10 --warmup 2 --min-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/r4u_bench.log 2>&1
find /tmp/prof_sp -name '*kernel_stats.csv' -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r4u_kernel_stats.csv \;
