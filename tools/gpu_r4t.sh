#!/bin/bash
# occupancy / round structure of scan_sparse_kernel: workgroups = 32 * B against 6 per CU x 256 CUs = 1536 resident
mkdir -p gpurun_out
for B in 8 16 24 32 48 64 72 96 128; do
  timeout 120 python tools/run_headline_kernel.py 8 $B 512 500 scan_sparse
done > gpurun_out/r4t_sweep.jsonl 2>&1
for B in 16 48 64; do
  SPARSE_PLAIN=1 timeout 120 python tools/run_headline_kernel.py 8 $B 512 500 scan_sparse
done > gpurun_out/r4t_sweep_plain.jsonl 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sp -o sp -- python $GRAFT_REPO_ROOT/bench.py --sampler scan_sparse --no-cpu --no-extras --steps 10 --warmup 2 --min-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/r4t_bench.log 2>&1
find /tmp/prof_sp -name '*kernel_stats.csv' -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r4t_kernel_stats.csv \;
