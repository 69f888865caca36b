#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_11_scan_sparse.py -x -q 2>&1 | tail -15 > gpurun_out/r4x_tests.log
for B in 8 16 32 48 64 96 128; do
  timeout 120 python tools/run_headline_kernel.py 8 $B 512 500 scan_sparse 2>/dev/null
done > gpurun_out/r4x_sweep.jsonl 2>&1
SPARSE_PLAIN=1 timeout 120 python tools/run_headline_kernel.py 8 64 512 500 scan_sparse 2>/dev/null >> gpurun_out/r4x_sweep.jsonl
timeout 120 python tools/run_headline_kernel.py 8 64 512 500 race_head 2>/dev/null >> gpurun_out/r4x_sweep.jsonl
timeout 300 python bench.py --sampler scan_sparse --no-cpu --no-extras --steps 10 --warmup 2 --min-seconds 0 2>/dev/null | tail -1 > gpurun_out/r4x_bench.json
