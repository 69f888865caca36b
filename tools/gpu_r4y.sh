#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_11_scan_sparse.py -x -q 2>&1 | tail -15 > gpurun_out/r4y_tests.log
for B in 8 48 64 128; do
  timeout 120 python tools/run_headline_kernel.py 8 $B 512 500 scan_sparse 2>/dev/null
done > gpurun_out/r4y_sweep.jsonl 2>&1
timeout 120 python tools/run_headline_kernel.py 8 64 512 500 race_head 2>/dev/null >> gpurun_out/r4y_sweep.jsonl
timeout 300 python tools/run_headline_kernel.py 6 64 2048 1000 scan_sparse 2>/dev/null >> gpurun_out/r4y_sweep.jsonl
timeout 300 python tools/run_headline_kernel.py 6 64 2048 1000 scan 2>/dev/null >> gpurun_out/r4y_sweep.jsonl
timeout 300 python tools/run_headline_kernel.py 4 64 2048 1000 race_head 2>/dev/null >> gpurun_out/r4y_sweep.jsonl
timeout 300 python bench.py --sampler scan_sparse --no-cpu --no-extras --steps 10 --warmup 2 --min-seconds 0 2>/dev/null | tail -1 > gpurun_out/r4y_bench.json
