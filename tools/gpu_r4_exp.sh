#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_11_scan_sparse.py -x -q 2>&1 | tail -6 > gpurun_out/exp_tests.log
timeout 600 python tools/soak_scan_sparse.py 45 1 > gpurun_out/exp_soak.jsonl 2>gpurun_out/exp_soak.err
for B in 48 64; do timeout 120 python tools/run_headline_kernel.py 8 $B 512 500 scan_sparse 2>/dev/null; done > gpurun_out/exp_sweep.jsonl
timeout 300 python tools/run_headline_kernel.py 6 64 2048 1000 scan_sparse 2>/dev/null >> gpurun_out/exp_sweep.jsonl
timeout 900 python bench.py --no-cpu --min-seconds 0 2>/dev/null | tail -1 > gpurun_out/exp_bench.json
