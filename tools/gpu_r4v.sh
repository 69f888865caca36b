#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_11_scan_sparse.py -x -q 2>&1 | tail -15 > gpurun_out/r4v_tests.log
for LS in 24 32; do
for B in 8 48 64 128; do
  DACO_SPARSE_LS=$LS timeout 120 python tools/run_headline_kernel.py 8 $B 512 500 scan_sparse 2>/dev/null | sed "s/^{/{\"ls\": $LS, /"
done; done > gpurun_out/r4v_sweep.jsonl 2>&1
timeout 120 python tools/run_headline_kernel.py 8 64 512 500 race_head 2>/dev/null >> gpurun_out/r4v_sweep.jsonl
timeout 120 python tools/run_headline_kernel.py 8 64 2048 1000 scan_sparse 2>/dev/null >> gpurun_out/r4v_sweep.jsonl
timeout 300 python bench.py --sampler scan_sparse --no-cpu --no-extras --steps 10 --warmup 2 --min-seconds 0 2>/dev/null | tail -1 > gpurun_out/r4v_bench.json
