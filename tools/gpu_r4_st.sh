#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_12_streams.py tests/test_gpu_06_parallel.py -x -q 2>&1 | tail -6 > gpurun_out/st_tests.log
timeout 300 python bench.py --no-cpu --no-extras --min-seconds 3 2>gpurun_out/st_bench4.err | tail -1 > gpurun_out/st_bench4.json
timeout 300 python bench.py --no-cpu --no-extras --min-seconds 3 --streams 1 2>/dev/null | tail -1 > gpurun_out/st_bench1.json
timeout 300 python bench.py --no-cpu --no-extras --min-seconds 3 --streams 8 2>/dev/null | tail -1 > gpurun_out/st_bench8.json
timeout 300 python bench.py --no-cpu --no-extras --min-seconds 3 --streams 2 2>/dev/null | tail -1 > gpurun_out/st_bench2.json
timeout 300 python bench.py --no-cpu --no-extras --min-seconds 3 --sampler scan_sparse 2>/dev/null | tail -1 > gpurun_out/st_bench4_sparse.json
