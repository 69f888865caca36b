#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_00_tsp.py tests/test_gpu_02_cvrp.py tests/test_gpu_06_parallel.py -x -q 2>&1 | tail -8 > gpurun_out/r4z_tests.log
timeout 300 python bench.py --no-cpu --no-extras --steps 10 --warmup 2 --min-seconds 0 2>/dev/null | tail -1 > gpurun_out/r4z_bench_scan.json
timeout 300 python bench.py --sampler scan_sparse --no-cpu --no-extras --steps 10 --warmup 2 --min-seconds 0 2>/dev/null | tail -1 > gpurun_out/r4z_bench_sparse.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sp -o sp -- python $GRAFT_REPO_ROOT/bench.py --sampler scan_sparse --no-cpu --no-extras --steps 10 --warmup 2 --min-seconds 0 > /dev/null 2>&1
find /tmp/prof_sp -name '*kernel_stats.csv' -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r4z_kernel_stats.csv \;
