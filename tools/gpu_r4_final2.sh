#!/bin/bash
mkdir -p gpurun_out
bash tools/profile_r4.sh r04b stats > gpurun_out/final_profile_stats.log 2>&1
timeout 900 python bench.py > gpurun_out/final_bench_default.log 2>gpurun_out/final_bench_default.err
