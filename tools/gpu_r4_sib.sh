#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_05_siblings.py -q 2>&1 | tail -40 > gpurun_out/sib_tests.log
