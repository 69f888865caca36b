#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout 600 python -m pytest tests/test_gpu_07_net.py -x -q -k "fused or batched_forward or eval_hip" 2>&1 | tail -3 > gpurun_out/r3k/log.txt
timeout 300 python tools/measure_configs.py gnn 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read())
for s in j['sizes']: print(s['n'], 'batch64 ms', round(s['hip_batch64_ms'],3))" >> gpurun_out/r3k/log.txt
cat gpurun_out/r3k/log.txt
