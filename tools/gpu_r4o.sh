#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_06_parallel.py tests/test_gpu_11_scan_sparse.py -x -q -m gpu 2>&1 | tail -8 > $O/pytest.log; cat $O/pytest.log
for smp in scan scan_wave; do
  timeout 200 python bench.py --no-cpu --no-extras --min-seconds 0 --batch 1 --steps 50 --warmup 5 --sampler $smp 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$smp B=1', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
