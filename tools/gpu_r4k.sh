#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_11_scan_sparse.py tests/test_gpu_00_tsp.py -x -q -m gpu 2>&1 | tail -5 > $O/pytest.log; cat $O/pytest.log
timeout 300 python bench.py --no-cpu --no-extras --min-seconds 0 --sampler race 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('race colony', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
