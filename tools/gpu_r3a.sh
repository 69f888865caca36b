#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
for v in none COSTS; do
  echo "== ablate $v"
  if [ $v = none ]; then timeout 300 python bench.py --no-cpu --no-extras --steps 20 --min-seconds 0.5 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['frac'])"
  else env DACO_ABLATE_$v=1 timeout 300 python bench.py --no-cpu --no-extras --steps 20 --min-seconds 0.5 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['frac'])"
  fi
done > gpurun_out/r3c/ablate.log 2>&1
timeout 600 python -m pytest tests/test_gpu_00_tsp.py -x -q 2>&1 | tail -3 >> gpurun_out/r3c/ablate.log
cat gpurun_out/r3c/ablate.log
