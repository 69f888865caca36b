#!/bin/bash
# Round 6, GPU session AK: the driver's command line three times (timed region against the sustained loop of the same run).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06ak
mkdir -p $OUT
cd $R
for i in 1 2 3; do
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras 2>/dev/null | tail -1 > $OUT/line_$i.json
  python -c "import json,sys; j=json.load(open('$OUT/line_$i.json')); print($i, j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['sustained']['value'])"
done
