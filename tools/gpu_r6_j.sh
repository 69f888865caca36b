#!/bin/bash
# Round 6, GPU session J: the driver's command line (--steps 20 --warmup 5) with one and two colonies (HIP streams) per GPU.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06j
mkdir -p $OUT
cd $R
for i in 1 2 3; do
  for st in 1 2 3; do
    timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras --min-seconds 3 --streams $st 2>/dev/null | tail -1 > $OUT/streams${st}_$i.json
  done
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r06j/streams*.json"))):
    try:
        j=json.load(open(f)); print(os.path.basename(f), j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], (j.get("sustained") or {}).get("value"))
    except Exception as e: print(f, e)
PY
