#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
timeout 900 python bench.py --min-seconds 3 > gpurun_out/r3e/bench.json 2> gpurun_out/r3e/bench.err
tail -5 gpurun_out/r3e/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3e/bench.json').read().strip().splitlines()[-1])
def show(d,ind=0):
    for k,v in d.items():
        if isinstance(v,dict): print(' '*ind+k+':'); show(v,ind+2)
        else: print(' '*ind+f'{k}: {v if not isinstance(v,str) else v[:110]}')
show(j)
PY
