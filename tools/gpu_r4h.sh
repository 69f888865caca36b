#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export DACO_GNN_INPLACE=1
(cd $R && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o p -- python tools/run_gnn_batch.py 500 50 64 6 > $R/$O/stats.log 2>&1)
python - <<PY
import csv,glob
for f in glob.glob("$R/$O/stats/**/p_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'daco' in r["Name"]: print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
