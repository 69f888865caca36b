#!/bin/bash
# Round 6, GPU session BG: kernel statistics of the single-instance inference pipeline (one TSP-500 instance, 50 ants: the
# reference's own call pattern) -- which small kernels take more than their work.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06bg
mkdir -p $OUT
cd $R
timeout 200 python tools/time_infer_pipeline.py 1 50 2>&1 | tail -6 | tee $OUT/pipeline_b1.txt | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_i -o g --output-format csv -- python $R/tools/time_infer_pipeline.py 1 50 > $OUT/prof.log 2>&1
f=$(find /tmp/prof_i -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats_infer_b1.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats_infer_b1.csv")))
for r in rows[:25]:
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:8.1f} us", f"{float(r['Percentage']):5.1f}%")
PY
