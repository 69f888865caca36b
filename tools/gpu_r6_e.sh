#!/bin/bash
# Round 6, GPU session E: unconditional row loads, tour lengths summed in the loop, grouped table, batched head-row fetches in
# the update.  Headline fused / pre-pass, the epilogue ablation, one instance, then the suite.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06e
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_11_scan_sparse.py tests/test_gpu_15_full_batch.py -q -x --timeout 200 -k "scan_sparse or head or lds or grouped or categorical" > $OUT/pytest_first.log 2>&1
echo "first rc=$?" | tee -a $OUT/pytest_first.log; tail -4 $OUT/pytest_first.log
for i in 1 2; do
  timeout 120 python bench.py --no-cpu --no-extras --min-seconds 2 --steps 20 2>/dev/null | tail -1 > $OUT/headline_fused_$i.json
  DACO_FUSE_HEAD_ROWS=0 timeout 120 python bench.py --no-cpu --no-extras --min-seconds 2 --steps 20 2>/dev/null | tail -1 > $OUT/headline_prepass_$i.json
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r06e/headline_*.json"))):
    try:
        j=json.load(open(f)); print(os.path.basename(f), j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], (j.get("sustained") or {}).get("value"))
    except Exception as e: print(f, e)
PY
timeout 120 python tools/ablate_epilogue.py > $OUT/ablate_epilogue.json 2>$OUT/ablate_epilogue.err; cat $OUT/ablate_epilogue.json
timeout 200 python tools/b1_modes.py 300 > $OUT/b1_modes.txt 2>&1; grep '"n": 500' $OUT/b1_modes.txt
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -o p -- python bench.py --no-cpu --no-extras --min-seconds 0 > $OUT/stats_headline.log 2>&1)
cp $OUT/stats_headline/p_kernel_stats.csv $OUT/kernel_stats_headline.csv 2>/dev/null; rm -rf $OUT/stats_headline
head -5 $OUT/kernel_stats_headline.csv | cut -c1-140
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b1 -o p -- python tools/b1_modes.py 100 > $OUT/stats_b1.log 2>&1)
cp $OUT/stats_b1/p_kernel_stats.csv $OUT/kernel_stats_b1.csv 2>/dev/null; rm -rf $OUT/stats_b1
grep -E "scan_sparse_kernel<2|deposit_rows" $OUT/kernel_stats_b1.csv | cut -c1-140
cd $R
timeout 700 python -m pytest tests -m gpu -q --timeout 240 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
ls $OUT
