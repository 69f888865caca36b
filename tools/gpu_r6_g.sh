#!/bin/bash
# Round 6, GPU session G (F again after the dense rows came back): A/B on ONE box -- round 5's tree (.ab_r5, library version 124) against this tree: the headline kernel
# alone (all outputs / plain), then the bench line of both; then this tree's fused / pre-pass, one instance.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06g
mkdir -p $OUT
cd $R
for i in 1 2 3; do
  (cd $R/.ab_r5 && timeout 100 python tools/run_headline_kernel.py 14 64 512 500 scan_sparse 2>/dev/null | tail -1) >> $OUT/ab_kernel_r5.txt
  (cd $R && timeout 100 python tools/run_headline_kernel.py 14 64 512 500 scan_sparse 2>/dev/null | tail -1) >> $OUT/ab_kernel_new.txt
  (cd $R/.ab_r5 && SPARSE_PLAIN=1 timeout 100 python tools/run_headline_kernel.py 14 64 512 500 scan_sparse 2>/dev/null | tail -1) >> $OUT/ab_plain_r5.txt
  (cd $R && SPARSE_PLAIN=1 timeout 100 python tools/run_headline_kernel.py 14 64 512 500 scan_sparse 2>/dev/null | tail -1) >> $OUT/ab_plain_new.txt
done
for f in ab_kernel_r5 ab_kernel_new ab_plain_r5 ab_plain_new; do echo $f; cut -c1-120 $OUT/$f.txt; done
for i in 1 2; do
  (cd $R/.ab_r5 && timeout 120 python bench.py --no-cpu --no-extras --min-seconds 2 --steps 20 2>/dev/null | tail -1) > $OUT/bench_r5_$i.json
  timeout 120 python bench.py --no-cpu --no-extras --min-seconds 2 --steps 20 2>/dev/null | tail -1 > $OUT/bench_fused_$i.json
  DACO_FUSE_HEAD_ROWS=0 timeout 120 python bench.py --no-cpu --no-extras --min-seconds 2 --steps 20 2>/dev/null | tail -1 > $OUT/bench_prepass_$i.json
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r06g/bench_*.json"))):
    try:
        j=json.load(open(f)); print(os.path.basename(f), j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], (j.get("sustained") or {}).get("value"))
    except Exception as e: print(f, e)
PY
timeout 120 python tools/ablate_epilogue.py > $OUT/ablate_epilogue.json 2>$OUT/ablate_epilogue.err; cat $OUT/ablate_epilogue.json
timeout 200 python tools/b1_modes.py 300 > $OUT/b1_modes.txt 2>&1; grep '"n": 500' $OUT/b1_modes.txt
(cd $R/.ab_r5 && timeout 200 python tools/b1_modes.py 300 2>&1 | grep '"n": 500') > $OUT/b1_modes_r5.txt; cat $OUT/b1_modes_r5.txt
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -o p -- python bench.py --no-cpu --no-extras --min-seconds 0 > $OUT/stats_headline.log 2>&1)
cp $OUT/stats_headline/p_kernel_stats.csv $OUT/kernel_stats_headline.csv 2>/dev/null; rm -rf $OUT/stats_headline
head -5 $OUT/kernel_stats_headline.csv | cut -c1-140
cd $R
timeout 500 python -m pytest tests/test_gpu_11_scan_sparse.py tests/test_gpu_15_full_batch.py tests/test_gpu_00_tsp.py -q --timeout 200 > $OUT/pytest_11.log 2>&1; tail -3 $OUT/pytest_11.log
ls $OUT
