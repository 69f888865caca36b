#!/bin/bash
# Round 6, GPU session D: head-row kernels without exponents (no calls in the kernels), the suite, headline fused / pre-pass,
# one instance, kernel stats.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06d
mkdir -p $OUT
cd $R
for i in 1 2; do
  timeout 120 python bench.py --no-cpu --no-extras --min-seconds 2 --steps 20 2>/dev/null | tail -1 > $OUT/headline_fused_$i.json
  DACO_FUSE_HEAD_ROWS=0 timeout 120 python bench.py --no-cpu --no-extras --min-seconds 2 --steps 20 2>/dev/null | tail -1 > $OUT/headline_prepass_$i.json
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r06d/headline_*.json"))):
    try:
        j=json.load(open(f)); print(os.path.basename(f), j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], (j.get("sustained") or {}).get("value"))
    except Exception as e: print(f, e)
PY
timeout 200 python tools/b1_modes.py 300 > $OUT/b1_modes.txt 2>&1; grep '"n": 500' $OUT/b1_modes.txt
timeout 900 python -m pytest tests -m gpu -q --timeout 240 -x > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -o p -- python bench.py --no-cpu --no-extras --min-seconds 0 > $OUT/stats_headline.log 2>&1)
cp $OUT/stats_headline/p_kernel_stats.csv $OUT/kernel_stats_headline.csv 2>/dev/null; rm -rf $OUT/stats_headline
head -6 $OUT/kernel_stats_headline.csv | cut -c1-140
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b1 -o p -- python tools/b1_modes.py 100 > $OUT/stats_b1.log 2>&1)
cp $OUT/stats_b1/p_kernel_stats.csv $OUT/kernel_stats_b1.csv 2>/dev/null; rm -rf $OUT/stats_b1
head -8 $OUT/kernel_stats_b1.csv | cut -c1-140
ls $OUT
