#!/bin/bash
# Round 6, GPU session B: the suite on the library with the fused head rows (no dense P, deposit emits the next iteration's rows),
# the headline with and without the fusion, kernel stats of the headline, the epilogue ablation.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06b
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
for i in 1 2; do
  python bench.py --no-cpu --no-extras --min-seconds 3 --steps 20 2>/dev/null | tail -1 > $OUT/headline_fused_$i.json
  DACO_FUSE_HEAD_ROWS=0 python bench.py --no-cpu --no-extras --min-seconds 3 --steps 20 2>/dev/null | tail -1 > $OUT/headline_prepass_$i.json
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r06b/headline_*.json"))):
    try:
        j=json.load(open(f)); print(os.path.basename(f), j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], (j.get("sustained") or {}).get("value"))
    except Exception as e: print(f, e)
PY
python tools/ablate_epilogue.py > $OUT/ablate_epilogue.json 2>$OUT/ablate_epilogue.err; cat $OUT/ablate_epilogue.json
python tools/b1_modes.py 300 > $OUT/b1_modes.txt 2>&1; cat $OUT/b1_modes.txt | tail -8
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -o p -- python bench.py --no-cpu --no-extras --min-seconds 0 > $OUT/stats_headline.log 2>&1)
cp $OUT/stats_headline/p_kernel_stats.csv $OUT/kernel_stats_headline.csv 2>/dev/null; rm -rf $OUT/stats_headline
head -8 $OUT/kernel_stats_headline.csv
ls $OUT
