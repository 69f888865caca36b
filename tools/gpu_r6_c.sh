#!/bin/bash
# Round 6, GPU session C: session B again with bounded tests (a hung kernel cost sessions A and B their test phase): the local
# search's queue fetch first, the suite with a per-test limit, the headline with and without the fused head rows, the epilogue
# ablation, one instance (the LDS-heads variant), kernel stats of the headline.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06c
mkdir -p $OUT
cd $R
timeout 200 python -m pytest tests/test_gpu_02_cvrp.py tests/test_gpu_13_hgs_ls.py -q -x --timeout 90 -k "surface_float64 or hgs" > $OUT/pytest_hgs.log 2>&1
echo "hgs rc=$?" | tee -a $OUT/pytest_hgs.log
tail -3 $OUT/pytest_hgs.log
timeout 900 python -m pytest tests -m gpu -q --timeout 240 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
for i in 1 2; do
  timeout 120 python bench.py --no-cpu --no-extras --min-seconds 2 --steps 20 2>/dev/null | tail -1 > $OUT/headline_fused_$i.json
  DACO_FUSE_HEAD_ROWS=0 timeout 120 python bench.py --no-cpu --no-extras --min-seconds 2 --steps 20 2>/dev/null | tail -1 > $OUT/headline_prepass_$i.json
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r06c/headline_*.json"))):
    try:
        j=json.load(open(f)); print(os.path.basename(f), j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], (j.get("sustained") or {}).get("value"))
    except Exception as e: print(f, e)
PY
timeout 120 python tools/ablate_epilogue.py > $OUT/ablate_epilogue.json 2>$OUT/ablate_epilogue.err; cat $OUT/ablate_epilogue.json
timeout 200 python tools/b1_modes.py 300 > $OUT/b1_modes.txt 2>&1; tail -8 $OUT/b1_modes.txt
DACO_SPARSE_LDS_HEADS=0 timeout 200 python tools/b1_modes.py 300 > $OUT/b1_modes_no_lds_heads.txt 2>&1; tail -8 $OUT/b1_modes_no_lds_heads.txt | grep scan_sparse
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -o p -- python bench.py --no-cpu --no-extras --min-seconds 0 > $OUT/stats_headline.log 2>&1)
cp $OUT/stats_headline/p_kernel_stats.csv $OUT/kernel_stats_headline.csv 2>/dev/null; rm -rf $OUT/stats_headline
head -8 $OUT/kernel_stats_headline.csv | cut -c1-160
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b1 -o p -- python tools/b1_modes.py 100 > $OUT/stats_b1.log 2>&1)
cp $OUT/stats_b1/p_kernel_stats.csv $OUT/kernel_stats_b1.csv 2>/dev/null; rm -rf $OUT/stats_b1
head -8 $OUT/kernel_stats_b1.csv | cut -c1-160
ls $OUT
