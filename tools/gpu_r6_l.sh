#!/bin/bash
# Round 6, GPU session L: smoke(), the suite, one instance, the soak of the head-row sampler, the counter passes session K left out.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06l
mkdir -p $OUT
cd $R
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log | cut -c1-300
timeout 900 python -m pytest tests -m gpu -q --timeout 240 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout 200 python tools/b1_modes.py 300 > $OUT/b1_modes.txt 2>&1; grep '"n": 500' $OUT/b1_modes.txt
for i in 1 2; do timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras --min-seconds 2 2>/dev/null | tail -1 > $OUT/headline_$i.json; done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r06l/headline_*.json"))):
    j=json.load(open(f)); print(os.path.basename(f), j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"])
PY
SOAK_FULL_B=16 timeout 300 python tools/soak_scan_sparse.py 60 6 > $OUT/soak_scan_sparse.txt 2>&1; tail -2 $OUT/soak_scan_sparse.txt | cut -c1-600
bash tools/profile_r6.sh r06l/prof rest > $OUT/profile_rest.log 2>&1
mkdir -p gpurun_out/r06l/prof; cp -r gpurun_out/r06k/prof/pmc_* gpurun_out/r06l/prof/ 2>/dev/null
python tools/make_counters.py gpurun_out/r06l/prof --merge > $OUT/make_counters.log 2>&1
cp profiles/counters.json profiles/hbm_traffic.json $OUT/ 2>/dev/null
grep -c . $OUT/make_counters.log; ls $OUT $OUT/prof | head -40
