#!/bin/bash
# Round 6, GPU session BI: the pheromone update's chunk loader issues its loads together: parity (every test that runs the update),
# the headline iteration against the previous build on one box (driver's command line, no CPU leg), kernel statistics.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06bi
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_00_tsp.py tests/test_gpu_11_scan_sparse.py tests/test_gpu_15_full_batch.py tests/test_gpu_02_cvrp.py tests/test_gpu_05_siblings.py tests/test_gpu_12_streams.py -m gpu -q --timeout 600 -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log | cut -c1-300
for i in 1 2 3; do
  for v in new prev; do
    L=$R/deepaco_amd/lib/libdeepaco_hip.so; [ $v = prev ] && L=$R/deepaco_amd/lib/libdeepaco_hip_prev.so
    DACO_LIB_PATH=$L timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" | tee -a $OUT/ab_bench.txt
  done
done
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_headline -o p -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras > $OUT/stats_headline.log 2>&1)
cp $(find /tmp/stats_headline -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_headline.csv
head -5 $OUT/kernel_stats_headline.csv | cut -c1-70,200-300
