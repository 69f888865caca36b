#!/bin/bash
# Regenerates the round's evidence under gpurun_out/<tag>/ in one GPU session (copy what should be
# judged into profiles/).  usage (on the GPU box, from the repo root):  bash tools/profile_all.sh r02
#   bench line (+cpu leg), rocprofv3 kernel stats of the same command, PMC passes (HBM bytes, L2 hit rate,
#   wave-time split), end-to-end configs, layout sweep, L2 row-stream ceiling, GNN batch profile.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python $R/bench.py --no-cpu > $OUT/bench_profiled.log 2>&1
cp $OUT/stats/p_kernel_stats.csv $OUT/kernel_stats_bench_default.csv
(cd $R && bash tools/pmc_pass.sh gpurun_out/$TAG/pmc final -- > /dev/null && python tools/pmc_summary.py gpurun_out/$TAG/pmc daco > $OUT/pmc_summary.txt)
python $R/bench.py --no-cpu --batch 1 --steps 50 --warmup 5 2>/dev/null | grep "^{" > $OUT/bench_b1.json
python $R/tools/measure_configs.py headline c2 c3 c4 gnn 2>/dev/null | grep "^{" > $OUT/configs_end_to_end.jsonl
python $R/tools/sweep_layouts.py 2>/dev/null | grep "^{" > $OUT/sweep_layouts.jsonl
if [ ! -x $R/tools/l2_row_stream_bench ]; then /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/tools/l2_row_stream_bench.hip -o $R/tools/l2_row_stream_bench 2>/dev/null; fi
for m in 0 1 4; do $R/tools/l2_row_stream_bench $m | tail -1; done > $OUT/l2_row_stream.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gnn_stats -o p -- python $R/tools/run_gnn_batch.py > /dev/null 2>&1
cp $OUT/gnn_stats/p_kernel_stats.csv $OUT/kernel_stats_gnn_batch64_n500.csv
tail -1 $OUT/bench_default.json | cut -c1-240
