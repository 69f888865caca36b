#!/bin/bash
# Regenerates the round's evidence under gpurun_out/<tag>/ in one GPU session (copy what should be judged into
# profiles/).  usage (on the GPU box, from the repo root):  bash tools/profile_all.sh r02
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. 2-opt counters first (bench.py's config-3 roofline reads profiles/two_opt_l2.json): one NLS iteration of config 3's
# colony, 16 instances (tools/run_nls_c3.py runs 3 iterations), all 2-opt kernels
(cd $R && bash tools/pmc_generic.sh gpurun_out/$TAG/pmc2opt 2opt -- python tools/run_nls_c3.py 16 > /dev/null)
python $R/tools/pmc_summary.py $OUT/pmc2opt two_opt > $OUT/pmc_2opt_nls_c3.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nls_stats -o p -- python $R/tools/run_nls_c3.py 16 > $OUT/nls_c3_b16.log 2>&1
cp $OUT/nls_stats/p_kernel_stats.csv $OUT/kernel_stats_nls_c3_b16.csv
python3 - <<EOF
import json, re
txt = open("$OUT/pmc_2opt_nls_c3.txt").read()
total = 0.0
for blk in txt.split("## ")[1:]:
    m = re.search(r"TCP_TCC_READ_REQ_sum\s+([0-9.e+]+)\s+\(mean of (\d+) dispatches\)", blk)
    if m:
        total += float(m.group(1)) * int(m.group(2))
if total:
    json.dump({"l2_read_bytes_per_tour_iteration": total * 128.0 / (3 * 16 * 256),
               "source": "profiles/${TAG}_pmc_2opt_nls_c3.txt: sum over the 2-opt kernels of TCP_TCC_READ_REQ_sum x dispatches x 128 B / "
                         "(3 iterations x 16 instances x 256 tours) of tools/run_nls_c3.py 16 (rocprofv3 --pmc)"},
              open("$R/profiles/two_opt_l2.json", "w"), indent=1)
    import shutil
    shutil.copy("$R/profiles/two_opt_l2.json", "$OUT/two_opt_l2.json")      # (profiles/ on the GPU box does not travel back)
EOF
# 2. the bench line, its kernel trace, its counters
python $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python $R/bench.py --no-cpu --no-extras --min-seconds 0 > $OUT/bench_profiled.log 2>&1
cp $OUT/stats/p_kernel_stats.csv $OUT/kernel_stats_bench_default.csv
(cd $R && bash tools/pmc_pass.sh gpurun_out/$TAG/pmc final -- --no-extras --min-seconds 0 > /dev/null && python tools/pmc_summary.py gpurun_out/$TAG/pmc daco > $OUT/pmc_summary.txt)
python $R/bench.py --no-cpu --no-extras --min-seconds 0 --batch 1 --steps 50 --warmup 5 2>/dev/null | grep "^{" > $OUT/bench_b1.json
# 3. multi-rank self-launch (two ranks on this one GPU, gloo rendezvous)
python $R/bench.py --gpus 2 --dist-backend gloo --force-device 0 --no-cpu --no-extras --min-seconds 0 --batch 32 2>/dev/null | grep "^{" > $OUT/bench_2ranks_one_gpu.json
# 4. training step: kernel list (no library GEMM) and timing
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_stats -o p -- python $R/tools/run_train_step.py 5 > $OUT/train_step.log 2>&1
cp $OUT/train_stats/p_kernel_stats.csv $OUT/kernel_stats_train_step.csv
python $R/tools/run_train_step.py 5 2>/dev/null | grep "^{" > $OUT/train_step.json
python $R/tools/run_train_step_500.py 5 2>/dev/null | grep "^{" >> $OUT/train_step.json
python $R/tools/bench_two_opt_nbr.py 16 > $OUT/two_opt_nbr.txt 2>/dev/null
python $R/tools/gnn_fused_check.py > $OUT/gnn_fused_check.txt 2>/dev/null
python $R/tools/run_single_instance_nls.py 500 2>/dev/null | grep "^{" > $OUT/single_instance_nls.txt
python $R/tools/run_single_instance_nls.py 200 2>/dev/null | grep "^{" >> $OUT/single_instance_nls.txt
python $R/tools/nls_sweep_stats.py 64 auto 2>/dev/null > $OUT/nls_sweep_stats.txt
# 5. GNN batch inference profile, shapes microbenchmarks
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gnn_stats -o p -- python $R/tools/run_gnn_batch.py > /dev/null 2>&1
cp $OUT/gnn_stats/p_kernel_stats.csv $OUT/kernel_stats_gnn_batch64_n500.csv
$R/tools/l2_bw_shapes 64 0 > $OUT/l2_bw_shapes.txt 2>&1
$R/tools/scan32_ablate > $OUT/scan32_ablate.txt 2>&1
tail -1 $OUT/bench_default.json | cut -c1-300
