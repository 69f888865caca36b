#!/bin/bash
# Round-3 evidence in one GPU session (copy what should be judged from gpurun_out/<tag>/ into profiles/).
# usage (on the GPU box, from the repo root):  bash tools/profile_r3.sh r03
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CGROUPS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
        "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
        "FETCH_SIZE TCC_HIT_sum"
        "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"
        "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
pmc() {   # pmc <tag> <command...>: one rocprofv3 pass per counter group (kernel trace only, as the pool requires for --pmc)
  local tag=$1; shift
  local i=0
  for grp in "${CGROUPS[@]}"; do
    i=$((i+1))
    (cd $R && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$tag/pmc_${tag}_$i -o p -- "$@" > $OUT/pmc_$tag.$i.log 2>&1)
  done
  python $R/tools/pmc_summary.py $OUT/pmc_$tag daco > $OUT/pmc_$tag.txt 2>&1
  find $OUT/pmc_$tag -name "*.db" -delete 2>/dev/null
}
if [ "${2:-all}" != "pmc" ]; then
# 1. the bench line (with extras and CPU legs), then the same command without the CPU legs under the kernel trace
(cd $R && python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log)
(cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python bench.py --no-cpu --min-seconds 0 > $OUT/bench_profiled.log 2>&1)
cp $OUT/stats/p_kernel_stats.csv $OUT/kernel_stats_bench_default.csv 2>/dev/null
find $OUT/stats -name "*.db" -delete 2>/dev/null
fi
# 2. counters of the dominant kernels
pmc headline python bench.py --no-cpu --no-extras --min-seconds 0 --steps 3 --warmup 1
pmc c2 python tools/measure_configs.py c2
pmc c4 python tools/measure_configs.py c4
pmc c5 python tools/measure_configs.py c5shard
pmc nls python tools/run_nls_c3.py 64
pmc gnn python tools/run_gnn_batch.py 500 50 64 3
if [ "${2:-all}" != "pmc" ]; then
# 3. single colony, two ranks on one GPU
(cd $R && python bench.py --no-cpu --no-extras --min-seconds 0 --batch 1 --steps 50 --warmup 5 2>/dev/null | grep "^{" > $OUT/bench_b1.json)
(cd $R && python bench.py --gpus 2 --dist-backend gloo --force-device 0 --no-cpu --no-extras --min-seconds 0 --batch 32 2>/dev/null | grep "^{" > $OUT/bench_2ranks_one_gpu.json)
# 4. CVRP local search and training step timings
(cd $R && python tools/measure_cvrp_ls.py > $OUT/cvrp_ls.json 2>&1)
(cd $R && python tools/run_train_step.py > $OUT/train_step.json 2>&1)
(cd $R && python tools/run_single_instance_nls.py > $OUT/single_instance_nls.json 2>&1)
fi
ls $OUT
