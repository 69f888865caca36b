#!/bin/bash
# Round-4 evidence in one GPU session: kernel stats and counter passes of the dominant kernels (copy what should be judged from
# gpurun_out/<tag>/ into profiles/; tools/make_counters.py turns the counter passes into profiles/counters.json and
# profiles/hbm_traffic.json).  usage (on the GPU box, from the repo root):  bash tools/profile_r4.sh r04 [pmc|stats|all]
set -u
TAG=${1:-r04}
WHAT=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CGROUPS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
        "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
        "FETCH_SIZE TCC_HIT_sum"
        "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"
        "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
        "TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TD_TC_STALL_sum")
pmc() {   # pmc <tag> <command...>: one rocprofv3 pass per counter group (kernel trace only, as the pool requires for --pmc)
  local tag=$1; shift
  local i=0
  for grp in "${CGROUPS[@]}"; do
    i=$((i+1))
    (cd $R && timeout 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$tag/pmc_${tag}_$i -o p -- "$@" > $OUT/pmc_$tag.$i.log 2>&1)
  done
  python $R/tools/pmc_summary.py $OUT/pmc_$tag daco > $OUT/pmc_$tag.txt 2>&1
  find $OUT/pmc_$tag -name "*.db" -delete 2>/dev/null
}
stats() {  # stats <tag> <command...>: rocprofv3 --kernel-trace --stats of one command
  local tag=$1; shift
  (cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$tag -o p -- "$@" > $OUT/stats_$tag.log 2>&1)
  cp $OUT/stats_$tag/p_kernel_stats.csv $OUT/kernel_stats_$tag.csv 2>/dev/null
  find $OUT/stats_$tag -name "*.db" -delete 2>/dev/null
}
if [ "$WHAT" != "pmc" ] && [ "$WHAT" != "sparse" ]; then
  # the headline launches ONLY (no extras, no learned / gap colonies in the same kernel row), then the whole default bench
  stats headline python bench.py --no-cpu --no-extras --min-seconds 0
  stats bench_default python bench.py --no-cpu --min-seconds 0
  (cd $R && python bench.py --no-cpu --no-extras --min-seconds 0 --batch 1 --steps 50 --warmup 5 2>/dev/null | grep "^{" > $OUT/bench_b1.json)
  (cd $R && python bench.py --gpus 2 --dist-backend gloo --force-device 0 --no-cpu --no-extras --min-seconds 0 --batch 32 2>/dev/null | grep "^{" > $OUT/bench_2ranks_one_gpu.json)
  (cd $R && python tools/run_train_step.py > $OUT/train_step.json 2>&1)
  (cd $R && python tools/run_single_instance_nls.py > $OUT/single_instance_nls.json 2>&1)
fi
if [ "$WHAT" = "sparse" ]; then      # the head / tail kernels only (the other kernels' passes stay valid while their code does)
  pmc scan_sparse python tools/run_headline_kernel.py 5 64 512 500 scan_sparse
  pmc race_head python tools/run_headline_kernel.py 4 64 512 500 race_head
  pmc c5_sparse python tools/run_headline_kernel.py 4 64 2048 1000 scan_sparse
  ls $OUT
  exit 0
fi
if [ "$WHAT" != "stats" ]; then
  pmc headline python tools/run_headline_kernel.py 5 64 512 500 scan
  pmc scan_sparse python tools/run_headline_kernel.py 5 64 512 500 scan_sparse
  pmc race python tools/run_headline_kernel.py 4 64 512 500 race
  pmc race_head python tools/run_headline_kernel.py 4 64 512 500 race_head
  pmc c5_sparse python tools/run_headline_kernel.py 4 64 2048 1000 scan_sparse
  pmc c2 python tools/measure_configs.py c2
  pmc c4 python tools/measure_configs.py c4
  pmc c5 python tools/measure_configs.py c5shard
  pmc nls python tools/run_nls_c3.py 64
  pmc gnn python tools/run_gnn_batch.py 500 50 64 3
  pmc cvrp_ls python tools/measure_cvrp_ls.py 16
fi
ls $OUT
