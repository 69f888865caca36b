#!/bin/bash
# Round-6 evidence in one GPU session: kernel stats (rocprofv3 --kernel-trace --stats) of the default bench, the headline alone,
# a training step and the siblings; counter passes (rocprofv3 --pmc, kernel trace only, one pass per counter group) of the
# dominant kernels.  Copy what should be judged from gpurun_out/<tag>/ into profiles/; tools/make_counters.py turns the counter
# passes into profiles/counters.json and profiles/hbm_traffic.json.
# usage (on the GPU box, from the repo root):  bash tools/profile_r6.sh r06 [pmc|stats|hgs|gnn|nls|deposit|headline|all]
set -u
TAG=${1:-r06}
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
        "TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TD_TC_STALL_sum"
        "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum")
pmc() {
  local tag=$1; shift
  local i=0
  for grp in "${CGROUPS[@]}"; do
    i=$((i+1))
    (cd $R && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$tag/pmc_${tag}_$i -o p -- "$@" > $OUT/pmc_$tag.$i.log 2>&1)
  done
  python $R/tools/pmc_summary.py $OUT/pmc_$tag daco > $OUT/pmc_$tag.txt 2>&1
  find $OUT/pmc_$tag -name "*.db" -delete 2>/dev/null
}
stats() {
  local tag=$1; shift
  (cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$tag -o p -- "$@" > $OUT/stats_$tag.log 2>&1)
  cp $OUT/stats_$tag/p_kernel_stats.csv $OUT/kernel_stats_$tag.csv 2>/dev/null
  find $OUT/stats_$tag -name "*.db" -delete 2>/dev/null
  rm -rf $OUT/stats_$tag
}
if [ "$WHAT" = "hgs" ]; then      # the local-search kernel alone (after a change to it)
  stats hgs_ls python tools/bench_hgs_ls.py --batch 64 --no-short
  pmc hgs_ls python tools/bench_hgs_ls.py --batch 64 --no-short --reps 2
  stats train python tools/run_train_step.py 5
  ls $OUT
  exit 0
fi
if [ "$WHAT" = "nls" ]; then      # the fused NLS kernel alone at config 3 (after a change to it)
  stats nls python tools/run_nls_c3.py 64
  pmc nls python tools/run_nls_c3.py 64
  ls $OUT
  exit 0
fi
if [ "$WHAT" = "deposit" ]; then  # the update with the head rows alone (after a change to it)
  pmc deposit_heads python bench.py --no-cpu --no-extras --min-seconds 0 --steps 5 --precondition-seconds 0
  ls $OUT
  exit 0
fi
if [ "$WHAT" = "gnn" ]; then      # the network's forward alone (after a change to its kernels)
  stats gnn python tools/time_gnn_batch.py
  pmc gnn python tools/run_gnn_batch.py 500 50 64 3
  ls $OUT
  exit 0
fi
if [ "$WHAT" = "rest" ]; then    # the counter passes `headline` leaves out (make_counters.py --merge puts the two together)
  pmc headline python tools/run_headline_kernel.py 5 64 512 500 scan
  pmc c2 python tools/measure_configs.py c2
  pmc c4 python tools/measure_configs.py c4
  pmc c5 python tools/measure_configs.py c5shard
  pmc hgs_ls python tools/bench_hgs_ls.py --batch 64 --no-short --reps 2
  ls $OUT
  exit 0
fi
if [ "$WHAT" = "headline" ]; then # the headline's kernels only: the construction kernel, the update with the head rows, one instance
  stats headline python bench.py --no-cpu --no-extras --min-seconds 0
  stats b1 python tools/b1_modes.py 100
  pmc scan_sparse python tools/run_headline_kernel.py 5 64 512 500 scan_sparse
  pmc deposit_heads python bench.py --no-cpu --no-extras --min-seconds 0 --steps 5 --precondition-seconds 0
  pmc b1_lds_heads python tools/b1_modes.py 20
  pmc nls python tools/run_nls_c3.py 64
  ls $OUT
  exit 0
fi
if [ "$WHAT" != "pmc" ]; then
  stats headline python bench.py --no-cpu --no-extras --min-seconds 0
  stats headline_dense python bench.py --no-cpu --no-extras --min-seconds 0 --sampler scan
  stats bench_default python bench.py --no-cpu --min-seconds 0
  stats train python tools/run_train_step.py 5
  stats siblings python tools/measure_siblings.py
  stats hgs_ls python tools/bench_hgs_ls.py --batch 64 --no-short
  (cd $R && python tools/run_train_step.py 10 > $OUT/train_step.json 2>/dev/null)
fi
if [ "$WHAT" != "stats" ]; then
  pmc scan_sparse python tools/run_headline_kernel.py 5 64 512 500 scan_sparse
  pmc deposit_heads python bench.py --no-cpu --no-extras --min-seconds 0 --steps 5 --precondition-seconds 0
  pmc b1_lds_heads python tools/b1_modes.py 20
  pmc headline python tools/run_headline_kernel.py 5 64 512 500 scan
  pmc hgs_ls python tools/bench_hgs_ls.py --batch 64 --no-short --reps 2
  pmc c2 python tools/measure_configs.py c2
  pmc c4 python tools/measure_configs.py c4
  pmc c5 python tools/measure_configs.py c5shard
  pmc nls python tools/run_nls_c3.py 64
  pmc gnn python tools/run_gnn_batch.py 500 50 64 3
fi
ls $OUT
