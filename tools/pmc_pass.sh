#!/bin/bash
# One rocprofv3 PMC pass per counter group (separate runs: gfx950 has 8 SQ / 4 TCC slots and
# FETCH_SIZE costs 3 TCC slots).  Kernel trace only, as the GPU pool requires for --pmc runs.
# usage: tools/pmc_pass.sh <outdir> <tag> -- <bench args...>
set -u
OUT=$1; TAG=$2; shift 3
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE TCC_HIT_sum" \
           "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_${TAG}_$i -o p -- python $ROOT/bench.py --no-cpu --steps 3 --warmup 1 "$@" > $ROOT/$OUT/pmc_${TAG}_$i.log 2>&1
  echo "pass $i rc=$?"
done
