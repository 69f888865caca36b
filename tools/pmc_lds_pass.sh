#!/bin/bash
# rocprofv3 PMC passes for the LDS / vector-memory pipes of the tour-construction kernel (kernel trace only).
# usage: tools/pmc_lds_pass.sh <outdir> <tag> -- <bench args...>
set -u
OUT=$1; TAG=$2; shift 3
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
i=0
for grp in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TA_BUSY_max TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_${TAG}_$i -o p -- python $ROOT/bench.py --no-cpu --no-extras --min-seconds 0 --steps 3 --warmup 1 "$@" > $ROOT/$OUT/pmc_${TAG}_$i.log 2>&1
  echo "pass $i rc=$?"
done
