#!/bin/bash
# rocprofv3 PMC passes (kernel trace only) for an arbitrary command.  usage: tools/pmc_generic.sh <outdir> <tag> -- <command...>
set -u
OUT=$1; TAG=$2; shift 3
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/$OUT
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "FETCH_SIZE TCC_HIT_sum" \
           "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  (cd $ROOT && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_${TAG}_$i -o p -- "$@" > $ROOT/$OUT/pmc_${TAG}_$i.log 2>&1)
  echo "pass $i rc=$?"
done
