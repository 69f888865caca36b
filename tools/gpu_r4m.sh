#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4m; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum"; do
  i=$((i+1))
  (cd $R && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/$O/pmc_s_$i -o p -- python tools/run_headline_kernel.py 4 64 512 500 scan_sparse > $R/$O/pmc_s_$i.log 2>&1)
done
cd $R && python tools/pmc_summary.py $O scan_sparse 2>&1 | tail -40
