#!/bin/bash
# memory-pipe counters of the headline launch (TA / TCP / TD), one rocprofv3 pass per group
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN2_sum" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  (cd $R && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/$O/pmc_h_$i -o p -- python tools/run_headline_kernel.py 6 > $R/$O/pmc_h_$i.log 2>&1)
  echo "pass $i rc=$? $(tail -1 $R/$O/pmc_h_$i.log | cut -c1-150)"
done
cd $R && python tools/pmc_summary.py $O scan32 2>&1 | tail -60
