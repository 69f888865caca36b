#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export DACO_GNN_INPLACE=1
i=0
for grp in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_IFETCH" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_CYCLES SQ_WAVES_EQ_64" \
           "TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TD_TC_STALL_sum"; do
  i=$((i+1))
  (cd $R && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/$O/pmc_g_$i -o p -- python tools/run_gnn_batch.py 500 50 64 3 > $R/$O/pmc_g_$i.log 2>&1)
  echo "pass $i rc=$?"
done
cd $R && python tools/pmc_summary.py $O fused2 2>&1 | tail -40
