#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3q
timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/r3q/pytest.log
timeout 300 python tools/soak_two_opt.py 120 4242 > gpurun_out/r3q/soak2.txt 2>&1
cat gpurun_out/r3q/pytest.log; tail -1 gpurun_out/r3q/soak2.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  (cd $R && timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_nls/pmc_nls_$i -o p -- python tools/run_nls_c3.py 64 > $O/pmc_nls.$i.log 2>&1)
done
python $R/tools/pmc_summary.py $O/pmc_nls nls_kernel > $O/pmc_nls.txt 2>&1
find $O -name "*.db" -delete
grep -E "INSTS_VALU|HBM-side|L2 hit|wave time|TCP_TCC_READ" $O/pmc_nls.txt
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python tools/run_nls_c3.py 64 > $O/nls_stats.log 2>&1)
grep nls_kernel $O/stats/p_kernel_stats.csv | cut -c1-200
find $O -name "*.db" -delete
