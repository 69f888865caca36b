#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j; mkdir -p $O
R=$GRAFT_REPO_ROOT
python tools/run_headline_kernel.py 6 64 512 500 race | tee $O/race_time.json
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  (cd $R && timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/$O/pmc_r_$i -o p -- python tools/run_headline_kernel.py 4 64 512 500 race > $R/$O/pmc_r_$i.log 2>&1)
done
cd $R && python tools/pmc_summary.py $O tsp_sample_kernel 2>&1 | tail -22
