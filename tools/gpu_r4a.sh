#!/bin/bash
# round-4 GPU session A: headline kernel after the VALU cut -- parity first, then time, then counters
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_00_tsp.py tests/test_gpu_03_two_opt.py -x -q -m gpu -k "not soak" 2>&1 | tail -5 > $O/pytest.log; tail -3 $O/pytest.log
timeout 200 python tools/ablate_scan32.py 500 > $O/ablate.json 2>$O/ablate.err; cat $O/ablate.json
timeout 300 python bench.py --no-cpu --no-extras --min-seconds 0 2>$O/bench.err | grep '^{' > $O/bench.json; cut -c1-400 $O/bench.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_1 -o p -- python $R/bench.py --no-cpu --no-extras --min-seconds 0 --steps 3 --warmup 1 > $R/$O/pmc_1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/$O/pmc_2 -o p -- python $R/bench.py --no-cpu --no-extras --min-seconds 0 --steps 3 --warmup 1 > $R/$O/pmc_2.log 2>&1
cd $R && python tools/pmc_summary.py $O scan32 2>&1 | tail -30
