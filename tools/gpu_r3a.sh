#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
O=gpurun_out/r3f
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python tests/soak_parity.py 2000 4321 > $O/soak_parity.txt 2>&1
tail -3 $O/soak_parity.txt
timeout 1500 bash tools/profile_r3.sh r03 > $O/profile.log 2>&1
tail -2 $O/profile.log
grep -h '^{' gpurun_out/r03/bench_default.json | cut -c1-300
