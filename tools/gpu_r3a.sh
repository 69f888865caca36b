#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
O=gpurun_out/r3j
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
