#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
O=gpurun_out/r3l
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 200 python bench.py --no-cpu --no-extras --min-seconds 0 2>/dev/null | grep '^{' | cut -c1-220
