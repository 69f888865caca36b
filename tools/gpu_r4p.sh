#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $O/pytest_full.log; cat $O/pytest_full.log
