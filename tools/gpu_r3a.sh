#!/bin/bash
cd $GRAFT_REPO_ROOT
for L in 0 8 16; do DACO_SCAN_LAYOUT=$L timeout 200 python tools/run_train_step.py 2>&1 | tail -1 | cut -c1-160; done
for L in 0 16; do DACO_SCAN_LAYOUT=$L timeout 200 python tools/time_layouts.py 100 30 20 2>&1 | tail -1; done
for L in 0 16; do DACO_SCAN_LAYOUT=$L timeout 200 python tools/time_layouts.py 100 50 1 2>&1 | tail -1; done
