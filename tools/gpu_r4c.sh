#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
for k in 0 1; do
  DACO_SCAN32_KNOB=$k timeout 120 python tools/ablate_scan32.py 500 2>/dev/null | tee -a $O/knob.jsonl
done
DACO_LD_PAD=64 timeout 120 python tools/ablate_scan32.py 500 2>/dev/null | tee -a $O/ldpad.jsonl
DACO_LD_PAD=64 DACO_SCAN32_KNOB=1 timeout 120 python tools/ablate_scan32.py 500 2>/dev/null | tee -a $O/ldpad.jsonl
