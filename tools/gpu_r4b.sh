#!/bin/bash
# round-4 GPU session B: how the headline kernel's time depends on the ants in flight per CU (occupancy capped with dynamic LDS)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
for pad in 0 2560 5120 9216 14848; do
  DACO_SCAN32_LDS_PAD=$pad timeout 120 python tools/ablate_scan32.py 500 2>/dev/null | tee -a $O/occupancy.jsonl
done
DACO_SCAN_LAYOUT=16 timeout 120 python tools/ablate_scan32.py 500 2>/dev/null | tee -a $O/layout16.json
