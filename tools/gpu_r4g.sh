#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
timeout 300 python tools/gnn_fused_check.py 2>&1 | grep -v '"8"\|"11"\|"16"' | tee $O/fused_v2.jsonl
DACO_GNN_INPLACE=1 timeout 300 python tools/gnn_fused_check.py 2>&1 | grep -v '"8"\|"11"\|"16"' | tee $O/fused_v2_inplace.jsonl
DACO_GNN_INPLACE=1 timeout 600 python -m pytest tests/test_gpu_07_net.py -x -q -m gpu 2>&1 | tail -2
bash tools/gpu_r4h.sh
