#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_07_net.py -x -q -m gpu 2>&1 | tail -4 > $O/pytest.log; cat $O/pytest.log
for v in 1 2; do
  DACO_GNN_FUSED_V=$v timeout 300 python tools/gnn_fused_check.py 2>&1 | tee $O/fused_v$v.jsonl
done
