#!/bin/bash
cd $GRAFT_REPO_ROOT
for nt in 0 1; do DACO_GNN_NT=$nt timeout 300 python tools/gnn_fused_check.py 2>&1 | grep '"-1"'; done
