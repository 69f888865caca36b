#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/gnn_group_experiment.py > gpurun_out/r4w_gnn.jsonl 2>gpurun_out/r4w_gnn.err
DACO_GNN_INPLACE=1 timeout 300 python tools/gnn_group_experiment.py >> gpurun_out/r4w_gnn.jsonl 2>>gpurun_out/r4w_gnn.err
