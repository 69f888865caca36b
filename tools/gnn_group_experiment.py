#!/usr/bin/env python3
"""GNN forward: does the edge state of a GROUP of graphs stay in the L2 across the twelve layers?  Times Net.forward_batch on
B graphs of TSP-500 (k = 50) for several B and nodes-per-wave settings and prints the time per graph.
usage: tools/gnn_group_experiment.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402
from deepaco_amd.tsp.net import Net  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = Net().to(dev).eval()
n, k = 500, 50
coords64 = torch.rand(64, n, 2, device=dev)
for B in (64, 32, 16, 8, 4, 2):
    coords = coords64[:B].contiguous()
    _, ei, ea = engine.tsp_knn_graph(coords, k, want_dist=False)
    for npw in ("-1", "2", "4", "8"):
        if npw == "-1":
            os.environ.pop("DACO_GNN_FUSED_NPW", None)
        else:
            os.environ["DACO_GNN_FUSED_NPW"] = npw
        with torch.no_grad():
            for _ in range(3):
                net.forward_batch(coords, ei, ea, k_sparse=k)
            torch.cuda.synchronize()
            reps = 20
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                net.forward_batch(coords, ei, ea, k_sparse=k)
            b.record()
            torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        print(json.dumps({"B": B, "npw": npw, "ms": round(ms, 4), "us_per_graph": round(ms * 1e3 / B, 2),
                          "inplace": os.environ.get("DACO_GNN_INPLACE", "")}), flush=True)
