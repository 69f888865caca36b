#!/usr/bin/env python3
"""Runs Net.forward_batch (B graphs, eval mode) a few times: the command profiled for the GNN rows of
profiles/ (rocprofv3 --kernel-trace --stats / --pmc).  usage: tools/run_gnn_batch.py [n] [k] [B] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402
from deepaco_amd.tsp.net import Net  # noqa: E402

n, k, B, reps = (int(x) for x in (sys.argv[1:] + ["500", "50", "64", "5"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = Net().to(dev).eval()
coords = torch.rand(B, n, 2, device=dev)
_, ei, ea = engine.tsp_knn_graph(coords, k, want_dist=False)
for _ in range(reps):
    heu = net.forward_batch(coords, ei, ea, k_sparse=k)
torch.cuda.synchronize()
print("ok", tuple(heu.shape), float(heu.mean()))
