#!/usr/bin/env python3
"""ms per Net.forward_batch (64 graphs of TSP-500, k = 50, eval): median of 20 timed forwards after 5 warm-ups."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402
from deepaco_amd.tsp.net import Net  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = Net().to(dev).eval()
n, k, B = 500, 50, 64
coords = torch.rand(B, n, 2, device=dev)
_, ei, ea = engine.tsp_knn_graph(coords, k, want_dist=False)
for _ in range(5):
    net.forward_batch(coords, ei, ea, k_sparse=k)
torch.cuda.synchronize()
ts = []
for _ in range(20):
    t0 = time.perf_counter()
    net.forward_batch(coords, ei, ea, k_sparse=k)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print(f"forward_batch 64 x TSP-500 (k = 50): median {ts[10]:.3f} ms, min {ts[0]:.3f} ms")
