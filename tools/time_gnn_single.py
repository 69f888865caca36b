#!/usr/bin/env python3
"""ms per Net.forward of ONE graph (the reference's inference call: one instance at a time), TSP-n with k = n / 10 neighbours, over
the fused layer kernel's nodes per wave (DACO_GNN_FUSED_NPW; default = the library's choice)."""
import json
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
for n, k in ((200, 20), (500, 50), (1000, 100)):
    coords = torch.rand(1, n, 2, device=dev)
    _, ei, ea = engine.tsp_knn_graph(coords, k, want_dist=False)
    row = {"n": n, "k": k}
    ref = None
    for npw in ("default", "1", "2", "4", "8", "16"):
        if npw == "default":
            os.environ.pop("DACO_GNN_FUSED_NPW", None)
        else:
            os.environ["DACO_GNN_FUSED_NPW"] = npw
        for _ in range(5):
            h = net.forward_batch(coords, ei, ea, k_sparse=k)
        if ref is None:
            ref = h
        row["max_abs_diff_" + npw] = float((h - ref).abs().max())
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            net.forward_batch(coords, ei, ea, k_sparse=k)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        row[npw] = round(ts[len(ts) // 2], 4)
    print(json.dumps(row), flush=True)
