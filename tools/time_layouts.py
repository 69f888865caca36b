#!/usr/bin/env python3
"""Time one AS iteration of batched TSP / CVRP colonies at a given size under the current DACO_SCAN_LAYOUT.
usage: tools/time_layouts.py n [A B]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1])
A = int(sys.argv[2]) if len(sys.argv) > 2 else 512
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
g = torch.Generator().manual_seed(n)
c = torch.rand(B, n, 2, generator=g)
d = (c[:, :, None] - c[:, None]).norm(dim=-1)
i = torch.arange(n)
d[:, i, i] = 1e9
out = {"n": n, "A": A, "B": B, "layout": os.environ.get("DACO_SCAN_LAYOUT", "default")}
col = engine.BatchedTSP(d.to(dev), n_ants=A)
for _ in range(3):
    col.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    col.step()
torch.cuda.synchronize()
out["tsp_ms"] = (time.perf_counter() - t0) / 20 * 1e3
d[:, i, i] = 1e-10
dem = torch.cat((torch.zeros(B, 1), torch.randint(1, 10, (B, n - 1), generator=g).float()), 1)
ccol = engine.BatchedCVRP(d.to(dev), dem.to(dev), n_ants=A, capacity=50.0)
for _ in range(3):
    ccol.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    ccol.step()
torch.cuda.synchronize()
out["cvrp_ms"] = (time.perf_counter() - t0) / 20 * 1e3
print(json.dumps(out))
