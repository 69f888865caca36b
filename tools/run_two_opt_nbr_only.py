#!/usr/bin/env python3
"""The candidate-list 2-opt kernel alone (kernel="nbr"), for rocprofv3: `heavy` = first pass on tours sampled with the dense
1/d heuristic (candidate lists of ~100 k per sweep), `light` = a repair pass on perturbed, nearly 2-optimal tours."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "light"
B, n, A = 16, 500, 256
g = torch.Generator().manual_seed(2)
c = torch.rand(B, n, 2, generator=g)
d = torch.cdist(c, c)
d = ((d + d.transpose(1, 2)) / 2).contiguous()
i = torch.arange(n)
d[:, i, i] = 1e9
d = d.to(dev)
eta = 1 / d
paths, _, _, _ = engine.tsp_sample(torch.ones_like(d), eta, A, mode="scan", seed=4, fixed_start=0)
tours = paths.permute(0, 2, 1).to(torch.int16).contiguous()
td = engine.TwoOptTables(d)
if mode == "light":
    hd = (1 / (eta / eta.amax(dim=-1, keepdim=True) + 1e-5)).contiguous()
    th = engine.TwoOptTables(hd)
    engine.two_opt_(d, tours, 10000, tables=td)                 # to convergence
    engine.two_opt_(hd, tours, 20, tables=th)                   # perturbed
torch.cuda.synchronize()
t0 = time.perf_counter()
_, sw = engine.two_opt_(d, tours.clone(), n // 4, want_sweeps=True, tables=td, kernel="nbr")
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"mode": mode, "seconds": dt, "sweeps": int(sw.sum()), "sweeps_per_s": float(sw.sum()) / dt}))
