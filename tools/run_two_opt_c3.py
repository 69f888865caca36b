#!/usr/bin/env python3
"""The 2-opt kernel on config 3's workload (TSP-500, 64 instances x 256 ACO-sampled tours, maxt = n//4 sweeps), alone,
for rocprofv3 (--kernel-trace --stats / --pmc).  Prints one JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
n, A, B = 500, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator().manual_seed(2)
c = torch.rand(B, n, 2, generator=g)
d = torch.cdist(c, c)
i = torch.arange(n)
d[:, i, i] = 1e9
d = d.to(dev)
col = engine.BatchedTSP(d, n_ants=A, seed=1, fixed_start=0)
col.sparsify(50)
paths, _, _, _ = engine.tsp_sample(col.pheromone, col.heuristic, A, seed=3, batch=B, fixed_start=0)
tours = paths.permute(0, 2, 1).to(torch.int16).contiguous()
t = tours.clone(); engine.two_opt_(d, t, n // 4, dist_t="symmetric")
torch.cuda.synchronize()
t0 = time.perf_counter()
t = tours.clone()
_, sw = engine.two_opt_(d, t, n // 4, want_sweeps=True, dist_t="symmetric")
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"workload": f"2-opt, TSP-{n}, {B} x {A} ACO-sampled tours, <= {n // 4} sweeps", "seconds": dt,
                  "sweeps": int(sw.sum()), "sweeps_per_s": float(sw.sum()) / dt, "tours_per_s": B * A / dt}))
