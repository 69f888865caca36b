#!/usr/bin/env python3
"""Config 3's NLS pass by pass (64 instances x 256 tours, TSP-500, sparsified heuristic): time and sweeps of every 2-opt call
of one iteration, candidate-list/auto path."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
n, A, B = 500, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 64
kernel = sys.argv[2] if len(sys.argv) > 2 else "auto"
g = torch.Generator().manual_seed(2)
c = torch.rand(B, n, 2, generator=g)
d = torch.cdist(c, c)
d = (d + d.transpose(1, 2)) / 2
i = torch.arange(n)
d[:, i, i] = 1e9
col = engine.BatchedTSP(d.to(dev), n_ants=A, seed=1, local_search="nls", fixed_start=0)
col.sparsify(50)
col.step()
paths, _, _, _ = engine.tsp_sample(col.pheromone, col.heuristic, A, seed=3, batch=B, fixed_start=0)
tours = paths.permute(0, 2, 1).to(torch.int16).contiguous()
hd = col._heuristic_dist()
rows = []


def call(name, m, t, maxit, tabs, mt):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if kernel == "dense":
        _, sw = engine.two_opt_(m, t, maxit, want_sweeps=True, dist_t=mt)
    else:
        _, sw = engine.two_opt_(m, t, maxit, want_sweeps=True, tables=tabs, kernel=kernel)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rows.append({"call": name, "ms": round(dt * 1e3, 2), "sweeps_mean": round(float(sw.float().mean()), 1),
                 "sweeps_max": int(sw.max()), "Msweeps_per_s": round(float(sw.sum()) / dt / 1e6, 1)})


cur = tours.clone()
call("first", col.distances, cur, n // 4, col._tables, col._dist_t)
for r in range(10):
    call(f"perturb{r}", hd, cur, 20, col._htables, col._hdist_t)
    call(f"repair{r}", col.distances, cur, n // 4, col._tables, col._dist_t)
for r in rows:
    print(json.dumps(r))
print(json.dumps({"total_ms": sum(r["ms"] for r in rows)}))
