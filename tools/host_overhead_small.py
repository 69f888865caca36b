#!/usr/bin/env python3
"""How much of an ACO iteration at the reference's small sizes is host time: ms per iteration to ENQUEUE (before the sync) and in
total, tsp.ACO at TSP-20 / 100 (dense kernels) and TSP-500 x 50 ants (head rows), cvrp.ACO at CVRP-20 / 100."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd.tsp.aco import ACO as TspACO  # noqa: E402
from deepaco_amd.cvrp.aco import ACO as CvrpACO  # noqa: E402

dev = "cuda:0"
T = 500
for n, A, k in ((20, 20, 10), (100, 20, 20), (500, 50, 50)):
    g = torch.Generator().manual_seed(n)
    c = torch.rand(n, 2, generator=g)
    d = (c[:, None] - c).norm(dim=-1)
    d[torch.arange(n), torch.arange(n)] = 1e9
    a = TspACO(d.to(dev), n_ants=A, device=dev)
    a.sparsify(k)
    a.run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a.run(T)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(json.dumps({"class": "tsp.ACO", "n": n, "ants": A, "enqueue_ms_per_iteration": round((t1 - t0) / T * 1e3, 4),
                      "total_ms_per_iteration": round((t2 - t0) / T * 1e3, 4)}), flush=True)
for n, A in ((20, 20), (100, 20)):
    g = torch.Generator().manual_seed(n + 2)
    loc = torch.cat((torch.full((1, 2), 0.5), torch.rand(n, 2, generator=g)))
    dem = torch.cat((torch.zeros(1), torch.randint(1, 10, (n,), generator=g).float()))
    d = (loc[:, None] - loc).norm(dim=-1)
    d[torch.arange(n + 1), torch.arange(n + 1)] = 1e-10
    a = CvrpACO(d.to(dev), dem.to(dev), n_ants=A, device=dev, capacity=50)
    a.run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a.run(T)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(json.dumps({"class": "cvrp.ACO", "n": n, "ants": A, "enqueue_ms_per_iteration": round((t1 - t0) / T * 1e3, 4),
                      "total_ms_per_iteration": round((t2 - t0) / T * 1e3, 4)}), flush=True)
