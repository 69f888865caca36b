#!/usr/bin/env python3
"""The reference's tsp_nls/test.py call pattern on ONE instance (infer_instance: ACO(n_ants = 48, local_search = 'nls'),
run(t, inference=True) for t = 1 .. 10 iterations, TSP-500, heuristic from the network with random weights + 1e-10):
seconds per ACO iteration with the drop-in class, candidate-list 2-opt (default) and dense kernel only."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd.tsp_nls.aco import ACO  # noqa: E402
from deepaco_amd.tsp_nls.net import Net  # noqa: E402
from deepaco_amd.tsp_nls.utils import gen_pyg_data  # noqa: E402
from deepaco_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
n, k, A, iters = int(sys.argv[1]) if len(sys.argv) > 1 else 500, None, 48, 10
k = n // 10
torch.manual_seed(0)
net = Net().to(dev).eval()
coords = torch.rand(n, 2, device=dev)
pyg, dist = gen_pyg_data(coords, k_sparse=k, start_node=0)
with torch.no_grad():
    heu = net.reshape(pyg, net(pyg)) + 1e-10
out = {}
for label in ("candidate_lists", "dense_only"):
    if label == "dense_only":
        keep = engine.two_opt_tables
        engine.two_opt_tables = lambda *a, **kw: None
    aco = ACO(n_ants=A, heuristic=heu.cpu(), distances=dist.cpu(), device="cpu", local_search="nls", seed=1)
    aco.run(1, inference=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    best = aco.run(iters, inference=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    out[label] = {"seconds_per_iteration": dt, "best_cost": float(best)}
    if label == "dense_only":
        engine.two_opt_tables = keep
print(json.dumps({"workload": f"one TSP-{n} instance, {A} ants, NLS inference (maxt = 10000), class surface", **out}))
