#!/usr/bin/env python3
"""One TSP-500 instance, 512 ants (the reference's own calling pattern, B = 1): colony iterations per second for the dense scan and
the head-row sampler, launched eagerly and replayed from a HIP graph (BatchedTSP.run(graph=True)).
usage: tools/b1_modes.py [iterations=400]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for n, A, k in ((500, 512, 50), (1000, 2048, 100)):
    c = torch.rand(1, n, 2, generator=g)
    d = (c[:, :, None] - c[:, None]).norm(dim=-1)
    i = torch.arange(n)
    d[:, i, i] = 1e9
    d = d.to(dev)
    for sampler in ("scan", "scan_sparse"):
        for graph in (False, True):
            col = engine.BatchedTSP(d, n_ants=A, seed=3, sampler=sampler)
            col.sparsify(k)
            col.heuristic = col.heuristic.contiguous()
            col.run(50, graph=graph)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            col.run(T, graph=graph)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / T
            print(json.dumps({"n": n, "ants": A, "sampler": sampler, "hip_graph": graph, "ms_per_iteration": round(dt * 1e3, 4),
                              "ant_tours_per_s": round(A / dt), "best": round(float(col.lowest_cost[0]), 4)}), flush=True)
