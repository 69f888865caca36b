#!/usr/bin/env python3
"""ms per ACO iteration at the reference's own call patterns (one instance per colony, its harness' ant counts): the drop-in classes
tsp.ACO (tsp/test.ipynb: TSP-20 / 100 / 500 with 20 / 20 / 50 ants), tsp_nls.ACO (tsp_nls/test.py: TSP-200 / 500 / 1000, 48 ants... NLS)
and cvrp.ACO (cvrp/test.py: CVRP-20 / 100 / 500, 20 ants), heuristic = the sparsified 1/d (no network: the colony loop alone)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

dev = "cuda:0"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def timed(make, iters):
    aco = make()
    aco.run(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    aco.run(iters)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3, float(aco.lowest_cost)


out = []
from deepaco_amd.tsp.aco import ACO as TspACO  # noqa: E402
for n, A, k in ((20, 20, 10), (100, 20, 20), (500, 50, 50), (1000, 50, 100)):
    g = torch.Generator().manual_seed(n)
    c = torch.rand(n, 2, generator=g)
    d = (c[:, None] - c).norm(dim=-1)
    d[torch.arange(n), torch.arange(n)] = 1e9

    def make():
        a = TspACO(d.to(dev), n_ants=A, device=dev)
        a.sparsify(k)
        return a
    ms, best = timed(make, T)
    out.append({"class": "tsp.ACO", "n": n, "ants": A, "k": k, "ms_per_iteration": round(ms, 4), "best": round(best, 4)})
    print(json.dumps(out[-1]), flush=True)
from deepaco_amd.tsp_nls.aco import ACO as NlsACO  # noqa: E402
for n, A in ((200, 48), (500, 48), (1000, 48)):
    g = torch.Generator().manual_seed(n + 1)
    c = torch.rand(n, 2, generator=g)
    d = (c[:, None] - c).norm(dim=-1)
    d[torch.arange(n), torch.arange(n)] = 1e9

    def make():
        a = NlsACO(d.to(dev), n_ants=A, device=dev, local_search="nls")
        a.sparsify(n // 10)
        return a
    ms, best = timed(make, max(3, T // 10))
    out.append({"class": "tsp_nls.ACO (inference NLS)", "n": n, "ants": A, "ms_per_iteration": round(ms, 4), "best": round(best, 4)})
    print(json.dumps(out[-1]), flush=True)
from deepaco_amd.cvrp.aco import ACO as CvrpACO  # noqa: E402
for n, A in ((20, 20), (100, 20), (500, 20)):
    g = torch.Generator().manual_seed(n + 2)
    loc = torch.cat((torch.full((1, 2), 0.5), torch.rand(n, 2, generator=g)))
    dem = torch.cat((torch.zeros(1), torch.randint(1, 10, (n,), generator=g).float()))
    d = (loc[:, None] - loc).norm(dim=-1)
    d[torch.arange(n + 1), torch.arange(n + 1)] = 1e-10

    def make():
        return CvrpACO(d.to(dev), dem.to(dev), n_ants=A, device=dev, capacity=50)
    ms, best = timed(make, T)
    out.append({"class": "cvrp.ACO", "n": n, "ants": A, "ms_per_iteration": round(ms, 4), "best": round(best, 4)})
    print(json.dumps(out[-1]), flush=True)
