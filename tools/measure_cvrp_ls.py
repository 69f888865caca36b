#!/usr/bin/env python3
"""CVRP local search (daco_cvrp_local_search) at config 4's scale: CVRP-100, 512 ants, B instances, the schedule of
cvrp_nls/aco.py:443-448 as deepaco_amd/cvrp_nls/aco.py runs it (search on the distances until no move improves, 10 moves on the
heuristic-derived matrix, search on the distances again)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
n, A, B = 100, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator().manual_seed(3)
loc = torch.cat((torch.full((B, 1, 2), 0.5), torch.rand(B, n, 2, generator=g)), 1)
dem = torch.cat((torch.zeros(B, 1), torch.randint(1, 10, (B, n), generator=g).float()), 1).to(dev)
d = (loc[:, :, None] - loc[:, None]).norm(dim=-1)
i = torch.arange(n + 1)
d[:, i, i] = 1e-10
d = d.to(dev)
heu = 1 / d
hd = (1 / (heu / heu.amax(dim=-1, keepdim=True) + 1e-5)).contiguous()
col = engine.BatchedCVRP(d, dem, n_ants=A, capacity=50, seed=1)
paths, costs = col.step(trim=True)
limit = 100000


moves = []


def search(p):
    moves.clear()
    for m, cnt in ((d, limit), (hd, 10), (d, limit)):
        _, _, mv = engine.cvrp_local_search_(m, dem, 50.0, p, cnt, want_stats=True)
        moves.append(mv)


w = paths.clone(); search(w)
torch.cuda.synchronize()
w = paths.clone()
t0 = time.perf_counter()
search(w)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
c1 = engine.tour_costs(d, w, closed=False)
nm = float(sum(m.float().sum() for m in moves))
L = float(col.last_lens.float().mean())
# a move evaluates every (i, j) of the sequence for nine move families: ~ L^2 pairs, up to 9 candidates each
print(json.dumps({"workload": f"CVRP-{n} local search, {B} x {A} solutions, to convergence + 10 perturbation moves + to convergence",
                  "seconds": dt, "solutions_per_s": B * A / dt, "moves_per_solution": nm / (B * A), "moves_per_s": nm / dt,
                  "pairs_evaluated_per_s": nm * L * L / dt, "mean_sequence_length": L,
                  "mean_cost_before": float(costs.mean()), "mean_cost_after": float(c1.mean())}))
