#!/usr/bin/env python3
"""One colony iteration of config 3 (TSP-500 + NLS, 256 ants, B instances; BatchedTSP(local_search='nls') with the
sparsified heuristic, maxt = n // 4) a few times, for rocprofv3 --kernel-trace --stats: which 2-opt kernel the time goes to."""
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
d = (c[:, :, None] - c[:, None]).norm(dim=-1)      # (not cdist: its matmul form returns exact zeros for close points)
i = torch.arange(n)
d[:, i, i] = 1e9
col = engine.BatchedTSP(d.to(dev), n_ants=A, seed=1, local_search="nls", fixed_start=0)
col.sparsify(50)
col.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 2
for _ in range(reps):
    col.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"workload": f"TSP-{n} + NLS iteration, {B} instances x {A} ants", "seconds_per_iteration": dt,
                  "ant_tours_per_s": B * A / dt, "mean_cost": float(col.lowest_cost.mean())}))
