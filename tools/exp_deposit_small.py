#!/usr/bin/env python3
"""Where the pheromone deposit's time goes at n = 100 (256 instances x 512 ants): the full AS deposit against the elitist one (one
ant per instance: the same loads and stores, a chain of one add).  Run under rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
n, A, B = 100, 512, 256
g = torch.Generator().manual_seed(1)
c = torch.rand(B, n, 2, generator=g)
d = (c[:, :, None] - c[:, None]).norm(dim=-1)
d[:, torch.arange(n), torch.arange(n)] = 1e9
for elitist in (False, True):
    col = engine.BatchedTSP(d.to(dev), n_ants=A, elitist=elitist, seed=3)
    for _ in range(12):
        col.step()
    torch.cuda.synchronize()
