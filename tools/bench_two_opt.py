#!/usr/bin/env python3
"""Micro-benchmark of daco_two_opt: T tours of n nodes, random permutations, fixed sweep cap."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine
n, T, cap = int(sys.argv[1]) if len(sys.argv) > 1 else 500, int(sys.argv[2]) if len(sys.argv) > 2 else 4096, int(sys.argv[3]) if len(sys.argv) > 3 else 60
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
c = torch.rand(n, 2, generator=g)
d = torch.cdist(c, c); d[torch.arange(n), torch.arange(n)] = 1e9
d = d.to(dev)
rng = np.random.default_rng(0)
tours = torch.from_numpy(np.stack([rng.permutation(n) for _ in range(T)]).astype(np.int16)).to(dev)
for _ in range(2):
    t = tours.clone(); engine.two_opt_(d, t, cap)
torch.cuda.synchronize(); t0 = time.perf_counter()
t = tours.clone(); _, sw = engine.two_opt_(d, t, cap, want_sweeps=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
sweeps = int(sw.sum())
pairs = sweeps * (n - 1) * (n - 2) / 2
print(f"n={n} T={T} cap={cap}: {dt*1e3:.2f} ms, {sweeps} sweeps, {pairs/dt/1e9:.1f} G pair-evals/s, {sweeps/dt/1e6:.3f} M sweeps/s")
