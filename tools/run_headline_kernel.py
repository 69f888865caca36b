#!/usr/bin/env python3
"""The headline launch alone, for profilers: tsp_scan32_kernel<4,false> at TSP-500 x 512 ants x 64 instances with the fused
tour lengths and neighbour table (what a BatchedTSP.step launches), `reps` times.  Prints the median HIP-event time.
usage: tools/run_headline_kernel.py [reps=8] [B=64] [A=512] [n=500] [mode=scan]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
A = int(sys.argv[3]) if len(sys.argv) > 3 else 512
n = int(sys.argv[4]) if len(sys.argv) > 4 else 500
mode = sys.argv[5] if len(sys.argv) > 5 else "scan"
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
c = torch.rand(B, n, 2, generator=g)
d = (c[:, :, None, :] - c[:, None, :, :]).norm(dim=-1)
i = torch.arange(n)
d[:, i, i] = 1e9
d = d.to(dev)
tau = torch.ones_like(d)
_, idx = torch.topk(d, k=max(5, n // 10), dim=2, largest=False)
sp = torch.full_like(d, 1e10)
sp.scatter_(2, idx, torch.gather(d, 2, idx))
eta = (1 / sp).contiguous()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for a, b in ev:
    a.record(); b.record()
torch.cuda.synchronize()
head = engine.sparse_head(eta, min(127, max(5, n // 10))) if mode in ("scan_sparse", "race_head") else None
stats = None
for r in range(reps):
    if mode in ("scan_sparse", "race_head"):
        plain = os.environ.get("SPARSE_PLAIN") == "1"
        out = engine.tsp_sample_sparse(tau, eta, A, head, seed=3, it=r, batch=B, events=ev[r], dist=None if plain else d,
                                       want_nbr=not plain, want_stats=(r == reps - 1), race=(mode == "race_head"))
        stats = out[4].cpu().tolist() if r == reps - 1 else stats
    else:
        engine.tsp_sample(tau, eta, A, mode=mode, seed=3, it=r, batch=B, events=ev[r], dist=d, want_nbr=True)
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) for a, b in ev[2:])
print(json.dumps({"n": n, "B": B, "A": A, "reps": reps, "kernel_ms_median": round(t[len(t) // 2], 4), "kernel_ms_min": round(t[0], 4),
                  "mode": mode, "sparse_stats": stats}))
