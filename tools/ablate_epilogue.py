#!/usr/bin/env python3
"""What the construction kernel's epilogue costs at the headline shape (TSP-500 x 512 ants x 64 instances, k = 50): the same
launch of scan_sparse_kernel with every output, without the int64 paths, with the update's table only, with the paths only.
HIP events around the kernel itself (not the pre-pass).   usage: tools/ablate_epilogue.py [reps=12] [B=64] [A=512] [n=500]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
A = int(sys.argv[3]) if len(sys.argv) > 3 else 512
n = int(sys.argv[4]) if len(sys.argv) > 4 else 500
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
c = torch.rand(B, n, 2, generator=g)
d = (c[:, :, None, :] - c[:, None, :, :]).norm(dim=-1)
i = torch.arange(n)
d[:, i, i] = 1e9
d = d.to(dev)
k = max(5, n // 10)
tau = torch.ones_like(d)
_, idx = torch.topk(d, k=k, dim=2, largest=False)
sp = torch.full_like(d, 1e10)
sp.scatter_(2, idx, torch.gather(d, 2, idx))
eta = (1 / sp).contiguous()
head = engine.sparse_head(eta, min(127, k))
variants = {"paths+costs+table": dict(dist=d, want_nbr=True, want_paths=True),
            "costs+table": dict(dist=d, want_nbr=True, want_paths=False),
            "table": dict(dist=None, want_nbr=True, want_paths=False),
            "paths": dict(dist=None, want_nbr=False, want_paths=True),
            "paths+costs": dict(dist=d, want_nbr=False, want_paths=True)}
out = {}
for name, kw in variants.items():
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); b.record()
    torch.cuda.synchronize()
    for r in range(reps):
        engine.tsp_sample_sparse(tau, eta, A, head, seed=3, it=r, batch=B, events=ev[r], **kw)
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev[2:])
    out[name] = {"kernel_ms_median": round(t[len(t) // 2], 4), "kernel_ms_min": round(t[0], 4)}
print(json.dumps({"n": n, "B": B, "A": A, "k": k, "variants": out}))
