#!/usr/bin/env python3
"""Stage by stage at config 3's size (B instances): is every stage bit-identical run to run?"""
import os
import sys

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
d = d.to(dev)
col = engine.BatchedTSP(d, n_ants=A, seed=1, local_search="nls", fixed_start=0)
col.sparsify(50)
tau = torch.rand(B, n, n, device=dev) + 0.5


def sample():
    return engine.tsp_sample(tau, col.heuristic, A, mode="scan", seed=1, it=3, fixed_start=0, batch=B, dist=d, want_nbr=True)


s1, s2 = sample(), sample()
print("sampler paths/costs/nbr equal:", [bool(torch.equal(s1[k], s2[k])) for k in (0, 4, 5)])
tours = s1[0].permute(0, 2, 1).to(torch.int16).contiguous()
hd = col._heuristic_dist()
td, th = engine.TwoOptTables(d), engine.TwoOptTables(hd)
td2 = engine.TwoOptTables(d)
print("tables equal:", bool(torch.equal(td.tables, td2.tables)))
for fused in (False, True):
    a = engine.nls_(d, hd, tours, n // 4, tables=td, heuristic_tables=th, fused=fused, want_costs=True)
    b = engine.nls_(d, hd, tours, n // 4, tables=td, heuristic_tables=th, fused=fused, want_costs=True)
    print("nls fused=%s tours/costs equal run to run:" % fused, bool(torch.equal(a[0], b[0])), bool(torch.equal(a[1], b[1])))
    if fused:
        print("fused == pass-by-pass:", bool(torch.equal(a[0], keep[0])), bool(torch.equal(a[1], keep[1])),
              "tours differing:", int((a[0] != keep[0]).any(dim=2).sum()))
    keep = a
paths = keep[0].permute(0, 2, 1).to(torch.int64).contiguous()
t1, t2 = tau.clone(), tau.clone()
engine.pheromone_update_(t1, paths, keep[1], 0.9)
engine.pheromone_update_(t2, paths, keep[1], 0.9)
print("update equal:", bool(torch.equal(t1, t2)))
# ---- diagnostics
perm = (paths.sort(dim=1).values == torch.arange(n, device=dev).view(1, n, 1)).all(dim=1)
print("NLS output tours that are permutations:", int(perm.sum()), "of", perm.numel())
perm_s = (s1[0].sort(dim=1).values == torch.arange(n, device=dev).view(1, n, 1)).all(dim=1)
print("sampled tours that are permutations:", int(perm_s.sum()), "of", perm_s.numel())
nb1, nb2 = s1[5], s2[5]
print("nbr shape", tuple(nb1.shape), "differing entries", int((nb1 != nb2).sum()))
for fused in (False, True):
    outs = [engine.nls_(d, hd, tours, n // 4, tables=td, heuristic_tables=th, fused=fused) for _ in range(3)]
    diff = (outs[0] != outs[1]).any(dim=2)
    print("fused", fused, "differing tours per instance (runs 0/1):", diff.sum(dim=1).tolist())
    diff = (outs[0] != outs[2]).any(dim=2)
    print("fused", fused, "differing tours per instance (runs 0/2):", diff.sum(dim=1).tolist())
# only the first 2-opt pass
for kernel in ("nbr", "auto", "cached", None):
    r = []
    for _ in range(2):
        t = tours.clone()
        if kernel is None:
            engine.two_opt_(d, t, n // 4)
        else:
            engine.two_opt_(d, t, n // 4, tables=td, kernel=kernel)
        r.append(t)
    print("first pass, kernel", kernel, "equal run to run:", bool(torch.equal(r[0], r[1])), "differing tours", int((r[0] != r[1]).any(dim=2).sum()))
    if kernel == "nbr":
        base = r[0]
    else:
        print("   equal to nbr kernel:", bool(torch.equal(r[0], base)))
