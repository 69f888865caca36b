#!/usr/bin/env python3
"""Experiment: does a spatially sorted node numbering (Morton order of the coordinates) speed the fused NLS up
(a list's matrix gathers then share cache lines)?  Same instances, relabelled; prints ms per iteration and the mean best cost
(which must not change: 2-opt ties break on tour positions, not on node ids)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
n, A, B = 500, 256, 64
g = torch.Generator().manual_seed(2)
c = torch.rand(B, n, 2, generator=g)


def morton(xy, bits=10):
    q = (xy * (1 << bits)).long().clamp_(0, (1 << bits) - 1)
    key = torch.zeros(xy.shape[:-1], dtype=torch.long)
    for b in range(bits):
        key |= ((q[..., 0] >> b) & 1) << (2 * b)
        key |= ((q[..., 1] >> b) & 1) << (2 * b + 1)
    return key


for tag in ("original", "morton", "random"):
    if tag == "original":
        cc = c
    elif tag == "morton":
        order = morton(c).argsort(dim=1)
        cc = torch.gather(c, 1, order[..., None].expand(-1, -1, 2))
    else:
        order = torch.stack([torch.randperm(n, generator=g) for _ in range(B)])
        cc = torch.gather(c, 1, order[..., None].expand(-1, -1, 2))
    d = (cc[:, :, None] - cc[:, None]).norm(dim=-1)
    i = torch.arange(n)
    d[:, i, i] = 1e9
    col = engine.BatchedTSP(d.to(dev), n_ants=A, seed=1, local_search="nls", fixed_start=0)
    col.sparsify(50)
    col.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        col.step()
    torch.cuda.synchronize()
    print(json.dumps({"labelling": tag, "ms_per_iteration": round((time.perf_counter() - t0) / 3 * 1e3, 2),
                      "mean_best_cost": round(float(col.lowest_cost.mean()), 5)}), flush=True)
