#!/usr/bin/env python3
"""What bounds tsp_scan32_kernel?  Times the tour-construction kernel alone (HIP events around the launch)
under ablations of the headline workload (TSP-500, 32768 ants in flight):
  fused / plain      with / without the fused tour costs (random reads of the distance matrix) and the neighbour table
  B x A              64 x 512 (4 instances per XCD in flight) vs 8 x 4096 (one instance per XCD: every row an L2 hit)
Occupancy is capped from outside with DACO_SCAN32_LDS_PAD (read once per process).
usage: tools/ablate_scan32.py [n=500]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500


def inst(B):
    g = torch.Generator().manual_seed(1)
    c = torch.rand(B, n, 2, generator=g)
    d = torch.cdist(c, c)
    i = torch.arange(n)
    d[:, i, i] = 1e9
    return d.to(dev)


def kernel_ms(B, A, fused, reps=6):
    d = inst(B)
    tau = torch.ones_like(d)
    _, idx = torch.topk(d, k=max(5, n // 10), dim=2, largest=False)
    sp = torch.full_like(d, 1e10)
    sp.scatter_(2, idx, torch.gather(d, 2, idx))
    eta = (1 / sp).contiguous()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); b.record()
    torch.cuda.synchronize()
    for r in range(reps):
        engine.tsp_sample(tau, eta, A, seed=3, it=r, batch=B, events=ev[r], dist=d if fused else None, want_nbr=fused)
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev[2:])
    return t[len(t) // 2]


out = {"n": n, "lds_pad": int(os.environ.get("DACO_SCAN32_LDS_PAD", "0"))}
for B, A in ((64, 512), (8, 4096)):
    for fused in (True, False):
        out[f"B{B}_A{A}_{'fused' if fused else 'plain'}_ms"] = round(kernel_ms(B, A, fused), 4)
print(json.dumps(out), flush=True)
