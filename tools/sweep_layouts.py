#!/usr/bin/env python3
"""Tour-construction time per layout over n: 'scan' (4 / 2 / 1 ants per wavefront by the n rule) vs
'scan_wave' (always one ant per wavefront).  usage: tools/sweep_layouts.py [n ...]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
A = 512
for n in [int(x) for x in sys.argv[1:]] or [32, 64, 65, 100, 128, 129, 200, 256, 384, 512, 768, 1024]:
    B = max(1, min(256, (64 * 500 * 500) // (n * n)))
    c = torch.rand(B, n, 2, device=dev)
    d = torch.cdist(c, c)
    i = torch.arange(n)
    d[:, i, i] = 1e9
    eta, tau = (1 / d).contiguous(), torch.ones_like(d)
    out = {"n": n, "B": B, "A": A}
    for mode in ("scan", "scan_wave"):
        for _ in range(2):
            engine.tsp_sample(tau, eta, A, mode=mode, seed=1, dist=d, want_nbr=True, batch=B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(5):
            engine.tsp_sample(tau, eta, A, mode=mode, seed=1, it=it, dist=d, want_nbr=True, batch=B)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        out[mode + "_ms"] = round(dt * 1e3, 4)
        out[mode + "_Mtours_s"] = round(B * A / dt / 1e6, 2)
    print(json.dumps(out), flush=True)
