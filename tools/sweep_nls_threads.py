#!/usr/bin/env python3
"""daco_tsp_nls alone (engine.nls_ with the tables built beforehand) over threads per tour, at the tour counts the callers form:
the training steps (20 x 30 tours of TSP-100, 8 x 50 of TSP-500), one instance of the inference harness (tsp_nls/test.py: one
colony's ants), a few instances, config 3.  Prints ms per call (HIP events, median of `reps`) per (shape, threads)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
SHAPES = [(100, 30, 20, 25), (500, 50, 8, 125), (200, 48, 1, 200), (500, 48, 1, 500), (1000, 48, 1, 1000), (500, 256, 4, 125),
          (500, 256, 16, 125), (500, 256, 64, 125)]
THREADS = ["default", "64", "128", "192", "256", "512", "1024"]
for n, A, B, maxt in SHAPES:
    g = torch.Generator().manual_seed(n + A)
    c = torch.rand(B, n, 2, generator=g)
    d = (c[:, :, None] - c[:, None]).norm(dim=-1)
    i = torch.arange(n)
    d[:, i, i] = 1e9
    d = d.to(dev)
    k = max(5, n // 10)
    _, idx = torch.topk(d, k=k, dim=2, largest=False)
    eta = torch.full_like(d, 1e-10).scatter_(2, idx, 1 / torch.gather(d, 2, idx))
    paths, _, _, _ = engine.tsp_sample(torch.ones_like(d), eta, A, mode="scan", seed=3, fixed_start=0)
    tours = paths.permute(0, 2, 1).to(torch.int16).contiguous()
    hd = (1 / (eta / eta.amax(dim=-1, keepdim=True) + 1e-5)).contiguous()
    td, th = engine.TwoOptTables(d), engine.TwoOptTables(hd)
    row = {"n": n, "tours_per_instance": A, "instances": B, "maxt": maxt}
    ref = None
    for nt in THREADS:
        if nt == "default":
            os.environ.pop("DACO_NLS_THREADS", None)
        else:
            if (nt == "64" and n + 1 > 128) or (nt == "128" and n + 1 > 256) or (nt == "192" and n + 1 > 576):
                continue
            os.environ["DACO_NLS_THREADS"] = nt
        out = engine.nls_(d, hd, tours, maxt, tables=td, heuristic_tables=th)
        if ref is None:
            ref = out
        assert torch.equal(out, ref), (n, A, B, nt)
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            engine.nls_(d, hd, tours, maxt, tables=td, heuristic_tables=th)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        row[nt] = round(sorted(ts)[len(ts) // 2], 3)
    print(json.dumps(row), flush=True)
