#!/usr/bin/env python3
"""daco_tsp_nls under two settings of one knob, alternating in one process at the shapes the callers form
(tools/sweep_nls_threads.py); both must return the same tours.  Default knob: DACO_NLS_OWNER_BITS = 1 (the list of a flat entry
from the bitmap of list starts) against 0 (the binary search over the compacted lists).  Usage: ab_nls_owner_bits.py [reps [VAR
a,b]], e.g. `7 DACO_NLS_GROUP 4,3`.  Prints ms per call (HIP events, median of `reps`)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
VAR = sys.argv[2] if len(sys.argv) > 2 else "DACO_NLS_OWNER_BITS"      # the knob that alternates, and its two settings
MODES = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("1", "0")
SHAPES = [(100, 30, 20, 25), (500, 50, 8, 125), (200, 48, 1, 200), (500, 48, 1, 500), (1000, 48, 1, 1000), (500, 256, 16, 125),
          (500, 256, 64, 125), (1000, 64, 16, 250)]
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
    ts = {m: [] for m in MODES}
    ref = None
    for r in range(reps + 1):
        for mode in MODES:
            os.environ[VAR] = mode
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = engine.nls_(d, hd, tours, maxt, tables=td, heuristic_tables=th)
            e1.record()
            e1.synchronize()
            if ref is None:
                ref = out
            assert torch.equal(out, ref), (n, A, B, mode)
            if r:
                ts[mode].append(e0.elapsed_time(e1))
    for m in MODES:
        row[f"{VAR}={m}"] = round(sorted(ts[m])[reps // 2], 3)
    print(json.dumps(row), flush=True)
os.environ.pop(VAR, None)
