#!/usr/bin/env python3
"""The headline iteration with and without the int64 paths tensor (BatchedTSP.step(want_paths=...)), alternating blocks on one
box: ms per iteration and the construction kernel's own duration (HIP events around the launch)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
n, A, B, k = 500, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 50
g = torch.Generator().manual_seed(1234)
c = torch.rand(B, n, 2, generator=g)
d = (c[:, :, None] - c[:, None]).norm(dim=-1)
i = torch.arange(n)
d[:, i, i] = 1e9
cols = {}
for tag in ("int64_paths", "compact"):
    col = engine.BatchedTSP(d.to(dev), n_ants=A, seed=1234, sampler="auto")
    col.sparsify(k)
    col.heuristic = col.heuristic.contiguous()
    cols[tag] = col
    for _ in range(10):
        col.step(want_paths=(tag == "int64_paths"))
torch.cuda.synchronize()
res = {t: {"ms": [], "kernel_ms": []} for t in cols}
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
for a, b in ev:
    a.record(); b.record()
torch.cuda.synchronize()
for rep in range(6):
    for tag, col in cols.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(50):
            col.step(events=ev[s], want_paths=(tag == "int64_paths"))
        torch.cuda.synchronize()
        res[tag]["ms"].append((time.perf_counter() - t0) / 50 * 1e3)
        res[tag]["kernel_ms"].append(sum(a.elapsed_time(b) for a, b in ev) / 50)
out = {"workload": f"TSP-{n} x {A} ants x {B} instances, k = {k}, sampler auto"}
for tag, r in res.items():
    ms, km = sorted(r["ms"]), sorted(r["kernel_ms"])
    out[tag] = {"ms_per_iteration_median": round(ms[len(ms) // 2], 4), "ms_per_iteration_min": round(ms[0], 4),
                "kernel_ms_median": round(km[len(km) // 2], 4), "ant_tours_per_s": round(B * A / (ms[len(ms) // 2] * 1e-3))}
out["same_best_costs"] = bool(torch.equal(cols["int64_paths"].lowest_cost, cols["compact"].lowest_cost))
print(json.dumps(out))
