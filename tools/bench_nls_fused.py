#!/usr/bin/env python3
"""Config 3 (TSP-500 + NLS, 256 ants, B instances): one colony iteration with the fused NLS kernel (daco_tsp_nls) in its
thread / queue shapes against the pass-by-pass driver; prints ms per iteration, sweeps and list entries walked."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
n, A, B = 500, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = torch.Generator().manual_seed(2)
c = torch.rand(B, n, 2, generator=g)
d = (c[:, :, None] - c[:, None]).norm(dim=-1)      # (not cdist: its matmul form returns exact zeros for close points)
i = torch.arange(n)
d[:, i, i] = 1e9


def run(tag, env):
    for k in ("DACO_NLS_FUSED", "DACO_NLS_THREADS", "DACO_NLS_QUEUE", "DACO_NLS_GROUP", "DACO_NLS_PROFILE", "DACO_NLS_OWNER_BITS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    col = engine.BatchedTSP(d.to(dev), n_ants=A, seed=1, local_search="nls", fixed_start=0)
    col.sparsify(50)
    col.nls_counters = torch.zeros(2, dtype=torch.int64, device=dev)
    col.step()
    torch.cuda.synchronize()
    col.nls_counters.zero_()
    t0 = time.perf_counter()
    for _ in range(reps):
        col.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    cnt = col.nls_counters.tolist()
    print(json.dumps({"variant": tag, "ms_per_iteration": round(dt * 1e3, 2), "ant_tours_per_s": round(B * A / dt),
                      "sweeps_per_iteration": cnt[0] / reps, "entries_walked_per_sweep": (cnt[1] / cnt[0]) if cnt[0] else None,
                      "mean_best_cost": round(float(col.lowest_cost.mean()), 5)}), flush=True)


VARIANTS = {
    "old": ("pass-by-pass (round 2)", {"DACO_NLS_FUSED": "0"}),
    "g1": ("fused 256 threads, 1 entry per thread and round", {"DACO_NLS_THREADS": "256", "DACO_NLS_GROUP": "1"}),
    "g2": ("fused 256 threads, 2 entries per thread and round", {"DACO_NLS_THREADS": "256", "DACO_NLS_GROUP": "2"}),
    "g4": ("fused 256 threads, 4 entries per thread and round", {"DACO_NLS_THREADS": "256", "DACO_NLS_GROUP": "4"}),
    "g3": ("fused 192 threads, 3 entries per thread and round", {"DACO_NLS_GROUP": "3"}),
    "g3s": ("fused 192 threads, 3 entries, binary search for the entry's list (round 3's form)", {"DACO_NLS_OWNER_BITS": "0", "DACO_NLS_GROUP": "3"}),
    "g4_192": ("fused 192 threads, 4 entries per thread and round (default)", {}),
    "t256": ("fused 256 threads, 3 entries per thread and round", {"DACO_NLS_THREADS": "256", "DACO_NLS_GROUP": "3"}),
    "t512": ("fused 512 threads", {"DACO_NLS_THREADS": "512"}),
    "t1024": ("fused 1024 threads", {"DACO_NLS_THREADS": "1024"}),
    "prof": ("fused, profiled", {"DACO_NLS_PROFILE": "1"}),
}
for key in (sys.argv[3].split(",") if len(sys.argv) > 3 else VARIANTS):
    run(*VARIANTS[key])
