#!/usr/bin/env python3
"""Does the headline iteration gain from running its instances as independent colonies on several HIP streams (the small kernels of one
colony -- deposit, prob_matrix, track_best: 10 % of an iteration -- under another colony's construction kernel)?
usage: tools/two_stream_experiment.py [steps=300]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
n, A, B, k = 500, 512, 64, 50
g = torch.Generator().manual_seed(1)
c = torch.rand(B, n, 2, generator=g)
d = (c[:, :, None, :] - c[:, None, :, :]).norm(dim=-1)
i = torch.arange(n)
d[:, i, i] = 1e9
d = d.to(dev)


def colonies(parts, sampler):
    cols = []
    per = B // parts
    for p in range(parts):
        col = engine.BatchedTSP(d[p * per:(p + 1) * per].contiguous(), n_ants=A, seed=5, sampler=sampler, ant_gid0=p * per * A)
        col.sparsify(k)
        col.heuristic = col.heuristic.contiguous()
        cols.append(col)
    return cols


for sampler in ("scan", "scan_sparse"):
    for parts in (1, 2, 4):
        cols = colonies(parts, sampler)
        streams = [torch.cuda.Stream(device=dev) for _ in cols]
        for s in streams:
            s.wait_stream(torch.cuda.current_stream(dev))

        def run(nsteps):
            for _ in range(nsteps):
                for col, s in zip(cols, streams):
                    with torch.cuda.stream(s):
                        col.step()
        run(100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(json.dumps({"sampler": sampler, "colonies_x_streams": parts, "ms_per_iteration_of_64_instances": round(dt * 1e3, 4),
                          "ant_tours_per_s": round(B * A / dt), "mean_best": round(float(torch.cat([c_.lowest_cost for c_ in cols]).mean()), 4)}), flush=True)
        del cols
