#!/usr/bin/env python3
"""The tsp_nls training step three ways: eager on the parameter list (train_tsp_nls_batch as round 5 ran it),
eager on the flat block (Net.flatten_parameters), and as one captured HIP graph (pipeline.TspNlsTrainer) -- 20 instances of
TSP-100 x 30 ants (tsp_nls/train.py:95-100) and tsp/train.ipynb's TSP-500 x 50 ants (k = 50) with the NLS as local search.
usage: time_train_step.py [steps] [--shape 100|500|both]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd.pipeline import TspNlsTrainer, train_tsp_nls_batch  # noqa: E402
from deepaco_amd.tsp_nls.net import Net  # noqa: E402

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
which = sys.argv[sys.argv.index("--shape") + 1] if "--shape" in sys.argv else "both"
shapes = {"100": (20, 100, 30, 10), "500": (8, 500, 50, 50)}
out = {}
for key, (B, n, A, k) in shapes.items():
    if which not in (key, "both"):
        continue
    res = {"workload": f"tsp_nls training step, {B} instances x TSP-{n} x {A} ants (k = {k}), NLS"}
    batches = [torch.rand(B, n, 2, device=dev) for _ in range(4)]
    modes = tuple(os.environ.get("TRAIN_MODES", "eager_list,eager_flat,graph").split(","))
    for mode in modes:
        torch.manual_seed(0)
        net = Net().to(dev)
        if mode == "eager_list":
            opt = torch.optim.AdamW(net.parameters(), lr=3e-4)
            step = lambda c, s: train_tsp_nls_batch(net, opt, c, A, k, seed=1, it=s)      # noqa: E731
        else:
            tr = TspNlsTrainer(net, B, n, A, k, lr=3e-4, seed=1, graph=(mode == "graph"))
            step = lambda c, s: tr.step(c)                                                  # noqa: E731
        for s in range(4):
            step(batches[s % 4], s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(steps):
            loss, c, c_ls = step(batches[s % 4], 4 + s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res[mode] = {"ms_per_step": round(dt * 1e3, 4), "instances_per_s": round(B / dt, 1), "loss": float(loss),
                     "mean_cost": float(c), "mean_cost_nls": float(c_ls),
                     "parameters_finite": bool(all(torch.isfinite(p).all() for p in net.parameters()))}
    out[f"tsp{n}"] = res
    print(json.dumps(res), flush=True)
if os.environ.get("DACO_NLS_THREADS"):
    out["DACO_NLS_THREADS"] = os.environ["DACO_NLS_THREADS"]
