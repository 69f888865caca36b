#!/usr/bin/env python3
"""As tools/run_train_step.py at the size where the GNN backward dominates: 8 instances of TSP-500 x 30 ants, 2-opt."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd.pipeline import train_tsp_nls_batch  # noqa: E402
from deepaco_amd.tsp_nls.net import Net  # noqa: E402

dev = torch.device("cuda:0")
B, n, A, k = 8, 500, 30, 50
torch.manual_seed(0)
net = Net().to(dev)
opt = torch.optim.AdamW(net.parameters(), lr=3e-4)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for s in range(2):
    train_tsp_nls_batch(net, opt, torch.rand(B, n, 2, device=dev), A, k, seed=1, it=s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(steps):
    loss, c, c_ls = train_tsp_nls_batch(net, opt, torch.rand(B, n, 2, device=dev), A, k, seed=1, it=2 + s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"workload": f"tsp_nls training step, {B} instances x TSP-{n} x {A} ants, NLS", "ms_per_step": dt * 1e3,
                  "instances_per_s": B / dt, "loss": float(loss), "mean_cost": float(c), "mean_cost_nls": float(c_ls)}))
