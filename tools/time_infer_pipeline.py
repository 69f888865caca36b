#!/usr/bin/env python3
"""Where the time of pipeline.infer_tsp_batch goes (default 64 x TSP-500, k = 50, 512 ants, the pretrained tsp500 network):
each stage timed with a device sync after it (ms; steady state = third call).  usage: time_infer_pipeline.py [B=64] [ants=512]
(1 50: one instance with the ant count of the reference's own harness, tsp/test.ipynb)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepaco_amd import engine  # noqa: E402
from deepaco_amd.tsp.net import Net  # noqa: E402

dev = torch.device("cuda:0")
wz = np.load(os.path.join(ROOT, "tests", "golden", "w_tsp_tsp500.npz"))
net = Net()
net.load_state_dict({k[3:]: torch.from_numpy(wz[k]) for k in wz.files}, strict=False)
net = net.to(dev).eval()
n, k = 500, 50
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
A = int(sys.argv[2]) if len(sys.argv) > 2 else 512


def stage(label, fn, acc):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    acc.setdefault(label, []).append((time.perf_counter() - t0) * 1e3)
    return out


acc = {}
for rep in range(3):
    coords = torch.rand(B, n, 2, device=dev)
    with torch.no_grad():
        dist, ei, ea = stage("kNN graph + distances", lambda: engine.tsp_knn_graph(coords, k), acc)
        heu = stage("network forward", lambda: net.forward_batch(coords, ei, ea, k_sparse=k), acc)
        mat = stage("reshape + eps", lambda: net.reshape_batch(n, ei, heu, eps=1e-10), acc)
        col = stage("colony set-up", lambda: engine.BatchedTSP(dist, n_ants=A, heuristic=mat, seed=rep), acc)
        stage("first iteration (sampler resolution, head rows)", lambda: col.run(1), acc)
        stage("19 more iterations", lambda: col.run(19), acc)
for kx, v in acc.items():
    print(f"{kx:50s} first {v[0]:9.3f}   steady {v[-1]:9.3f}")
print("sampler:", col.resolved_sampler(), f"B = {B}, ants = {A}")
