#!/usr/bin/env python3
"""Time BASELINE.json's single-GPU configurations (2, 3, 4 and the headline) end to end.
Prints one JSON object per configuration; run under rocprofv3 --kernel-trace --stats for the
per-kernel split.  usage: tools/measure_configs.py [cfg ...]   cfg in {headline, c2, c3, c4, gnn}"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

dev = torch.device("cuda:0")


def tsp_instances(B, n, seed):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(B, n, 2, generator=g)
    d = (c[:, :, None] - c[:, None]).norm(dim=-1)            # (not cdist: its matmul form returns exact zeros for close points)
    i = torch.arange(n)
    d[:, i, i] = 1e9
    return d.to(dev)


def timeit(fn, steps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def run(cfg):
    if cfg in ("headline", "c2", "c5shard"):
        n, A, B, k = {"headline": (500, 512, 64, 50), "c2": (100, 512, 256, 20), "c5shard": (1000, 2048, 64, 100)}[cfg]
        col = engine.BatchedTSP(tsp_instances(B, n, 1), n_ants=A, seed=1, sampler="scan")   # (the dense scan: these passes are of tsp_scan32 / scan16)
        col.sparsify(k)
        dt = timeit(col.step, 3 if n >= 1000 else 10)
        return dict(config=cfg, desc=f"TSP-{n}, {A} ants, {B} instances, AS iteration", ms_per_iteration=dt * 1e3,
                    ant_tours_per_s=B * A / dt)
    if cfg.startswith("c3"):
        n, A, B = (int(cfg.split(":")[1]) if ":" in cfg else 500), 256, 64
        col = engine.BatchedTSP(tsp_instances(B, n, 2), n_ants=A, seed=1, local_search="nls", fixed_start=0)
        col.sparsify(max(5, n // 10))
        dt = timeit(col.step, 2, warm=1)
        return dict(config=cfg, desc=f"TSP-{n} + NLS (2-opt kernel, T_nls=10, T_p=20, maxt={n//4}), {A} ants, {B} instances",
                    ms_per_iteration=dt * 1e3, ant_tours_per_s=B * A / dt)
    if cfg == "c4":
        n, A, B = 100, 512, 256
        g = torch.Generator().manual_seed(3)
        loc = torch.cat((torch.full((B, 1, 2), 0.5), torch.rand(B, n, 2, generator=g)), 1)
        dem = torch.cat((torch.zeros(B, 1), torch.randint(1, 10, (B, n), generator=g).float()), 1)
        d = (loc[:, :, None] - loc[:, None]).norm(dim=-1)
        i = torch.arange(n + 1)
        d[:, i, i] = 1e-10
        col = engine.BatchedCVRP(d.to(dev), dem.to(dev), n_ants=A, capacity=50, seed=1)
        dt = timeit(col.step, 10)
        return dict(config=cfg, desc=f"CVRP-{n} (capacity mask), {A} ants, {B} instances, AS iteration",
                    ms_per_iteration=dt * 1e3, ant_tours_per_s=B * A / dt)
    if cfg == "gnn":
        from deepaco_amd.tsp.net import Net
        from deepaco_amd.tsp.utils import gen_pyg_data
        torch.manual_seed(0)
        net = Net().to(dev).eval()
        out = []
        for n, k in ((100, 20), (500, 50), (1000, 100)):
            pyg, _ = gen_pyg_data(torch.rand(n, 2, device=dev), k)
            with torch.no_grad():
                t_hip = timeit(lambda: net(pyg), 20)
                t_torch = timeit(lambda: net.par_net_heu(net.emb_net(pyg.x, pyg.edge_index, pyg.edge_attr)), 5)
            B = 64
            coords = torch.rand(B, n, 2, device=dev)
            _, ei, ea = engine.tsp_knn_graph(coords, k, want_dist=False)
            with torch.no_grad():
                t_batch = timeit(lambda: net.forward_batch(coords, ei, ea, k_sparse=k), 5)
            out.append(dict(n=n, k=k, E=n * k, hip_ms=t_hip * 1e3, torch_ops_ms=t_torch * 1e3,
                            hip_batch64_ms=t_batch * 1e3, hip_batch64_ms_per_graph=t_batch * 1e3 / B))
        return dict(config=cfg, desc="Net.forward eval, one instance (HIP kernels vs torch ops on the same GPU)", sizes=out)
    raise SystemExit(f"unknown config {cfg}")


if __name__ == "__main__":
    for c in (sys.argv[1:] or ["headline", "c2", "c3", "c4", "gnn"]):
        print(json.dumps(run(c)), flush=True)
