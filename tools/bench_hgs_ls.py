#!/usr/bin/env python3
"""Throughput of the route-exact CVRP local search (daco_hgs_local_search) next to round 3's best-improvement kernel:
CVRP-n, A ants, B instances, solutions sampled by the colony, the schedule of cvrp_nls/aco.py:443-448."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--ants", type=int, default=512)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--cap", type=float, default=50.0)
    ap.add_argument("--limit", type=int, default=100)
    ap.add_argument("--old", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-short", action="store_true", help="only the three-stage launches (counter passes: one kernel shape)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n, A, B = args.n, args.ants, args.batch
    g = torch.Generator().manual_seed(3)
    loc = torch.cat((torch.full((B, 1, 2), 0.5, dtype=torch.double), torch.rand(B, n, 2, generator=g, dtype=torch.double)), 1)
    dem = torch.cat((torch.zeros(B, 1, dtype=torch.double), torch.randint(1, 10, (B, n), generator=g).double() / args.cap), 1).to(dev)
    d = (loc[:, :, None] - loc[:, None]).norm(dim=-1)
    ii = torch.arange(n + 1)
    d[:, ii, ii] = 1e-10
    d = d.to(dev)
    heu = 1 / d
    hd = 1 / (heu / heu.amax(dim=-1, keepdim=True) + 1e-5)
    col = engine.BatchedCVRP(d.float(), dem, n_ants=A, capacity=1.0, seed=1)
    paths, costs0 = col.step(trim=True)
    print(f"CVRP-{n}, {B} x {A} solutions, L = {paths.shape[1]}, mean sampled cost {float(costs0.mean()):.3f}", flush=True)
    t0 = time.perf_counter()
    td, th = engine.HgsTables(d), engine.HgsTables(hd)
    torch.cuda.synchronize()
    print(f"tables: {(time.perf_counter() - t0) * 1e3:.2f} ms (two matrices x {B} instances)")
    stages = [(td, args.limit), (th, 10), (td, args.limit)]
    for rep in range(args.reps):
        w = paths.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, status, stats = engine.hgs_local_search_(w, stages, dem, want_stats=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c1 = engine.tour_costs(d.float(), w, closed=False)
        print(f"hgs: {dt * 1e3:.2f} ms = {B * A / dt / 1e3:.1f} k solutions/s; moves/solution {float(stats[..., 0].float().mean()):.1f}, "
              f"loops {float(stats[..., 1].float().mean()):.2f}, rounds {float(stats[..., 2].float().mean()):.0f}, cost {float(c1.mean()):.4f}, status != 0: {int((status != 0).sum())}", flush=True)
    for cnt in (() if args.no_short else (0, 1)):
        w = paths.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        engine.hgs_local_search_(w, [(td, cnt)], dem)
        torch.cuda.synchronize()
        print(f"hgs one stage, count {cnt}: {(time.perf_counter() - t0) * 1e3:.2f} ms")
    if args.old:
        dls, hdl = d.float().contiguous(), hd.float().contiguous()
        for rep in range(2):
            w = paths.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for mtx, cnt in ((dls, 100000), (hdl, 10), (dls, 100000)):
                engine.cvrp_local_search_(mtx, dem.float(), 1.0, w, cnt)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            c1 = engine.tour_costs(dls, w, closed=False)
            print(f"best-improvement kernel: {dt * 1e3:.2f} ms = {B * A / dt / 1e3:.1f} k solutions/s, cost {float(c1.mean()):.4f}")


if __name__ == "__main__":
    main()
