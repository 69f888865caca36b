#!/usr/bin/env python3
"""Randomised soak: scan-draw tours (all lane layouts, TSP and CVRP) against the CPU oracle on random sizes,
ant counts, seeds and value distributions (uniform, heavy-tailed, sparse with exact zeros, tiny).
usage: python tests/soak_parity.py [cases] [seed]   -- prints one line per mismatch and a summary
(test infrastructure: tests/test_gpu_soak.py runs a short soak in the GPU suite)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from deepaco_amd import engine  # noqa: E402

def run(cases, seed, save_failures=True):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    bad = 0
    for c in range(cases):
        bad += 0 if one_case(c, rng, dev, save_failures) else 1
    return bad


def one_case(c, rng, dev, save_failures):
    n = int(rng.choice([rng.integers(2, 65), rng.integers(65, 257), rng.integers(257, 513), rng.integers(513, 700)]))
    A = int(rng.integers(1, 40))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        P = rng.random((n, n)) + 1e-3
    elif kind == 1:
        P = np.exp(rng.normal(0, 6, (n, n)))
    elif kind == 2:
        P = np.exp(rng.uniform(-60, 10, (n, n)))
        P[rng.random((n, n)) < 0.4] = 0.0
        idx = np.arange(n)
        for off in (1, 2, 3, 5):
            P[idx, (idx + off) % n] = np.maximum(P[idx, (idx + off) % n], 1e-30)
    else:
        P = rng.random((n, n)) * 1e-38 + 1e-40
    P = P.astype(np.float32)
    seed, it = int(rng.integers(1, 2 ** 40)), int(rng.integers(0, 1000))
    wave = bool(rng.integers(0, 4) == 0)
    tau = torch.from_numpy(P)[None].contiguous().to(dev)
    eta = torch.ones(1, n, n, device=dev)
    cvrp = n >= 4 and rng.integers(0, 3) == 0
    if cvrp:
        demand = np.concatenate(([0.0], rng.integers(1, 10, n - 1))).astype(np.float32)
        cap = float(rng.integers(10, 60))
        paths, _, _, lens, flags = engine.cvrp_sample(tau, eta, torch.from_numpy(demand).to(dev), cap, A,
                                                      mode="scan_wave" if wave else "scan", seed=seed, it=it)
        rp, _, L = oracle.cvrp_sample_rng(P, demand, cap, A, "scan_wave" if wave else "scan", seed, it)
        if L < 0:
            ok = int(flags[0]) != 0
        else:
            ok = int(flags[0]) == 0 and L == int(lens.max()) and np.array_equal(paths[0, :L].cpu().numpy(), rp)
    else:
        paths, _, _, flags = engine.tsp_sample(tau, eta, A, mode="scan_wave" if wave else "scan", seed=seed, it=it)
        rp, _, rc = oracle.tsp_sample_scan(P, A, seed, it, wave=wave)
        if rc:                         # a draw without feasible candidate: flagged, and both move to node 0
            ok = int(flags[0]) == 1 and np.array_equal(paths[0].cpu().numpy(), rp)
        else:
            ok = int(flags[0]) == 0 and np.array_equal(paths[0].cpu().numpy(), rp)
    if not ok:
        if save_failures:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez(os.path.join(ROOT, "gpurun_out", f"soak_fail_{c}.npz"), P=P, A=A, seed=seed, it=it, wave=wave,
                     cvrp=cvrp, gpu_paths=paths[0].cpu().numpy(), flags=flags.cpu().numpy(),
                     ref_paths=rp if rp is not None else np.zeros(1))
        print(f"MISMATCH case {c}: n={n} A={A} kind={kind} wave={wave} cvrp={cvrp} seed={seed} it={it}", flush=True)
    return ok


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    t0 = time.time()
    n_bad = run(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{n_cases} cases, {n_bad} mismatches, {time.time() - t0:.1f} s")
    sys.exit(1 if n_bad else 0)
