#!/usr/bin/env python3
"""Soak of the candidate-list 2-opt kernels -- per-sweep lists (daco_two_opt_nbr), the hand-over driver (daco_two_opt_auto) and
the dirty-list kernel (daco_tsp_nls without rounds, every thread / group shape) -- against the dense incremental kernel (which
the suite pins on the oracle and on the reference's golden vectors): random sizes, matrix kinds, tour sources, sweep caps and
hand-over thresholds for a given number of seconds; tours and sweep counts must be identical.  Every fourth case also runs
the whole NLS schedule (random T_nls, T_p, a second random matrix as perturbation matrix) fused in one launch against the
pass-by-pass driver over the dense kernels.  usage: tools/soak_two_opt.py [seconds] [seed]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260927
rng = np.random.default_rng(seed)
t_end = time.time() + budget
cases = tours_total = sweeps_total = nls_cases = 0
kinds = {}
while time.time() < t_end:
    n = int(rng.choice([rng.integers(4, 40), rng.integers(40, 300), rng.integers(300, 700), rng.integers(700, 1025)],
                       p=[0.25, 0.45, 0.25, 0.05]))
    B = int(rng.integers(1, 4))
    Tn = int(rng.choice([1, 3, 16, 64])) if n <= 300 else int(rng.choice([1, 4, 16]))
    kind = str(rng.choice(["euclid", "grid", "rowscaled", "plateau", "asym", "signed"]))
    c = rng.random((B, n, 2)).astype(np.float32)
    if kind == "grid":
        c = rng.integers(0, 8, size=(B, n, 2)).astype(np.float32)
    d = np.sqrt(((c[:, :, None] - c[:, None]) ** 2).sum(-1)).astype(np.float32)
    if kind == "rowscaled":
        d = (d * rng.uniform(1.0, 300.0, size=(B, n, 1))).astype(np.float32)
    elif kind == "plateau":                                   # 1 / (h / rowmax + 1e-5) with h on a k-NN graph only
        k = max(2, n // 10)
        idx = np.argsort(d + np.eye(n, dtype=np.float32) * 1e9, axis=2)[:, :, :k]
        h = np.zeros_like(d)
        np.put_along_axis(h, idx, rng.random((B, n, k)).astype(np.float32) + 0.05, 2)
        d = (1 / (h / h.max(-1, keepdims=True) + 1e-5)).astype(np.float32)
    elif kind == "asym":
        d = (rng.random((B, n, n)) * 10 ** rng.uniform(-2, 4)).astype(np.float32)
    elif kind == "signed":
        d = rng.uniform(-1, 1, size=(B, n, n)).astype(np.float32)
    i = np.arange(n)
    d[:, i, i] = 0.0 if kind == "signed" else 1e9
    dd = torch.from_numpy(d).to(dev)
    if rng.random() < 0.5 and kind in ("euclid", "grid") and n >= 8:      # tours as the colony samples them
        eta = 1 / (dd + 1e-2)                                 # (grid: coincident points have distance 0)
        paths, _, _, _ = engine.tsp_sample(torch.ones_like(dd), eta, Tn, mode="scan", seed=int(rng.integers(1 << 30)), fixed_start=0)
        tours = paths.permute(0, 2, 1).to(torch.int16).contiguous()
    else:
        tours = torch.from_numpy(np.stack([[rng.permutation(n) for _ in range(Tn)] for _ in range(B)]).astype(np.int16)).to(dev)
    assert bool((tours.long().sort(dim=2).values == torch.arange(n, device=dev)).all()), "input tours must be permutations"
    maxit = int(rng.choice([1, 5, 20, n // 4 + 1, 10000])) if kind not in ("asym", "signed", "plateau") else int(rng.choice([1, 5, 20, 60]))
    tabs = engine.TwoOptTables(dd)
    ref, rs = engine.two_opt_(dd, tours.clone(), maxit, want_sweeps=True)
    for kernel in ("nbr", "auto", "cached"):
        os.environ["DACO_TWO_OPT_WIDE"] = str(int(rng.integers(0, 2)))
        os.environ["DACO_NLS_THREADS"] = str(rng.choice([192, 256, 512, 1024]))
        os.environ["DACO_NLS_GROUP"] = str(rng.choice([1, 2, 3, 4]))
        if kernel == "auto" and rng.random() < 0.5:
            sw = int(rng.integers(1, n * n))
            os.environ["DACO_TWO_OPT_SWITCH"], os.environ["DACO_TWO_OPT_BACK"] = str(sw), str(int(rng.integers(0, sw + 1)))
        try:
            out, so = engine.two_opt_(dd, tours.clone(), maxit, want_sweeps=True, tables=tabs, kernel=kernel)
        finally:
            for k_ in ("DACO_TWO_OPT_WIDE", "DACO_TWO_OPT_SWITCH", "DACO_TWO_OPT_BACK", "DACO_NLS_THREADS", "DACO_NLS_GROUP"):
                os.environ.pop(k_, None)
        if not (torch.equal(out, ref) and torch.equal(so, rs)):
            print(json.dumps({"MISMATCH": True, "case": cases, "n": n, "B": B, "T": Tn, "kind": kind, "maxit": maxit, "kernel": kernel, "seed": seed}))
            sys.exit(1)
    if cases % 4 == 0:
        # the whole NLS: a second matrix of the same family as perturbation matrix (a row-scaled image of the first: asymmetric)
        hd = (dd * torch.from_numpy(rng.uniform(0.5, 2.0, size=(B, n, 1)).astype(np.float32)).to(dev)).contiguous()
        hd[:, torch.arange(n), torch.arange(n)] = dd[:, torch.arange(n), torch.arange(n)]
        th = engine.TwoOptTables(hd)
        T_nls, T_p = int(rng.integers(0, 4)), int(rng.integers(1, 8))
        cap = int(min(maxit, 200))
        os.environ["DACO_NLS_THREADS"] = str(rng.choice([192, 256, 512, 1024]))
        os.environ["DACO_NLS_GROUP"] = str(rng.choice([1, 2, 3, 4]))
        try:
            a = engine.nls_(dd, hd, tours, cap, T_nls=T_nls, T_p=T_p, tables=tabs, heuristic_tables=th, fused=False, want_costs=True)
            b = engine.nls_(dd, hd, tours, cap, T_nls=T_nls, T_p=T_p, tables=tabs, heuristic_tables=th, fused=True, want_costs=True)
        finally:
            os.environ.pop("DACO_NLS_THREADS", None); os.environ.pop("DACO_NLS_GROUP", None)
        if not (torch.equal(a[0], b[0]) and torch.equal(a[1].view(torch.int32), b[1].view(torch.int32))):
            print(json.dumps({"MISMATCH": True, "case": cases, "n": n, "B": B, "T": Tn, "kind": kind, "maxit": cap, "kernel": "fused nls",
                              "T_nls": T_nls, "T_p": T_p, "seed": seed}))
            sys.exit(1)
        nls_cases += 1
    cases += 1
    tours_total += B * Tn
    sweeps_total += int(rs.sum())
    kinds[kind] = kinds.get(kind, 0) + 1
print(json.dumps({"soak": "two_opt candidate-list / hand-over / dirty-list kernels == dense incremental kernel; fused NLS == pass-by-pass",
                  "seconds": budget, "seed": seed, "cases": cases, "nls_schedules": nls_cases, "tours": tours_total, "sweeps": sweeps_total, "kinds": kinds, "mismatches": 0}))
