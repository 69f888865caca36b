#!/usr/bin/env python3
"""Soak of the CVRP local-search kernel (daco_cvrp_local_search: nine move families, wave-parallel bookkeeping) against its
CPU restatement (oracle/cvrp_ls.py, an independent pure-Python implementation of the same specification): random small
instances -- Euclidean, integer-grid (ties), row-scaled and random asymmetric matrices (the reversal terms), tight and loose
capacities, normalised demands with exact fits, move caps -- sequences, lengths and move counts must be identical.
usage: tools/soak_cvrp_ls.py [seconds] [seed]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402
from oracle import cvrp_ls as ols  # noqa: E402

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260927
rng = np.random.default_rng(seed)
t_end = time.time() + budget
cases = sols = moves_total = 0
applied = []
kinds = {}
while time.time() < t_end:
    n = int(rng.integers(4, 34))                               # customers
    A = int(rng.integers(1, 7))
    kind = str(rng.choice(["euclid", "grid", "rowscaled", "asym"]))
    c = rng.random((n + 1, 2)).astype(np.float32)
    if kind == "grid":
        c = rng.integers(0, 6, size=(n + 1, 2)).astype(np.float32)
    d = np.sqrt(((c[:, None] - c[None]) ** 2).sum(-1)).astype(np.float32)
    if kind == "rowscaled":
        d = (d * rng.uniform(1.0, 50.0, size=(n + 1, 1))).astype(np.float32)
    elif kind == "asym":
        d = (rng.random((n + 1, n + 1)) * 10 ** rng.uniform(-1, 2)).astype(np.float32)
    np.fill_diagonal(d, 1e-10)
    if rng.random() < 0.5:                                    # integer demands, capacity 20..50
        dem = np.concatenate(([0], rng.integers(1, 10, size=n))).astype(np.float32)
        cap = float(rng.integers(max(10, int(dem.max())), 51))
    else:                                                     # normalised (k / cap, capacity 1): exact fits are common
        k = int(rng.choice([30, 40, 50]))
        dem = (np.concatenate(([0], rng.integers(1, 10, size=n))) / k).astype(np.float32)
        cap = 1.0
    dd, dm = torch.from_numpy(d).to(dev), torch.from_numpy(dem).to(dev)
    paths, _, _, lens, flags = engine.cvrp_sample(torch.ones_like(dd)[None], (1 / (dd + 1e-3))[None], dm, cap, A,
                                                  seed=int(rng.integers(1 << 30)))
    if int(flags.sum()) != 0:
        continue
    paths = paths[:, :int(lens.max()) + 2].contiguous()
    maxm = int(rng.choice([1, 3, 10, 2000]))                   # (2000: never reached -- the search ends by itself; bounds a runaway)
    before = paths.clone()
    out, ln, nm = engine.cvrp_local_search_(dd, dm, cap, paths, maxm, want_stats=True)
    for a in range(A):
        ref, ref_moves = ols.local_search(before[0, :, a].cpu().numpy(), d, dem, cap, maxm, kinds=applied)
        got = out[0, :, a].cpu().numpy()
        if not (int(ln[0, a]) == len(ref) and int(nm[0, a]) == ref_moves and np.array_equal(got[:len(ref)], np.array(ref))
                and not got[len(ref):].any()):
            print(json.dumps({"MISMATCH": True, "case": cases, "n": n, "A": A, "ant": a, "kind": kind, "cap": cap, "maxm": maxm, "seed": seed}))
            sys.exit(1)
        moves_total += ref_moves
    cases += 1
    if cases % 50 == 0:
        print(json.dumps({"progress": cases, "moves": moves_total}), flush=True)
    sols += A
    kinds[kind] = kinds.get(kind, 0) + 1
print(json.dumps({"soak": "cvrp_ls_kernel == oracle/cvrp_ls.py (sequences, lengths, move counts)", "seconds": budget, "seed": seed,
                  "cases": cases, "solutions": sols, "moves": moves_total,
                  "moves_by_kind": {ols.KINDS[k]: applied.count(k) for k in range(len(ols.KINDS))}, "kinds": kinds, "mismatches": 0}))
