#!/usr/bin/env python3
"""Candidate-list 2-opt kernel (daco_two_opt_nbr) against the dense incremental kernel on the three kinds of 2-opt call the
NLS makes at config 3's sizes (TSP-500, 256 ants per instance): (1) the first pass on freshly sampled tours (many long
edges: the candidate lists are long), (2) 20 perturbation sweeps on the heuristic-derived matrix, (3) the repair pass."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n, A = 500, 256
g = torch.Generator().manual_seed(2)
c = torch.rand(B, n, 2, generator=g)
d = torch.cdist(c, c)
i = torch.arange(n)
d[:, i, i] = 1e9
d = d.to(dev)
d = ((d + d.transpose(1, 2)) / 2).contiguous()          # (cdist is not bit-symmetric; the reference's norm-based matrix is)
eta = 1 / d
paths, _, _, _ = engine.tsp_sample(torch.ones_like(d), eta, A, mode="scan", seed=4, fixed_start=0)
tours = paths.permute(0, 2, 1).to(torch.int16).contiguous()
hd = (1 / (eta / eta.amax(dim=-1, keepdim=True) + 1e-5)).contiguous()
hdt = engine.transposed_for_two_opt(hd)


def timed(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return r, time.perf_counter() - t0


(td, th), t_prep = timed(lambda: (engine.TwoOptTables(d), engine.TwoOptTables(hd, hdt)))
(td, th), t_prep = timed(lambda: (engine.TwoOptTables(d), engine.TwoOptTables(hd, hdt)))
out = {"instances": B, "tours": B * A, "prepare_ms_both_matrices": t_prep * 1e3}
cur = tours
dsym = engine.transposed_for_two_opt(d)
for name, m, mt, tabs, maxit in (("first_pass_maxt125", d, dsym, td, n // 4), ("perturb_20", hd, hdt, th, 20),
                                 ("repair", d, dsym, td, n // 4), ("perturb_20_b", hd, hdt, th, 20),
                                 ("repair_b", d, dsym, td, n // 4)):
    (r_dense, s_dense), t_dense = timed(lambda: engine.two_opt_(m, cur.clone(), maxit, want_sweeps=True, dist_t=mt))
    (r_nbr, s_nbr), t_nbr = timed(lambda: engine.two_opt_(m, cur.clone(), maxit, want_sweeps=True, tables=tabs, kernel="nbr"))
    (r_auto, s_auto), t_auto = timed(lambda: engine.two_opt_(m, cur.clone(), maxit, want_sweeps=True, tables=tabs))
    out[name] = {"dense_ms": t_dense * 1e3, "nbr_ms": t_nbr * 1e3, "auto_ms": t_auto * 1e3, "sweeps_mean": float(s_dense.float().mean()),
                 "equal": bool(torch.equal(r_dense, r_nbr) and torch.equal(s_dense, s_nbr) and torch.equal(r_dense, r_auto)
                               and torch.equal(s_dense, s_auto)),
                 "dense_Msweeps_s": float(s_dense.sum()) / t_dense / 1e6, "nbr_Msweeps_s": float(s_nbr.sum()) / t_nbr / 1e6}
    cur = r_dense
# first pass to convergence (inference setting)
(r_dense, s_dense), t_dense = timed(lambda: engine.two_opt_(d, tours.clone(), 10000, want_sweeps=True, dist_t=dsym))
(r_nbr, s_nbr), t_nbr = timed(lambda: engine.two_opt_(d, tours.clone(), 10000, want_sweeps=True, tables=td, kernel="nbr"))
(r_auto, s_auto), t_auto = timed(lambda: engine.two_opt_(d, tours.clone(), 10000, want_sweeps=True, tables=td))
out["first_pass_to_convergence"] = {"dense_ms": t_dense * 1e3, "nbr_ms": t_nbr * 1e3, "auto_ms": t_auto * 1e3,
                                    "sweeps_mean": float(s_dense.float().mean()),
                                    "equal": bool(torch.equal(r_dense, r_nbr) and torch.equal(s_dense, s_nbr)
                                                  and torch.equal(r_dense, r_auto) and torch.equal(s_dense, s_auto))}
print(json.dumps(out, indent=1))
