#!/usr/bin/env python3
"""Randomised soak of sampler "scan_sparse" (daco_tsp_sample_sparse) against its CPU restatement (oracle draw_scan_sparse):
random sizes, ant counts, head widths (64 / 128 slots), head contents (k-sparse heuristic, arbitrary subsets, tiny heads),
start modes and WEIGHT SCALES -- heavy-tailed pheromone so that running sums absorb small terms and the rounding branches
(no running sum reaches the remainder -> the lane's last positive slot; r past the head by an ulp) are taken.
Tours and the three step counters must agree bit for bit.  usage: tools/soak_scan_sparse.py [seconds=60] [seed=0]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from deepaco_amd import engine  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
t0 = time.time()
runs = mism = 0
steps = np.zeros(3, dtype=np.int64)
kinds = {}
while time.time() - t0 < budget:
    n = int(rng.choice([129, 160, 200, 255, 256, 257, 300, 400, 500, 512, 513, 640, 777, 1000, 1024]))
    A = int(rng.integers(1, 40))
    wide = bool(rng.integers(0, 2))
    slots = 128 if wide else 64
    kind = str(rng.choice(["ksparse", "subset", "tiny"]))
    c = rng.random((n, 2))
    d = np.sqrt(((c[:, None] - c[None]) ** 2).sum(-1)).astype(np.float32)
    np.fill_diagonal(d, 1e9)
    scale = str(rng.choice(["flat", "heavy", "huge_head"]))
    tau = (0.5 + rng.random((n, n))).astype(np.float32)
    if scale == "heavy":
        tau = np.exp(rng.normal(0, 6, (n, n))).astype(np.float32)          # twelve decades inside a row
    if kind == "ksparse":
        k = int(rng.integers(5, slots))
        k = min(k, n - 1)
        idx = np.argsort(d, axis=1)[:, :k]
        eta = np.full((n, n), 1e-10, dtype=np.float32)
        np.put_along_axis(eta, idx, 1 / np.take_along_axis(d, idx, axis=1), axis=1)
        ids = np.zeros((n, slots), dtype=np.uint16)
        ids[:, :k] = np.sort(idx, axis=1)
        cnt = np.full(n, k, dtype=np.uint8)
    else:
        eta = (1 / d).astype(np.float32)
        ids = np.zeros((n, slots), dtype=np.uint16)
        hi = 4 if kind == "tiny" else slots
        cnt = rng.integers(1, hi, n).astype(np.uint8)
        for r in range(n):
            ids[r, :cnt[r]] = np.sort(rng.choice(n, int(cnt[r]), replace=False))
    if scale == "huge_head":                                               # one head entry dwarfs its lane: later terms are absorbed
        r_ = rng.integers(0, n, n // 4)
        for r in r_:
            tau[r, ids[r, rng.integers(0, max(1, cnt[r]))]] *= np.float32(3e7)
    fixed = int(rng.integers(-1, n)) if rng.integers(0, 2) else -1
    seed, it = int(rng.integers(0, 1 << 30)), int(rng.integers(0, 50))
    head = ids.astype(np.int64).copy()
    head[:, slots - 1] = cnt
    head_t = torch.from_numpy(head[None]).to(torch.int16).contiguous().to(dev)
    paths, flags, _, _, stats = engine.tsp_sample_sparse(torch.from_numpy(tau[None]).to(dev), torch.from_numpy(eta[None]).to(dev), A, head_t,
                                                        seed=seed, it=it, fixed_start=fixed, want_stats=True)
    P = oracle.prob_matrix(tau, eta)
    ref, rc, st = oracle.tsp_sample_scan_sparse(P, ids, cnt, A, seed=seed, it=it, fixed_start=fixed)
    got = paths[0].cpu().numpy()
    ok = rc == 0 and int(flags.sum()) == 0 and np.array_equal(got, ref) and np.array_equal(stats.cpu().numpy(), st)
    runs += 1
    steps += st
    kinds[(kind, scale, slots)] = kinds.get((kind, scale, slots), 0) + 1
    if not ok:
        mism += 1
        bad = np.nonzero((got != ref).any(axis=0))[0]
        print(json.dumps({"mismatch": {"n": n, "A": A, "kind": kind, "scale": scale, "slots": slots, "seed": seed, "it": it, "fixed": fixed,
                                       "ants": bad[:4].tolist(), "first_step": [int(np.nonzero(got[:, a] != ref[:, a])[0][0]) for a in bad[:4]],
                                       "stats": stats.cpu().numpy().tolist(), "ref_stats": st.tolist(), "rc": int(rc)}}), flush=True)
# every ant of full-size launches: the rounding branch of a head draw (no running sum reaches the remainder) is taken about once
# in a million draws, so only launches of 1e7 draws exercise it -- the oracle counts how often it did
full = {}
for n, A, B, k in ((500, 512, int(os.environ.get("SOAK_FULL_B", "64")), 50), (1000, 512, 4, 100)):
    g = torch.Generator().manual_seed(n)
    c = torch.rand(B, n, 2, generator=g)
    d = (c[:, :, None] - c[:, None]).norm(dim=-1)
    i = torch.arange(n)
    d[:, i, i] = 1e9
    _, idx = torch.topk(d, k=k, dim=2, largest=False)
    eta = (1 / torch.full_like(d, 1e10).scatter_(2, idx, torch.gather(d, 2, idx))).contiguous()
    tau = (0.5 + torch.rand(B, n, n, generator=g)).contiguous()
    head = engine.sparse_head(eta.to(dev), k)
    paths, flags, _, _, stats = engine.tsp_sample_sparse(tau.to(dev), eta.to(dev), A, head, seed=11, it=2, want_stats=True)
    got = paths.cpu().numpy()
    before = oracle.sparse_rounding_picks()
    bad_ants = 0
    ref_stats = np.zeros(3, dtype=np.int64)
    hid = head.cpu().numpy().view(np.uint16)
    for b in range(B):
        P = oracle.prob_matrix(tau[b].numpy(), eta[b].numpy())
        ids = hid[b].copy()
        cnt = ids[:, -1].astype(np.uint8)
        ids[:, -1] = 0
        ref, rc, st = oracle.tsp_sample_scan_sparse(P, ids, cnt, A, seed=11, it=2, ant_gid0=b * A)
        bad_ants += int((got[b] != ref).any(axis=0).sum()) + (rc != 0)
        ref_stats += st
    full[f"tsp{n}_a{A}_b{B}_k{k}"] = {"draws": B * A * (n - 1), "ants_that_differ": bad_ants, "stats_equal": bool(np.array_equal(stats.cpu().numpy(), ref_stats)),
                                      "oracle_rounding_picks": oracle.sparse_rounding_picks() - before, "stats": ref_stats.tolist()}
    mism += bad_ants
print(json.dumps({"full_size": full}))
print(json.dumps({"runs": runs, "mismatches": mism, "dense_steps": int(steps[0]), "tail_walks": int(steps[1]), "rejections": int(steps[2]),
                  "oracle_rounding_picks": oracle.sparse_rounding_picks(), "seconds": round(time.time() - t0, 1), "cases": {"/".join(map(str, k)): v for k, v in sorted(kinds.items())}}))
