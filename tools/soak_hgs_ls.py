#!/usr/bin/env python3
"""Randomised soak of the route-exact CVRP local search (daco_hgs_local_search) against oracle/hgs_ls.c: random sizes, capacities,
ant counts, loop bounds, one to three stages, matrix kinds (Euclidean, integer grid with many tied distances, asymmetric
perturbation matrices, scaled), solutions from random permutations (far from optimal: empty routes appear and are used).
usage: tools/soak_hgs_ls.py [seconds=60] [seed=0]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepaco_amd import engine  # noqa: E402
import oracle  # noqa: E402

dev = torch.device("cuda:0")


def instance(rng, n, kind):
    if kind == "grid":
        pos = rng.integers(0, 12, (n + 1, 2)).astype(np.float64)
    else:
        pos = rng.random((n + 1, 2))
    d = np.linalg.norm(pos[:, None] - pos[None], axis=-1)
    if kind == "scaled":
        d = d * rng.choice([0.5, 7.0, 300.0])
    d[np.arange(n + 1), np.arange(n + 1)] = 1e-10
    if kind == "grid":
        d = np.maximum(d, 1e-10)
    return pos, d


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0 = time.time()
    cases = sols = moves = bad = thrown = 0
    while time.time() - t0 < seconds:
        n = int(rng.choice([5, 8, 13, 21, 40, 64, 100, 129, 200, 300]))
        cap = int(rng.choice([10, 20, 30, 50]))
        A = int(rng.integers(1, 40))
        kind = str(rng.choice(["euclid", "grid", "scaled", "euclid"]))
        pos, d = instance(rng, n, kind)
        dem = np.concatenate(([0.0], rng.integers(1, 10, n) / cap))
        hd = 1 / ((1 / d) / (1 / d).max(-1, keepdims=True) * (0.2 + rng.random(d.shape)) + 1e-5)
        cols = []
        for _ in range(A):
            seq, load = [0], 0.0
            for c in rng.permutation(np.arange(1, n + 1)):
                if load + dem[c] > 1.0:
                    seq.append(0); load = 0.0
                seq.append(int(c)); load += dem[c]
            cols.append(seq + [0])
        L = max(map(len, cols)) + int(rng.integers(0, 4))
        paths = np.zeros((L, A), dtype=np.int64)
        for a, s in enumerate(cols):
            paths[:len(s), a] = s
        nst = int(rng.integers(1, 4))
        mats = [d if (i % 2 == 0) else hd for i in range(nst)]
        if rng.random() < 0.2:
            mats[0] = hd
        counts = [int(rng.choice([0, 1, 2, 3, 10, 100])) for _ in range(nst)]
        tabs = {id(m): engine.HgsTables(torch.as_tensor(m).to(dev)) for m in (d, hd)}
        work = torch.as_tensor(paths)[None].to(dev).contiguous()
        _, status, stats = engine.hgs_local_search_(work, [(tabs[id(m)], c) for m, c in zip(mats, counts)], torch.as_tensor(dem).to(dev),
                                                    want_stats=True)
        got = work[0].cpu().numpy()
        for a in range(A):
            seq, any_rc = paths[:, a], 0
            for m, c in zip(mats, counts):
                seq, rc = oracle.hgs_local_search(pos, m, dem, seq, c, out_len=L)
                any_rc |= rc
            thrown += any_rc
            if not np.array_equal(got[:, a], seq) or int(status[0, a]) != any_rc:
                bad += 1
                if bad <= 5:
                    print("MISMATCH", dict(n=n, cap=cap, A=A, kind=kind, counts=counts, ant=a, status=int(status[0, a]), rc=any_rc), flush=True)
        cases += 1
        sols += A
        moves += int(stats[..., 0].sum())
    print(f"soak_hgs_ls: {cases} cases, {sols} solutions, {moves} moves applied, {thrown} with a refused stage, {bad} mismatches "
          f"in {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
