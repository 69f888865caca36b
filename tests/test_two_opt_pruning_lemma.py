"""CPU check of the two arithmetic facts the candidate-list 2-opt kernel (csrc/daco_two_opt_nbr.hip) rests on, in numpy
float32 with the reference's expression order (tsp_nls/two_opt.py:16-19), on matrices of very different kinds:

 1. pruning: a pair whose computed change is negative satisfies a < fl(c + tol) or b < fl(e + tol) with
    tol = 4 ulp(2 max|d|) -- so walking only those candidates cannot miss the reference's minimum;
 2. incumbent filter: with u the pair's second load and dmin the smallest off-diagonal entry,
    ((t + dmin) - c) - e <= ((t + u) - c) - e in float32 (monotonicity of rounding) -- so a candidate whose lower
    bound is above an achieved change can be skipped without changing the minimum.
"""
import numpy as np
import pytest


def change_terms(d, t):
    n = len(t)
    prev, nxt = np.roll(t, 1), np.roll(t, -1)
    a = d[prev[:, None], t[None, :]]                       # d[t[i-1]][t[j]]
    b = d[t[:, None], nxt[None, :]]                        # d[t[i]][t[j+1]]
    c = d[prev, t][:, None]                                # d[t[i-1]][t[i]]
    e = d[t, nxt][None, :]                                 # d[t[j]][t[j+1]]
    i, j = np.arange(n)[:, None], np.arange(n)[None, :]
    valid = (i >= 1) & (i <= n - 2) & (j > i)
    return a, b, np.broadcast_to(c, a.shape), np.broadcast_to(e, a.shape), valid


def matrices(rng, n):
    c = rng.random((n, 2)).astype(np.float32)
    eu = np.sqrt(((c[:, None] - c[None]) ** 2).sum(-1)).astype(np.float32)
    yield "euclid", eu
    g = rng.integers(0, 6, size=(n, 2)).astype(np.float32)
    yield "grid", np.sqrt(((g[:, None] - g[None]) ** 2).sum(-1)).astype(np.float32)
    yield "rowscaled", (eu * rng.uniform(1, 300, size=(n, 1))).astype(np.float32)
    h = np.zeros((n, n), np.float32)
    k = max(2, n // 10)
    idx = np.argsort(eu + np.eye(n, dtype=np.float32) * 1e9, axis=1)[:, :k]
    np.put_along_axis(h, idx, rng.random((n, k)).astype(np.float32) + 0.05, 1)
    yield "plateau", (1 / (h / h.max(1, keepdims=True) + 1e-5)).astype(np.float32)
    yield "asym", (rng.random((n, n)) * 10 ** rng.uniform(-2, 4)).astype(np.float32)
    yield "signed", rng.uniform(-1, 1, size=(n, n)).astype(np.float32)


@pytest.mark.parametrize("seed", range(6))
def test_negative_changes_are_candidates_and_lower_bound_holds(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(8, 120))
    off = ~np.eye(n, dtype=bool)
    for kind, d in matrices(rng, n):
        d = d.copy()
        M = np.float32(np.abs(d[off]).max())
        dmin = np.float32(d[off].min())
        tol = np.float32(4.0) * np.spacing(np.float32(2.0) * M)
        for _ in range(4):
            t = rng.permutation(n)
            a, b, c, e, valid = change_terms(d, t)
            change = ((a + b) - c) - e                                        # float32, left to right
            cand = (a < (c + tol)) | (b < (e + tol))                          # thresholds rounded as the table build rounds them
            neg = valid & (change < 0)
            assert not (neg & ~cand).any(), (kind, n)
            # the filter's lower bounds, for the side whose table value is a (gathers b) and the side whose value is b
            lb_a = ((a + dmin) - c) - e
            lb_b = ((b + dmin) - c) - e
            assert (lb_a[valid] <= change[valid]).all() and (lb_b[valid] <= change[valid]).all(), (kind, n)
