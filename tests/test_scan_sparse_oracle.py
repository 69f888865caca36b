"""scan_sparse on the CPU: the restatement (oracle/daco_oracle.c draw_scan_sparse) draws from the reference's categorical.

tsp/aco.py:165-177 draws the next node from Categorical(P[cur] * mask).  scan_sparse splits a row into a head and a tail
and rejects visited tail entries (a superset scheme): the accepted outcome must still have probability P_ij / sum(open P).
Checked by chi-square on many ants that share a state: the first step from a fixed start, and the second step of the ants
that made the most common first choice -- with heads that are NOT the heavy entries (random subsets, so that tail walks and
rejections actually happen), with the reference's k-sparse heuristic (where they never do), and with an exhausted head."""
import numpy as np
import pytest

import oracle


def _instance(n, seed, kind):
    tau, eta, head = _instance_parts(n, seed, kind)
    return oracle.prob_matrix(tau, eta), head


def _instance_parts(n, seed, kind):
    """(tau, eta, (head ids, live counts)): the GPU suite draws from the same instances through the C ABI."""
    rng = np.random.default_rng(seed)
    c = rng.random((n, 2))
    d = np.sqrt(((c[:, None] - c[None]) ** 2).sum(-1)).astype(np.float32)
    np.fill_diagonal(d, 1e9)
    if kind == "ksparse":                                  # tsp/aco.py:52-67: k nearest keep 1/d, the rest 1e-10
        k = n // 10
        idx = np.argsort(d, axis=1)[:, :k]
        eta = np.full((n, n), 1e-10, dtype=np.float32)
        np.put_along_axis(eta, idx, 1 / np.take_along_axis(d, idx, axis=1), axis=1)
        head = oracle.sparse_head_ids(eta, k)
    elif kind == "dense_random_head":                      # every entry matters, the head is an arbitrary subset
        eta = (1 / d).astype(np.float32)
        ids = np.zeros((n, 64), dtype=np.uint16)
        for i in range(n):
            ids[i, :20] = np.sort(rng.choice(n, 20, replace=False))
        head = (ids, np.full(n, 20, dtype=np.uint8))
    elif kind == "dense_wide_head":                        # 128 slots (eight per lane): an arbitrary subset of 90
        eta = (1 / d).astype(np.float32)
        ids = np.zeros((n, 128), dtype=np.uint16)
        for i in range(n):
            ids[i, :90] = np.sort(rng.choice(n, 90, replace=False))
        head = (ids, np.full(n, 90, dtype=np.uint8))
    elif kind == "ksparse_wide":                           # k = 100 of 160: the 128-slot head holds every live entry
        idx = np.argsort(d, axis=1)[:, :100]
        eta = np.full((n, n), 1e-10, dtype=np.float32)
        np.put_along_axis(eta, idx, 1 / np.take_along_axis(d, idx, axis=1), axis=1)
        head = oracle.sparse_head_ids(eta, 100)
        assert head[0].shape == (n, 128)
    else:                                                  # a head of 3: exhausted after a few steps (dense steps)
        eta = (1 / d).astype(np.float32)
        head = oracle.sparse_head_ids(eta, 3)
    tau = (0.5 + rng.random((n, n))).astype(np.float32)
    return tau, eta, head


def _chi2(counts, probs):
    keep = probs * counts.sum() >= 5
    exp = probs[keep] * counts.sum()
    return float(((counts[keep] - exp) ** 2 / exp).sum()), int(keep.sum()) - 1, int(counts[~keep].sum())


KINDS = ["dense_random_head", "ksparse", "tiny_head", "dense_wide_head", "ksparse_wide"]


@pytest.mark.parametrize("kind", KINDS)
def test_scan_sparse_draws_the_categorical(kind):
    n, A = 160, 30000
    P, (hid, cnt) = _instance(n, 5, kind)
    paths, rc, stats = oracle.tsp_sample_scan_sparse(P, hid, cnt, A, seed=11, fixed_start=0)
    assert rc == 0
    check_first_two_steps(kind, P, paths, stats)


def check_first_two_steps(kind, P, paths, stats):
    """paths [n, A] from fixed start 0, stats = (dense steps, tail walks, rejections): chi-square of the first step and of the
    conditioned second step against Categorical(P[cur] * mask) (tsp/aco.py:165-177).  Shared with the GPU suite, which holds the
    HIP kernel's draws to the same bound (tests/test_gpu_11_scan_sparse.py)."""
    n, A = paths.shape
    assert (np.sort(paths, axis=0) == np.arange(n)[:, None]).all()                     # every column a permutation
    # first step: Categorical(P[0][k], k != 0)
    p1 = P[0].astype(np.float64).copy(); p1[0] = 0; p1 /= p1.sum()
    c1 = np.bincount(paths[1], minlength=n).astype(np.float64)
    x2, dof, rest = _chi2(c1, p1)
    assert x2 < dof + 5 * np.sqrt(2 * dof) + 10, (kind, x2, dof)
    assert c1[0] == 0
    # second step of the ants whose first choice was the most common one
    j = int(np.argmax(c1))
    sel = paths[1] == j
    p2 = P[j].astype(np.float64).copy(); p2[[0, j]] = 0; p2 /= p2.sum()
    c2 = np.bincount(paths[2][sel], minlength=n).astype(np.float64)
    x2, dof, _ = _chi2(c2, p2)
    assert x2 < dof + 5 * np.sqrt(2 * dof) + 10, (kind, "step 2", x2, dof)
    if kind in ("dense_random_head", "dense_wide_head"):
        assert stats[1] > 0 and stats[2] > 0                                          # tail walks and rejections happened
    if kind in ("ksparse", "ksparse_wide"):
        assert stats[1] == 0 and stats[2] == 0 and 0 <= stats[0] < 0.1 * A * n         # never past the head; a few dense steps late in the tours
    if kind == "tiny_head":
        assert stats[0] > 0.1 * A * n                                                  # many dense steps (a quarter, measured)


def test_scan_sparse_head_values_and_tail_total():
    n = 200
    P, (hid, cnt) = _instance(n, 9, "ksparse")
    hv = oracle.sparse_head_values(P, hid, cnt)
    for i in (0, 57, n - 1):
        k = int(cnt[i])
        assert np.array_equal(hv[i, :k], P[i, hid[i, :k]]) and not hv[i, k:63].any()
        tail = np.delete(P[i].astype(np.float64), hid[i, :k])
        assert abs(float(hv[i, 63]) - tail.sum()) <= 1e-6 * tail.sum() + 1e-12
