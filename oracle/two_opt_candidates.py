"""Test infrastructure: a numpy restatement of the candidate-list 2-opt SPECIFICATION that
deepaco_amd/csrc/daco_two_opt_nbr.hip implements (tables, tolerance ranks, the two sides of an edge, the single walk per
node list for symmetric matrices, the closing edge), so that the specification itself -- not only the kernel -- is checked
against the reference's full evaluation (tsp_nls/two_opt.py:6-39, restated in oracle/daco_oracle.c orc_two_opt_batch).
Only tests import this module.
"""
import numpy as np


def tolerance(d):
    n = d.shape[0]
    off = ~np.eye(n, dtype=bool)
    M = np.float32(np.abs(d[off]).max())
    return np.float32(4.0) * np.spacing(np.float32(2.0) * M)


def build_tables(d):
    """nb_id[x][k], nb_d[x][k]: row x sorted by (d, id) ascending; rk[x][y] = #{v : d[x][v] < fl(d[x][y] + tol)}."""
    d = np.ascontiguousarray(d, dtype=np.float32)
    n = d.shape[0]
    tol = tolerance(d)
    order = np.lexsort((np.broadcast_to(np.arange(n), (n, n)), d), axis=1)
    nb_d = np.take_along_axis(d, order, 1)
    thr = (d + tol).astype(np.float32)
    rk = np.empty((n, n), dtype=np.int64)
    for x in range(n):
        rk[x] = np.searchsorted(nb_d[x], thr[x], side="left")
    return order, nb_d, rk


def best_move(d, t, tab, tabT, symmetric):
    """The reference's sweep restricted to the candidate pairs: (change, i, j) of the strict minimum, ties to the first
    (i, j) in row-major order; (0, -1, -1) if no candidate has a negative change."""
    n = len(t)
    pos = np.empty(n, dtype=np.int64)
    pos[t] = np.arange(n)
    nxt = np.roll(t, -1)
    e = d[t, nxt]                                            # e[m] = d[t[m]][t[m+1]], edge n-1 closes the tour
    nb_id, nb_d, rk = tab
    nbT_id, nbT_d, rkT = tabT
    best = (np.float32(0.0), -1, -1)

    def consider(i, j, a, b):
        nonlocal best
        change = np.float32(np.float32(np.float32(a + b) - e[i - 1]) - e[j])
        if change < best[0] or (change == best[0] and best[1] >= 0 and (i, j) < (best[1], best[2])):
            best = (change, i, j)

    if symmetric:
        for m in range(n + 1):                               # the list of node t[m] (t[n] = t[0]: closing edge, side B only)
            x = t[m % n]
            cA = rk[t[m], t[m + 1]] if m <= n - 3 else 0                     # side A of edge m:   i = m + 1
            cB = rkT[t[m % n], t[m - 1]] if m >= 3 else 0                    # side B of edge m-1: j = m - 1
            for k in range(max(cA, cB)):
                v, dv = nb_id[x, k], nb_d[x, k]
                pw = pos[v]
                if pw > m + 1:
                    if k < cA:
                        consider(m + 1, pw, dv, d[t[m + 1], t[(pw + 1) % n]])
                elif 1 <= pw and pw + 1 < m:
                    if k < cB:
                        consider(pw, m - 1, d[t[pw - 1], t[m - 1]], dv)
    else:
        for m in range(n):
            x, y = t[m], t[(m + 1) % n]
            if m <= n - 3:                                   # side A: i = m + 1, candidates v = t[j] near x
                for k in range(rk[x, y]):
                    j = pos[nb_id[x, k]]
                    if j > m + 1:
                        consider(m + 1, j, nb_d[x, k], d[t[m + 1], t[(j + 1) % n]])
            if m >= 2:                                       # side B: j = m, candidates u = t[i] near y in the transposed matrix
                for k in range(rkT[y, x]):
                    i = pos[nbT_id[y, k]]
                    if 1 <= i < m:
                        consider(i, m, d[t[i - 1], t[m]], nbT_d[y, k])
    return best


def two_opt(d, tour, max_iterations):
    """Tours and sweep counts of the candidate-list search; must equal the reference's for every input."""
    d = np.ascontiguousarray(d, dtype=np.float32)
    t = np.array(tour, dtype=np.int64)
    symmetric = bool(np.array_equal(d, d.T))
    tab = build_tables(d)
    tabT = tab if symmetric else build_tables(np.ascontiguousarray(d.T))
    it = 0
    while it < max_iterations:
        it += 1
        change, i, j = best_move(d, t, tab, tabT, symmetric)
        if not (i >= 0 and float(change) < -1e-6):
            break
        t[i:j + 1] = t[i:j + 1][::-1]
    return t.astype(np.uint16), it
