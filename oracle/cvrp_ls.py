"""CPU restatement of the CVRP local search specified in deepaco_amd/csrc/daco_cvrp_ls.hip.  TEST INFRASTRUCTURE.

PARITY UNPINNED against the reference: cvrp_nls/aco.py:114-126,443-448 hands every ant's routes to the vendored
HGS-CVRP C++ (cvrp_nls/swapstar.py:240-271 -> Program/C_Interface.cpp:128-172), whose LocalSearch visits its
neighbourhoods in a randomised order and is not restated.  What is restated here is THIS repository's deterministic
best-improvement search (relocate, swap, intra-route 2-opt; float32, the expression order of the kernel's header
comment; ties to the smallest (kind, i, j)), so that the kernel can be held bit-exact against an independent
implementation, next to the properties any such search must have (feasible, never worse, a local optimum at the end).
Pure Python / numpy float32 scalars: small cases only.
"""
import numpy as np

F = np.float32


def compress(seq):
    out = []
    for v in seq:
        v = int(v)
        if v == 0 and out and out[-1] == 0:
            continue
        out.append(v)
    if not out or out[-1] != 0:
        out.append(0)
    return out


def _tables(s, d, dem):
    L = len(s)
    rid, load, asym = [0] * L, [], [np.float64(0)] * L
    r, w = -1, np.float64(0)
    for k in range(L):
        if s[k] == 0:
            r += 1
            load.append(F(0))
        else:
            load[r] = F(load[r] + dem[s[k]])
        rid[k] = r
        asym[k] = w
        if k + 1 < L:
            w = w + (np.float64(d[s[k + 1], s[k]]) - np.float64(d[s[k], s[k + 1]]))
    return rid, load, asym


def best_move(s, d, dem, cap):
    """(delta, kind, i, j) of the best move, or None; the kernel's evaluation and tie-break."""
    L = len(s)
    rid, load, asym = _tables(s, d, dem)
    best = (F(0), 3, 0, 0)

    def consider(delta, kind, i, j):
        nonlocal best
        if delta < best[0] or (delta == best[0] and (kind, i, j) < best[1:]):
            best = (delta, kind, i, j)

    for i in range(1, L - 1):
        u = s[i]
        if u == 0:
            continue
        a, c = s[i - 1], s[i + 1]
        for j in range(0, L - 1):
            if j == i or j == i - 1:
                continue
            if rid[j] != rid[i] and F(load[rid[j]] + dem[u]) > cap:
                continue
            v, w = s[j], s[j + 1]
            rem = F(F(d[a, c] - d[a, u]) - d[u, c])
            add = F(F(d[v, u] + d[u, w]) - d[v, w])
            consider(F(rem + add), 0, i, j)
    for i in range(1, L - 1):
        u = s[i]
        if u == 0:
            continue
        for j in range(i + 1, L - 1):
            v = s[j]
            if v == 0:
                continue
            if rid[i] != rid[j]:
                if F(F(load[rid[i]] - dem[u]) + dem[v]) > cap or F(F(load[rid[j]] - dem[v]) + dem[u]) > cap:
                    continue
            a, g = s[i - 1], s[j + 1]
            if j == i + 1:
                nw = F(F(d[a, v] + d[v, u]) + d[u, g])
                od = F(F(d[a, u] + d[u, v]) + d[v, g])
                delta = F(nw - od)
            else:
                c, e = s[i + 1], s[j - 1]
                t1 = F(F(d[a, v] + d[v, c]) - F(d[a, u] + d[u, c]))
                t2 = F(F(d[e, u] + d[u, g]) - F(d[e, v] + d[v, g]))
                delta = F(t1 + t2)
            consider(delta, 1, i, j)
            if rid[i] == rid[j]:
                ends = F(F(d[a, v] + d[u, g]) - F(d[a, u] + d[v, g]))
                inner = F(asym[j] - asym[i])
                consider(F(ends + inner), 2, i, j)
    return None if best[1] == 3 else best


def apply_move(s, kind, i, j):
    s = list(s)
    if kind == 0:
        u = s.pop(i)
        s.insert(j if j > i else j + 1, u)
        s = compress(s)
    elif kind == 1:
        s[i], s[j] = s[j], s[i]
    else:
        s[i:j + 1] = s[i:j + 1][::-1]
    return s


def local_search(seq, dist, demand, capacity, max_moves):
    """seq: route sequence (zero-padded) -> (improved sequence without empty routes, moves applied)."""
    d = np.asarray(dist, dtype=np.float32)
    dem = np.asarray(demand, dtype=np.float32)
    cap = F(capacity)
    s = compress(seq)
    moves = 0
    while moves < max_moves:
        mv = best_move(s, d, dem, cap)
        if mv is None or not (mv[0] < F(-1e-6)):
            break
        s = apply_move(s, *mv[1:])
        moves += 1
    return s, moves


def route_cost(s, dist):
    d = np.asarray(dist, dtype=np.float64)
    return float(sum(d[s[k], s[k + 1]] for k in range(len(s) - 1)))


def feasible(s, demand, capacity, n):
    cust = [v for v in s if v != 0]
    if sorted(cust) != list(range(1, n)) or s[0] != 0 or s[-1] != 0:
        return False
    load = 0.0
    for v in s:
        load = 0.0 if v == 0 else load + float(demand[v])
        if load > capacity + 1e-6:
            return False
    return True
