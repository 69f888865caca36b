"""CPU restatement of the CVRP local search specified in deepaco_amd/csrc/daco_cvrp_ls.hip.  TEST INFRASTRUCTURE.

The reference's CVRP local search is the vendored HGS-CVRP C++ (cvrp_nls/aco.py:114-126,443-448 -> cvrp_nls/swapstar.py:240-271
-> HGS-CVRP-main/Program/C_Interface.cpp:128-172 -> LocalSearch.cpp): first improvement over a granular neighbourhood in
a shuffled order (std::minstd_rand, std::shuffle), load penalties instead of hard capacity, SWAP* on top.  It is NOT
restated move for move.  What is restated here is THIS repository's deterministic best-improvement search over HGS's
classical move families (LocalSearch.cpp move1 .. move9: relocate one / two / two reversed, swap 1-1 / 2-1 / 2-2,
2-opt, 2-opt* in both reconnections), hard capacity, so that the kernel can be held bit-exact against an independent
implementation; parity with the reference is pinned on COST: tests/golden/g8_cvrp_ls_*.npz hold routes in / routes out of
the reference's own swapstar() / neural_swapstar() built from its sources (oracle/_ref), and the kernel's schedule has to
reach their mean cost (tests/test_gpu_09_cvrp_ls.py).

Specification (the kernel's header has the same text).  A solution is the route sequence 0 a b 0 c d 0 ... 0 without
empty routes, L entries.  Per move every candidate (kind, i, j) below is evaluated in float32 as
    change = (((+a1 + a2) + a3) + a4) - (((r1 + r2) + r3) + r4)  [+ (float32)(reversal term, float64)]
(a* = lengths of the edges the move adds, r* = of those it removes, in the order listed; missing terms are skipped), the
smallest change wins, ties to the smallest (kind, i, j), and it is applied if it is below -1e-6.  Loads are float64 sums
of the float32 demands along a route; a route is feasible if its load is <= capacity * (1 + 1e-6) (an exactly full route
of normalised demands must pass: the reference hands HGS capacity 1000.001 for the same reason, swapstar.py:254).
Pure Python / numpy scalars: small cases only.
"""
import numpy as np

F = np.float32
KINDS = ("rel1", "rel2", "rel2r", "swap11", "swap21", "swap22", "2opt", "tails", "cross")


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
    """rid[k]: route of position k (a depot starts the route it opens; the last depot gets the count of routes);
    pf[k]: float64 load of the route up to and including k; rl[r]: route loads; start[r]: its opening depot;
    asym[k]: float64 sum over the route's edges before k of d[s[t+1]][s[t]] - d[s[t]][s[t+1]] (0 at the opening depot);
    asymT[r]: the same over the whole route, closing edge included."""
    L = len(s)
    rid, pf, asym = [0] * L, [0.0] * L, [0.0] * L
    rl, start, asymT = [], [], []
    r = -1
    for k in range(L):
        if s[k] == 0:
            if r >= 0:
                asymT.append(asym_run + (float(d[s[k], s[k - 1]]) - float(d[s[k - 1], s[k]])) if k > 0 else 0.0)
            r += 1
            start.append(k)
            rl.append(0.0)
            load, asym_run = 0.0, 0.0
        else:
            asym_run = asym_run + (float(d[s[k], s[k - 1]]) - float(d[s[k - 1], s[k]]))
            load = load + float(dem[s[k]])
            rl[r] = load
        rid[k] = r
        pf[k] = load
        asym[k] = asym_run
    return rid, pf, rl, start, asym, asymT


def _change(d, added, removed, extra=None):
    a = F(0)
    first = True
    for (p, q) in added:
        a = F(d[p, q]) if first else F(a + d[p, q])
        first = False
    r = F(0)
    first = True
    for (p, q) in removed:
        r = F(d[p, q]) if first else F(r + d[p, q])
        first = False
    c = F(a - r)
    if extra is not None:
        c = F(c + F(extra))
    return c


def candidates(s, d, dem, cap):
    """Every candidate move as (change, kind, i, j)."""
    L = len(s)
    rid, pf, rl, start, asym, asymT = _tables(s, d, dem)
    R = len(rl) - 1                                         # routes (the last depot opened an empty sentinel)
    capT = float(cap) * (1.0 + 1e-6)
    cust = [v != 0 for v in s]
    out = []
    for i in range(L - 1):
        for j in range(L - 1):
            u, v = s[i], s[j]
            a = s[i - 1] if i > 0 else 0
            # ---- moves of the customer (pair) at i
            if cust[i]:
                c = s[i + 1]
                w = s[j + 1]
                other = rid[j] != rid[i]
                if j != i and j != i - 1:                   # REL1: u goes between s[j] and s[j+1]
                    if not other or rl[rid[j]] + float(dem[u]) <= capT:
                        out.append((_change(d, [(a, c), (v, u), (u, w)], [(a, u), (u, c), (v, w)]), 0, i, j))
                if cust[i + 1] and j not in (i - 1, i, i + 1):
                    x, c2 = s[i + 1], s[i + 2]
                    if not other or rl[rid[j]] + float(dem[u]) + float(dem[x]) <= capT:
                        out.append((_change(d, [(a, c2), (v, u), (x, w)], [(a, u), (x, c2), (v, w)]), 1, i, j))
                        out.append((_change(d, [(a, c2), (v, x), (x, u), (u, w)], [(a, u), (u, x), (x, c2), (v, w)]), 2, i, j))
                if cust[j] and j > i:                       # SWAP11 and 2OPT
                    e, g = s[j - 1], s[j + 1]
                    ok = True
                    if other:
                        du, dv = float(dem[u]), float(dem[v])
                        ok = rl[rid[i]] - du + dv <= capT and rl[rid[j]] - dv + du <= capT
                    if ok:
                        if j == i + 1:
                            out.append((_change(d, [(a, v), (v, u), (u, g)], [(a, u), (u, v), (v, g)]), 3, i, j))
                        else:
                            out.append((_change(d, [(a, v), (v, c), (e, u), (u, g)], [(a, u), (u, c), (e, v), (v, g)]), 3, i, j))
                    if not other:
                        out.append((_change(d, [(a, v), (u, g)], [(a, u), (v, g)], asym[j] - asym[i]), 6, i, j))
                if cust[i + 1] and cust[j] and (j >= i + 3 or j <= i - 2):     # SWAP21: pair (u, x) with the single v
                    x, c2, e, g = s[i + 1], s[i + 2], s[j - 1], s[j + 1]
                    ok = True
                    if other:
                        dp, dv = float(dem[u]) + float(dem[x]), float(dem[v])
                        ok = rl[rid[i]] - dp + dv <= capT and rl[rid[j]] - dv + dp <= capT
                    if ok:
                        out.append((_change(d, [(a, v), (v, c2), (e, u), (x, g)], [(a, u), (x, c2), (e, v), (v, g)]), 4, i, j))
                if cust[i + 1] and cust[j] and j >= i + 3 and cust[j + 1]:       # SWAP22
                    x, c2, e, y, g = s[i + 1], s[i + 2], s[j - 1], s[j + 1], s[j + 2]
                    ok = True
                    if other:
                        dp, dq = float(dem[u]) + float(dem[x]), float(dem[v]) + float(dem[y])
                        ok = rl[rid[i]] - dp + dq <= capT and rl[rid[j]] - dq + dp <= capT
                    if ok:
                        out.append((_change(d, [(a, v), (y, c2), (e, u), (x, g)], [(a, u), (x, c2), (e, v), (y, g)]), 5, i, j))
            # ---- 2-opt*: cut r1 after position i, r2 after position j (r1 before r2)
            if rid[i] < rid[j] and rid[j] < R:
                r1, r2 = rid[i], rid[j]
                x, y = s[i + 1], s[j + 1]
                if pf[i] + (rl[r2] - pf[j]) <= capT and pf[j] + (rl[r1] - pf[i]) <= capT:
                    out.append((_change(d, [(u, y), (v, x)], [(u, x), (v, y)]), 7, i, j))
                if pf[i] + pf[j] <= capT and (rl[r1] - pf[i]) + (rl[r2] - pf[j]) <= capT:
                    rev = (asym[j] - 0.0) + (asymT[r1] - asym[i + 1] if s[i + 1] != 0 else 0.0)
                    out.append((_change(d, [(u, v), (x, y)], [(u, x), (v, y)], rev), 8, i, j))
    return out


def best_move(s, d, dem, cap):
    """(change, kind, i, j) of the best move (smallest change, ties to the smallest (kind, i, j)), or None."""
    best = None
    for c in candidates(s, d, dem, cap):
        if best is None or c[0] < best[0] or (c[0] == best[0] and c[1:] < best[1:]):
            best = c
    return best


def apply_move(s, kind, i, j):
    s = list(s)
    L = len(s)
    if kind in (0, 1, 2):
        n = 1 if kind == 0 else 2
        seg = s[i:i + n]
        if kind == 2:
            seg = seg[::-1]
        if j > i:
            s = s[:i] + s[i + n:j + 1] + seg + s[j + 1:]
        else:
            s = s[:j + 1] + seg + s[j + 1:i] + s[i + n:]
    elif kind == 3:
        s[i], s[j] = s[j], s[i]
    elif kind == 4:
        if j > i:
            s = s[:i] + [s[j]] + s[i + 2:j] + s[i:i + 2] + s[j + 1:]
        else:
            s = s[:j] + s[i:i + 2] + s[j + 1:i] + [s[j]] + s[i + 2:]
    elif kind == 5:
        s = s[:i] + s[j:j + 2] + s[i + 2:j] + s[i:i + 2] + s[j + 2:]
    elif kind == 6:
        s[i:j + 1] = s[i:j + 1][::-1]
    else:
        e1 = next(k for k in range(i + 1, L) if s[k] == 0)             # closing depot of r1
        s2 = max(k for k in range(j + 1) if s[k] == 0)                 # opening depot of r2
        e2 = next(k for k in range(j + 1, L) if s[k] == 0)
        if kind == 7:
            s = s[:i + 1] + s[j + 1:e2] + s[e1:j + 1] + s[i + 1:e1] + s[e2:]
        else:
            s = s[:i + 1] + s[s2 + 1:j + 1][::-1] + s[e1:s2 + 1] + s[i + 1:e1][::-1] + s[j + 1:e2] + s[e2:]
    return compress(s)


def local_search(seq, dist, demand, capacity, max_moves):
    """seq: route sequence (zero-padded) -> (improved sequence without empty routes, moves applied)."""
    d = np.asarray(dist, dtype=np.float32)
    dem = np.asarray(demand, dtype=np.float32)
    s = compress(seq)
    moves = 0
    while moves < max_moves:
        mv = best_move(s, d, dem, capacity)
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
        load = 0.0 if v == 0 else load + float(np.float32(demand[v]))
        if load > float(capacity) * (1.0 + 1e-6):
            return False
    return True
