"""CPU restatement of the CVRP local search specified in deepaco_amd/csrc/daco_cvrp_ls.hip.  TEST INFRASTRUCTURE.

The reference's CVRP local search is the vendored HGS-CVRP C++ (cvrp_nls/aco.py:114-126,443-448 -> cvrp_nls/swapstar.py:240-271
-> HGS-CVRP-main/Program/C_Interface.cpp:128-172 -> LocalSearch.cpp): first improvement over a granular neighbourhood in
a shuffled order (std::minstd_rand, std::shuffle), load penalties instead of hard capacity, SWAP* on top.  It is NOT
restated move for move.  What is restated here is THIS repository's deterministic best-improvement search over HGS's
move families (LocalSearch.cpp move1 .. move9: relocate one / two / two reversed, swap 1-1 / 2-1 / 2-2, 2-opt, 2-opt* in
both reconnections; swapStar: two customers of different routes change routes, each at its best position, tried when no
classical move improves), hard capacity, so that the kernel can be held bit-exact against an independent
implementation; parity with the reference is pinned on COST: tests/golden/g8_cvrp_ls_*.npz hold routes in / routes out of
the reference's own swapstar() / neural_swapstar() built from its sources (oracle/_ref), and the kernel's schedule has to
reach their mean cost (tests/test_gpu_09_cvrp_ls.py).

Specification (the kernel's header has the same text).  A solution is the route sequence 0 a b 0 c d 0 ... 0 without
empty routes, L entries.  Per move every candidate (kind, i, j) below is evaluated in float32 as
    change = (((+a1 + a2) + a3) + a4) - (((r1 + r2) + r3) + r4)  [+ (float32)(reversal term, float64)]
(a* = lengths of the edges the move adds, r* = of those it removes, in the order listed; missing terms are skipped), the
smallest change wins, ties to the smallest (kind, i, j), and it is applied if it is below -eps, eps = max(1e-6, M * 2^-17)
with M the largest |entry| of the matrix: a change is a sum of up to a dozen f32 terms of size <= M, so its rounding error
stays below 2e-6 * M and every applied move lowers the true cost -- the search cannot cycle (with an absolute 1e-6 it did,
on matrices with entries in the hundreds, once SWAP* with its different association of the terms was in the move set).  Loads are float64 sums
of the float32 demands along a route; a route is feasible if its load is <= capacity * (1 + 1e-6) (an exactly full route
of normalised demands must pass: the reference hands HGS capacity 1000.001 for the same reason, swapstar.py:254).
SWAP* (kind 9, evaluated only when no candidate of kinds 0-8 is below -1e-6): u = s[i] in route r1, v = s[j] in route
r2 > r1, change = ((remU + remV) + insU) + insV, remU = D(a,c) - (D(a,u) + D(u,c)), insU = the cheaper of "in place of v"
(D(e,u) + D(u,g)) - D(e,g) and the cheapest (D(s[p],u) + D(u,s[p+1])) - D(s[p],s[p+1]) over r2's positions p (opening depot
included) not next to v; ties: the smallest p, and the place of v before any p.
Pure Python / numpy scalars: small cases only.
"""
import numpy as np

F = np.float32
KINDS = ("rel1", "rel2", "rel2r", "swap11", "swap21", "swap22", "2opt", "tails", "cross", "swapstar")


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


def _ins_best(s, d, node, k0, k1, skip):
    """cheapest insertion of `node` between s[p] and s[p+1], p in [k0, k1), without the customer at position skip:
    (cost, position it goes behind; skip = the place of the removed customer)."""
    best, bp = F(np.inf), skip
    for p in range(k0, k1):
        if p == skip or p == skip - 1:
            continue
        c = F(F(d[s[p], node] + d[node, s[p + 1]]) - d[s[p], s[p + 1]])
        if c < best:
            best, bp = c, p
    e, g = s[skip - 1], s[skip + 1]
    cin = F(F(d[e, node] + d[node, g]) - d[e, g])
    return (cin, skip) if cin <= best else (best, bp)


def swap_star_candidates(s, d, dem, cap):
    """Every SWAP* candidate as (change, 9, i, j): u = s[i] and v = s[j] customers, route of i before route of j."""
    L = len(s)
    rid, pf, rl, start, asym, asymT = _tables(s, d, dem)
    capT = float(cap) * (1.0 + 1e-6)
    out = []
    for i in range(L - 1):
        for j in range(L - 1):
            u, v = s[i], s[j]
            if u == 0 or v == 0 or not rid[i] < rid[j]:
                continue
            du, dv = float(dem[u]), float(dem[v])
            if not (rl[rid[i]] - du + dv <= capT and rl[rid[j]] - dv + du <= capT):
                continue
            a, c, e, g = s[i - 1], s[i + 1], s[j - 1], s[j + 1]
            rem_u = F(d[a, c] - F(d[a, u] + d[u, c]))
            rem_v = F(d[e, g] - F(d[e, v] + d[v, g]))
            ins_u, _ = _ins_best(s, d, u, start[rid[j]], start[rid[j] + 1], j)
            ins_v, _ = _ins_best(s, d, v, start[rid[i]], start[rid[i] + 1], i)
            out.append((F(F(F(rem_u + rem_v) + ins_u) + ins_v), 9, i, j))
    return out


def _best_of(cands):
    best = None
    for c in cands:
        if best is None or c[0] < best[0] or (c[0] == best[0] and c[1:] < best[1:]):
            best = c
    return best


def best_swap_star(s, d, dem, cap):
    return _best_of(swap_star_candidates(s, d, dem, cap))


def best_move(s, d, dem, cap):
    """(change, kind, i, j) of the best classical move (kinds 0-8; smallest change, ties to the smallest (kind, i, j)), or None."""
    best = None
    for c in candidates(s, d, dem, cap):
        if best is None or c[0] < best[0] or (c[0] == best[0] and c[1:] < best[1:]):
            best = c
    return best


def apply_move(s, kind, i, j, d=None):
    s = list(s)
    L = len(s)
    if kind == 9:
        rid, _, _, start, _, _ = _tables(s, d, np.zeros(int(max(s)) + 1, dtype=np.float32))
        u, v = s[i], s[j]
        _, pu = _ins_best(s, d, u, start[rid[j]], start[rid[j] + 1], j)
        _, pv = _ins_best(s, d, v, start[rid[i]], start[rid[i] + 1], i)
        new = []
        for k in range(L):
            if k == i:
                if pv == i:
                    new.append(v)
                continue
            if k == j:
                if pu == j:
                    new.append(u)
                continue
            new.append(s[k])
            if k == pv:
                new.append(v)
            if k == pu:
                new.append(u)
        return compress(new)
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


def threshold(dist):
    """eps of the acceptance rule: max(1e-6, M * 2^-17), M = largest |entry| (f32)."""
    m = F(np.abs(np.asarray(dist, dtype=np.float32)).max())
    return max(F(1e-6), F(m * F(2.0 ** -17)))


def local_search(seq, dist, demand, capacity, max_moves, kinds=None):
    """seq: route sequence (zero-padded) -> (improved sequence without empty routes, moves applied); the kinds of the
    applied moves are appended to `kinds` if a list is given."""
    d = np.asarray(dist, dtype=np.float32)
    dem = np.asarray(demand, dtype=np.float32)
    s = compress(seq)
    eps = threshold(d)
    moves = 0
    while moves < max_moves:
        mv = best_move(s, d, dem, capacity)
        if mv is None or not (mv[0] < -eps):
            mv = best_swap_star(s, d, dem, capacity)            # only when no classical move improves
            if mv is None or not (mv[0] < -eps):
                break
        s = apply_move(s, *mv[1:], d=d)
        moves += 1
        if kinds is not None:
            kinds.append(mv[1])
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
