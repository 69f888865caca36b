"""CPU checks of oracle/cvrp_ls.py, the restatement the CVRP local-search kernel is held against: every candidate's
evaluated change is the change of the route cost when the move is applied (so the specification's index arithmetic and
its apply rules agree with each other), capacity is respected, and the search ends where no move of the ten families
improves."""
import numpy as np
import pytest

from oracle import cvrp_ls as ols


def instance(n, seed, asym=False):
    rng = np.random.default_rng(seed)
    c = rng.random((n, 2)).astype(np.float32)
    d = np.sqrt(((c[:, None] - c[None]) ** 2).sum(-1)).astype(np.float32)
    if asym:
        d = (d * (1 + 0.3 * rng.random((n, n)))).astype(np.float32)
    np.fill_diagonal(d, 1e-10)
    dem = np.concatenate(([0], rng.integers(1, 10, n - 1))).astype(np.float32)
    order = rng.permutation(np.arange(1, n))
    s, load = [0], 0.0
    for v in order:
        if load + dem[v] > 25.0:
            s.append(0)
            load = 0.0
        s.append(int(v))
        load += dem[v]
    s.append(0)
    return d, dem, s


@pytest.mark.parametrize("n,seed,asym", [(12, 1, False), (18, 2, False), (18, 3, True), (25, 4, True)])
def test_every_candidate_change_is_the_cost_change(n, seed, asym):
    d, dem, s = instance(n, seed, asym)
    base = ols.route_cost(s, d)
    cands = ols.candidates(s, d, dem, 25.0) + ols.swap_star_candidates(s, d, dem, 25.0)
    assert {c[1] for c in cands} >= {0, 3, 6, 7, 9}
    for change, kind, i, j in cands:
        out = ols.apply_move(s, kind, i, j, d=d)
        assert ols.feasible(out, dem, 25.0, n), (kind, i, j)
        assert abs((ols.route_cost(out, d) - base) - float(change)) < 2e-5, (kind, i, j)


@pytest.mark.parametrize("n,seed,asym", [(14, 5, False), (20, 6, True)])
def test_search_ends_at_a_local_optimum_of_all_ten_families(n, seed, asym):
    d, dem, s = instance(n, seed, asym)
    out, moves = ols.local_search(s, d, dem, 25.0, 1000)
    assert moves > 0 and ols.feasible(out, dem, 25.0, n) and ols.route_cost(out, d) < ols.route_cost(s, d)
    for best in (ols.best_move(out, d, dem, 25.0), ols.best_swap_star(out, d, dem, 25.0)):
        assert best is None or not (best[0] < -ols.threshold(d))


@pytest.mark.parametrize("scale", [1.0, 300.0])
def test_search_terminates_on_large_valued_matrices(scale):
    """the acceptance threshold scales with the matrix (eps = max(1e-6, M 2^-17)): with the absolute 1e-6 a row-scaled
    instance of 26 customers cycled between two solutions whose evaluated changes were one ulp of the edge lengths."""
    rng = np.random.default_rng(5)
    for _ in range(12):
        n = int(rng.integers(10, 30))
        d, dem, s = instance(n, int(rng.integers(1 << 20)))
        d = (d * rng.uniform(1.0, scale, size=(n, 1))).astype(np.float32)
        out, moves = ols.local_search(s, d, dem, 25.0, 400)
        assert moves < 400 and ols.route_cost(out, d) <= ols.route_cost(s, d)


def test_swap_star_puts_a_customer_where_it_belongs_not_where_its_partner_was():
    """two routes on a line: node 3 (x = 5.5) sits in route A, node 7 (x = 3) between 4 and 5 in route B.  Exchanged in place
    node 3 would land between 4 and 5; SWAP* puts it between 5 and 6."""
    xs = np.array([0, 1, 2, 5.5, 4, 5, 6, 3, 8], dtype=np.float32)
    d = np.abs(xs[:, None] - xs[None]).astype(np.float32)
    dem = np.concatenate(([0], np.ones(8))).astype(np.float32)
    s = [0, 1, 2, 3, 0, 4, 7, 5, 6, 8, 0]
    cands = {(c[2], c[3]): c[0] for c in ols.swap_star_candidates(s, d, dem, 5.0)}
    out = ols.apply_move(s, 9, 3, 6, d=d)
    assert out == [0, 1, 2, 7, 0, 4, 5, 3, 6, 8, 0]
    assert abs((ols.route_cost(out, d) - ols.route_cost(s, d)) - float(cands[(3, 6)])) < 1e-6
    assert ols.best_swap_star(s, d, dem, 5.0)[1:] == (9, 3, 6)
