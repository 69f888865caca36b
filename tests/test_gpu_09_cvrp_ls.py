"""GPU tests of the CVRP local search (daco_cvrp_local_search; cvrp_nls/aco.py:114-126, 443-448).

The reference's local search is the vendored HGS-CVRP C++ (first improvement in a shuffled order, load penalties,
SWAP*); it is not reproduced move for move.  The kernel -- best improvement over HGS's move families 1-9 and SWAP*, hard capacity --
is held (a) bit-exact against the independent CPU restatement of ITS OWN specification (oracle/cvrp_ls.py), (b) to the
properties any valid search has: feasible solutions, costs that never increase, a local optimum of the move set when it
stops by itself, and (c) COST-PINNED against the reference: g8 fixtures = routes in / routes out of the reference's own
swapstar() / neural_swapstar() (HGS built from the reference's sources, oracle/_ref); the drop-in's schedule has to reach
their mean cost."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

from oracle import cvrp_ls as ols

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def instance(n_cust, seed, cap=30.0):
    g = torch.Generator().manual_seed(seed)
    loc = torch.cat((torch.full((1, 2), 0.5), torch.rand(n_cust, 2, generator=g)), 0)
    dem = torch.cat((torch.zeros(1), torch.randint(1, 10, (n_cust,), generator=g).float()))
    d = torch.norm(loc[:, None] - loc, dim=2, p=2)
    i = torch.arange(n_cust + 1)
    d[i, i] = 1e-10
    return d, dem, cap


def sample_paths(d, dem, cap, A, seed):
    from deepaco_amd import engine
    tau = torch.ones_like(d)
    paths, _, _, lens, flags = engine.cvrp_sample(tau.to(dev())[None], (1 / d).to(dev())[None], dem.to(dev()), cap, A, seed=seed)
    assert int(flags.sum()) == 0
    L = int(lens.max())
    return paths[:, :L + 2].contiguous()                  # two spare rows: relocations never lengthen a sequence anyway


@pytest.mark.parametrize("n_cust,A,moves,asym", [(12, 6, 1000, False), (20, 8, 1000, False), (20, 5, 7, False), (25, 6, 1000, True),
                                               (40, 4, 1000, False)])
def test_local_search_equals_cpu_restatement(n_cust, A, moves, asym):
    from deepaco_amd import engine
    d, dem, cap = instance(n_cust, 100 + n_cust)
    if asym:                                              # the perturbation matrix of neural_swapstar is not symmetric
        d = d * (1 + 0.3 * torch.rand(d.shape, generator=torch.Generator().manual_seed(5)))
    paths = sample_paths(d, dem, cap, A, seed=3)
    before = paths.clone()
    out, lens, nmoves = engine.cvrp_local_search_(d.to(dev()), dem.to(dev()), cap, paths, moves, want_stats=True)
    for a in range(A):
        ref, ref_moves = ols.local_search(before[0, :, a].cpu().numpy(), d.numpy(), dem.numpy(), cap, moves)
        got = out[0, :, a].cpu().numpy()
        assert int(lens[0, a]) == len(ref) and int(nmoves[0, a]) == ref_moves, a
        assert np.array_equal(got[:len(ref)], np.array(ref)) and not got[len(ref):].any(), a


@pytest.mark.parametrize("n_cust,A,B", [(100, 32, 3), (200, 8, 2)])
def test_local_search_properties_at_size(n_cust, A, B):
    """config-4-sized instances (and n > 160: the matrix stays in global memory): feasible, never worse, and when the
    search stopped by itself no improving move of the ten move families (SWAP* included) is left (checked with the restated
    evaluation on a few ants)."""
    from deepaco_amd import engine
    cap = 50.0
    ds, dems = zip(*[instance(n_cust, 7 + b, cap)[:2] for b in range(B)])
    d, dem = torch.stack(ds).to(dev()), torch.stack(dems).to(dev())
    tau = torch.ones_like(d)
    paths, _, _, lens, flags = engine.cvrp_sample(tau, 1 / d, dem, cap, A, seed=9)
    assert int(flags.sum()) == 0
    c0 = engine.tour_costs(d, paths, closed=False)
    work = paths.clone()
    _, lens2, moves = engine.cvrp_local_search_(d, dem, cap, work, 100000, want_stats=True)
    c1 = engine.tour_costs(d, work, closed=False)
    assert bool((c1 <= c0 + 1e-4).all()) and bool((c1 < c0 - 1e-3).any())
    for b in range(B):
        for a in range(A):
            s = work[b, :int(lens2[b, a]), a].cpu().tolist()
            assert ols.feasible(s, dems[b].numpy(), cap, n_cust + 1), (b, a)
    for b, a in ((0, 0), (B - 1, A - 1)):
        s = work[b, :int(lens2[b, a]), a].cpu().tolist()
        for mv in (ols.best_move(s, ds[b].numpy(), dems[b].numpy(), np.float32(cap)),
                   ols.best_swap_star(s, ds[b].numpy(), dems[b].numpy(), np.float32(cap))):
            assert mv is None or not (mv[0] < -ols.threshold(ds[b].numpy())), (b, a, mv)
    # a second call finds nothing to do
    again = work.clone()
    _, _, m2 = engine.cvrp_local_search_(d, dem, cap, again, 100000, want_stats=True)
    assert int(m2.sum()) == 0 and torch.equal(again, work)


def test_local_search_accepts_the_large_sizes_of_the_reference():
    """cvrp_nls/utils.py:5 lists capacities up to n = 2000 and its script writes test sets for n = 1000 and 2000: route
    sequences of more than 1024 entries (the limit until round 3; now 4111) go through the kernel -- here n = 1000 customers,
    capacity 200, four sampled solutions of ~1100 entries, 40 moves each: feasible, shorter, exactly 40 moves made; and the
    size that is still too large is refused with a message, not a fault."""
    from deepaco_amd import engine
    n_cust, cap, A = 1000, 200.0, 4
    d, dem, _ = instance(n_cust, 77, cap)
    paths = sample_paths(d, dem, cap, A, seed=5)
    assert paths.shape[1] > 1024
    c0 = engine.tour_costs(d.to(dev())[None], paths, closed=False)
    work = paths.clone()
    _, lens, moves = engine.cvrp_local_search_(d.to(dev()), dem.to(dev()), cap, work, 40, want_stats=True)
    c1 = engine.tour_costs(d.to(dev())[None], work, closed=False)
    assert bool((moves == 40).all()) and bool((c1 < c0 - 1e-3).all())
    for a in range(A):
        assert ols.feasible(work[0, :int(lens[0, a]), a].cpu().tolist(), dem.numpy(), cap, n_cust + 1), a
    big = torch.zeros((1, 4200, 1), dtype=torch.int64, device=dev())
    with pytest.raises(ValueError, match="must stay below"):
        engine.cvrp_local_search_(d.to(dev()), dem.to(dev()), cap, big, 1)


def test_cvrp_nls_class_surface():
    """cvrp_nls/aco.py's surface: sample_nls() -> (costs, log_probs, costs_raw), multiple_swap_star in place, run() with
    swapstar=True; solutions stay feasible and the best cost does not get worse with more iterations."""
    from deepaco_amd.cvrp_nls.aco import ACO, get_subroutes, merge_subroutes
    d, dem, _ = instance(30, 77)
    dem = dem / 30.0                                       # cvrp_nls normalises demands, capacity 1.0
    heu = (1 / d).to(dev()).requires_grad_(True)
    with pytest.raises(AssertionError):                    # cvrp_nls/aco.py:73: swapstar needs the positions
        ACO(d.to(dev()), dem.to(dev()), n_ants=12, heuristic=heu, device="cuda:0", swapstar=True, seed=5)
    aco = ACO(d.to(dev()), dem.to(dev()), n_ants=12, heuristic=heu, device="cuda:0", swapstar=True, seed=5,
              positions=torch.rand(31, 2))
    costs, log_probs, costs_raw = aco.sample_nls()
    assert costs.shape == costs_raw.shape == (12,) and bool((costs <= costs_raw + 1e-5).all()) and log_probs.requires_grad
    paths = aco.gen_path()
    c0 = aco.gen_path_costs(paths)
    idx = c0.topk(4, largest=False).indices
    out = aco.multiple_swap_star(paths, indexes=idx)
    c1 = aco.gen_path_costs(out)
    keep = torch.ones(12, dtype=torch.bool, device=c0.device)
    keep[idx] = False
    assert torch.equal(c1[keep], c0[keep]) and bool((c1[idx] <= c0[idx] + 1e-5).all())
    for a in idx.tolist():
        s = out[:, a].cpu().tolist()
        while len(s) > 1 and s[-1] == 0 and s[-2] == 0:
            s.pop()
        assert ols.feasible(s, dem.numpy(), 1.0, 31)
    sub = get_subroutes(out[:, int(idx[0])])
    assert torch.equal(merge_subroutes(sub, out.shape[0], out.device), out[:, int(idx[0])])
    best1 = float(aco.run(2))
    best2 = float(aco.run(3))
    assert best2 <= best1 + 1e-6


@pytest.mark.parametrize("n", [20, 50, 100])
def test_class_reproduces_the_reference_routes(n):
    """g8: solutions sampled by the reference's ACO and improved by the reference's neural_swapstar (its Python over HGS built
    from its sources).  The drop-in's multiple_swap_star (default local_search='hgs') on the same sampled solutions returns
    THE SAME `paths`, entry for entry -- and swapstar(count = 10 / 100)'s through one-stage calls."""
    from deepaco_amd import engine
    from deepaco_amd.cvrp_nls.aco import ACO
    g = np.load(os.path.join(GOLDEN, f"g8_cvrp_ls_n{n}.npz"))
    A = g["paths_in"].shape[1]
    aco = ACO(torch.from_numpy(g["distances"]).to(dev()), torch.from_numpy(g["demands"]).to(dev()), n_ants=A,
              heuristic=torch.from_numpy(g["heuristic"]).to(dev()), device="cuda:0", swapstar=True,
              positions=torch.from_numpy(g["positions"]))
    assert int(g["limit"]) == max(aco.problem_size, 50)
    paths = torch.from_numpy(g["paths_in"]).to(dev())
    out = aco.multiple_swap_star(paths.clone())
    np.testing.assert_array_equal(out.cpu().numpy(), g["paths_nls"])
    c64 = np.array([ols.route_cost(ols.compress(out[:, a].cpu().tolist()), g["distances"]) for a in range(A)])
    np.testing.assert_allclose(c64, g["costs_nls"], rtol=1e-12)
    td, _ = aco._hgs_stage_tables()
    for cnt in (10, 100):
        one = engine.hgs_local_search_(paths.clone().unsqueeze(0).contiguous(), [(td, cnt)], torch.from_numpy(g["demands"]).to(dev()))
        np.testing.assert_array_equal(one[0].cpu().numpy(), g[f"paths_ls{cnt}"])
    # only some columns (cvrp_nls/aco.py:143-146: the best 8 of an iteration)
    idx = torch.tensor([3, 0, A - 1], device=dev())
    part = aco.multiple_swap_star(paths.clone(), indexes=idx).cpu().numpy()
    keep = np.ones(A, dtype=bool); keep[idx.cpu().numpy()] = False
    np.testing.assert_array_equal(part[:, ~keep], g["paths_nls"][:, ~keep])
    np.testing.assert_array_equal(part[:, keep], g["paths_in"][:, keep])


@pytest.mark.parametrize("n", [20, 50, 100, 200, 500])
def test_class_reproduces_the_reference_costs_on_many_instances(n):
    """g8b: 36 instances at n = 20 ... 500, eight sampled solutions each; the reference's neural_swapstar under the training
    (limit = max(n, 50)) and inference (limit = 100000 there, 10000 in the fixture's call) schedules.  The route-exact local
    search reaches the reference's float64 cost on every one of the 288 solutions (the fixture holds costs, not routes)."""
    from deepaco_amd.cvrp_nls.aco import ACO
    g = np.load(os.path.join(GOLDEN, "g8b_cvrp_ls_many.npz"))
    pos, dem, paths_in = g[f"n{n}_positions"], g[f"n{n}_demands"], g[f"n{n}_paths_in"]
    for i in range(pos.shape[0]):
        p = torch.from_numpy(pos[i])
        d = torch.norm(p[:, None] - p, dim=2, p=2, dtype=torch.double)      # cvrp_nls/utils.py:32-36
        d[torch.arange(n + 1), torch.arange(n + 1)] = 1e-10
        A = paths_in.shape[2]
        for inference, key in ((False, "costs_nls"), (True, "costs_nls_inf")):
            aco = ACO(d.to(dev()), torch.from_numpy(dem[i]).to(dev()), n_ants=A, heuristic=(1.0 / d).to(dev()), device="cuda:0",
                      swapstar=True, positions=p, inference=inference)
            out = aco.multiple_swap_star(torch.from_numpy(paths_in[i].astype(np.int64)).to(dev()))
            c = np.array([ols.route_cost(ols.compress(out[:, a].cpu().tolist()), d.numpy()) for a in range(A)])
            np.testing.assert_allclose(c, g[f"n{n}_{key}"][i], rtol=1e-12, err_msg=f"instance {i} inference {inference}")


@pytest.mark.parametrize("n", [20, 50, 100])
def test_local_search_reaches_the_reference_cost(n):
    """g8 (tests/golden/gen_g8_cvrp_ls.py): solutions sampled by the reference's ACO and improved by the reference's
    neural_swapstar (HGS LocalSearch: search on the distances, 10 loops on the heuristic-derived matrix, search again).
    The drop-in's multiple_swap_star with local_search="best_improvement" (round 3's kernel, kept as an option) on the same
    sampled solutions: every result feasible, none worse than its input, and the
    mean cost not more than 0.5 % above the reference's (measured on 24 / 24 / 16 solutions: ratio 0.9857 / 0.9985 / 0.9867 --
    best improvement over moves 1-9 and SWAP* with hard capacity against HGS's penalised first-improvement search; without
    SWAP* it was 0.9989 / 1.0004 / 0.9972; the sampled solutions are 2.5 x as long)."""
    from deepaco_amd.cvrp_nls.aco import ACO
    g = np.load(os.path.join(GOLDEN, f"g8_cvrp_ls_n{n}.npz"))
    dist64, dem = g["distances"], g["demands"]
    A = g["paths_in"].shape[1]
    aco = ACO(torch.from_numpy(dist64).to(dev()), torch.from_numpy(dem).to(dev()), n_ants=A,
              heuristic=torch.from_numpy(g["heuristic"]).to(dev()), device="cuda:0", swapstar=True,
              positions=torch.from_numpy(g["positions"]), local_search="best_improvement")
    np.testing.assert_allclose(aco.heuristic_dist.cpu().numpy(), g["heuristic_dist"], rtol=2e-6)
    paths = torch.from_numpy(g["paths_in"]).to(dev())
    out = aco.multiple_swap_star(paths.clone())
    costs = []
    for a in range(A):
        s = ols.compress(out[:, a].cpu().tolist())
        assert ols.feasible(s, dem, 1.0, n + 1), a
        costs.append(ols.route_cost(s, dist64))
    costs = np.array(costs)
    assert bool((costs <= g["costs_in"] + 1e-6).all())
    ratio = costs.mean() / g["costs_nls"].mean()
    print(f"n = {n}: mean cost {costs.mean():.4f} vs the reference's {g['costs_nls'].mean():.4f} (ratio {ratio:.4f}; sampled {g['costs_in'].mean():.4f})")
    assert ratio < 1.005


@pytest.mark.parametrize("n", [20, 50, 100, 200, 500])
def test_local_search_cost_distribution_on_many_instances(n):
    """g8b (tests/golden/gen_g8b_cvrp_ls_many.py): ten instances per size (four at n = 200, two at n = 500), eight solutions
    each, sampled by the reference's ACO.gen_path and improved by the reference's swapstar(count = 10 / 100) and
    neural_swapstar under the training (limit = max(n, 50)) and inference (limit = 10000) schedules -- HGS LocalSearch
    built from the reference's sources.  (The four reference columns: ls10 == ls100 and training == inference on every
    one of these 288 solutions -- HGS's loop count is not what limits it, so the drop-in's single schedule loses nothing
    by ignoring `inference`.)  The drop-in's multiple_swap_star on the same sampled solutions, per solution:
    feasible, never worse than its input; the distribution of cost / reference cost (not only its mean) is held:
    mean <= 1.003, nine solutions in ten within 3.5 %, none worse than 11 % (small instances: two local optima of a 20-customer
    instance differ by that much either way -- the best ratio is 0.90); and the plain local search of the reference
    (ls100, no perturbation stage) is matched or beaten on average."""
    from deepaco_amd.cvrp_nls.aco import ACO
    g = np.load(os.path.join(GOLDEN, "g8b_cvrp_ls_many.npz"))
    pos, dem, paths_in = g[f"n{n}_positions"], g[f"n{n}_demands"], g[f"n{n}_paths_in"]
    ratios, ratios_ls = [], []
    for i in range(pos.shape[0]):
        p = torch.from_numpy(pos[i])
        d = torch.norm(p[:, None] - p, dim=2, p=2, dtype=torch.double)      # cvrp_nls/utils.py:32-36
        d[torch.arange(n + 1), torch.arange(n + 1)] = 1e-10
        A = paths_in.shape[2]
        aco = ACO(d.to(dev()), torch.from_numpy(dem[i]).to(dev()), n_ants=A, heuristic=(1.0 / d).to(dev()), device="cuda:0",
                  swapstar=True, positions=p, local_search="best_improvement")
        out = aco.multiple_swap_star(torch.from_numpy(paths_in[i].astype(np.int64)).to(dev()))
        for a in range(A):
            s = ols.compress(out[:, a].cpu().tolist())
            assert ols.feasible(s, dem[i], 1.0, n + 1), (i, a)
            c = ols.route_cost(s, d.numpy())
            assert c <= g[f"n{n}_costs_in"][i, a] + 1e-6
            ratios.append(c / g[f"n{n}_costs_nls"][i, a])
            ratios_ls.append(c / g[f"n{n}_costs_ls100"][i, a])
        assert np.array_equal(g[f"n{n}_costs_nls"][i], g[f"n{n}_costs_nls_inf"][i])
    r = np.sort(np.array(ratios))
    print(f"n = {n}: {len(r)} solutions, cost / neural_swapstar cost: mean {r.mean():.4f} median {np.median(r):.4f} "
          f"p90 {r[int(0.9 * (len(r) - 1))]:.4f} max {r[-1]:.4f} min {r[0]:.4f}; vs swapstar(count=100): mean {np.mean(ratios_ls):.4f}")
    # measured (profiles/r04_cvrp_ls_cost_distribution.txt): mean 0.991-0.999, p90 1.015-1.029, max 1.021 (n = 200) ... 1.099 (n = 20)
    assert r.mean() <= 1.003 and r[int(0.9 * (len(r) - 1))] <= 1.035 and r[-1] <= 1.11
    assert np.mean(ratios_ls) <= 1.0
