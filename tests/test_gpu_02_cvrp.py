"""GPU parity tests for the CVRP rollout path (cvrp/aco.py:107-205) through the C ABI."""
import glob
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev())


def cvrp_instance(n, seed, B=1):
    """cvrp/utils.py:9-22: depot (0.5,0.5), demands 1..9, diag 1e-10."""
    g = torch.Generator().manual_seed(seed)
    loc = torch.rand(B, n, 2, generator=g)
    dem = torch.randint(1, 10, (B, n), generator=g).float()
    allloc = torch.cat((torch.full((B, 1, 2), 0.5), loc), 1)
    demand = torch.cat((torch.zeros(B, 1), dem), 1)
    d = torch.cdist(allloc, allloc)
    i = torch.arange(n + 1)
    d[:, i, i] = 1e-10
    tau = torch.rand(B, n + 1, n + 1, generator=g) + 0.1
    eta = torch.rand(B, n + 1, n + 1, generator=g) + 1e-10
    return d, demand, tau, eta


@pytest.mark.parametrize("name", names("g1_cvrp"))
def test_cvrp_golden(name):
    from deepaco_amd.cvrp.aco import ACO
    g = load_golden(name)
    A = g["paths"].shape[1]
    aco = ACO(T(g["distances"]), T(g["demand"]), n_ants=A, heuristic=T(g["heuristic"]), pheromone=T(g["pheromone"]),
              capacity=float(g["capacity"]), device="cuda:0")
    paths, logp = aco.gen_path(True, _noise=T(g["noise"]))
    assert np.array_equal(paths.cpu().numpy(), g["paths"])
    np.testing.assert_allclose(logp.cpu().numpy(), g["log_probs"], atol=2e-6, rtol=1e-5)
    costs = aco.gen_path_costs(paths)
    np.testing.assert_allclose(costs.cpu().numpy(), g["costs"], rtol=1e-5)
    assert np.array_equal(costs.cpu().numpy(), oracle.tour_costs(g["distances"], g["paths"], closed=False))
    # directed deposit (AS and elitist) bitwise vs the reference, fed the reference's costs
    aco.update_pheronome(paths, T(g["costs"]))
    assert np.array_equal(aco.pheromone.cpu().numpy().view(np.uint32), g["pheromone_as"].view(np.uint32))
    el = ACO(T(g["distances"]), T(g["demand"]), n_ants=A, pheromone=T(g["pheromone"]), elitist=True,
             capacity=float(g["capacity"]), device="cuda:0")
    el.update_pheronome(paths, T(g["costs"]))
    assert np.array_equal(el.pheromone.cpu().numpy().view(np.uint32), g["pheromone_elitist"].view(np.uint32))


@pytest.mark.parametrize("mode", ["scan", "race"])
@pytest.mark.parametrize("n,A,B,cap", [(5, 3, 1, 50), (20, 9, 2, 50), (63, 5, 1, 30), (64, 5, 1, 50), (100, 17, 2, 50),
                                        (128, 4, 1, 50), (200, 6, 1, 40), (300, 4, 1, 50), (500, 3, 1, 60),
                                        # every chunk count of the sixteen- / eight-ants-per-wavefront layouts, ragged ant counts
                                        (17, 35, 1, 30), (40, 18, 1, 30), (50, 65, 1, 40), (70, 70, 1, 40), (90, 19, 2, 50), (113, 16, 1, 50),
                                        (130, 9, 1, 50), (170, 33, 1, 50), (225, 5, 1, 50), (256, 12, 1, 50)])
def test_cvrp_philox_bit_exact_vs_oracle(mode, n, A, B, cap):
    from deepaco_amd import engine
    d, demand, tau, eta = cvrp_instance(n, 50 + n, B)
    seed, it, gid0 = 987654321, 2, 77
    paths, logp, _, lens, flags = engine.cvrp_sample(tau.to(dev()), eta.to(dev()), demand.to(dev()), cap, A, mode=mode,
                                                  seed=seed, it=it, ant_gid0=gid0, require_prob=True)
    assert int(flags.sum()) == 0
    for b in range(B):
        P = oracle.prob_matrix(tau[b].numpy(), eta[b].numpy())
        rp, rl, L = oracle.cvrp_sample_rng(P, demand[b].numpy(), cap, A, mode, seed, it, gid0 + b * A, require_prob=True)
        assert L == int(lens[b].max())
        assert np.array_equal(paths[b, :L].cpu().numpy(), rp), (mode, n, b)
        assert bool((paths[b, L:] == 0).all())
        np.testing.assert_allclose(logp[b, :L - 1].cpu().numpy(), rl, atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("mode", ["scan", "race"])
def test_cvrp_full_size_properties(mode):
    """CVRP-100 (config 4 shape): feasibility of every route, costs, deposit vs oracle."""
    from deepaco_amd import engine
    B, n, A, cap = 4, 100, 512, 50.0
    d, demand, tau, eta = cvrp_instance(n, 4242, B)
    D, DM, TA, ET = d.to(dev()), demand.to(dev()), tau.to(dev()), eta.to(dev())
    paths, _, _, lens, flags = engine.cvrp_sample(TA, ET, DM, cap, A, mode=mode, seed=3, it=0)
    assert int(flags.sum()) == 0
    L = int(lens.max())
    p = paths[:, :L]
    # starts and ends at the depot; every customer exactly once
    assert bool((p[:, 0] == 0).all()) and bool((p[:, -1] == 0).all())
    counts = torch.zeros(B, A, n + 1, device=dev()).scatter_add_(2, p.transpose(1, 2), torch.ones(B, A, L, device=dev()))
    assert bool((counts[:, :, 1:] == 1).all())
    # capacity respected on every route: running load resets at the depot
    dm = torch.gather(DM.unsqueeze(1).expand(B, A, n + 1), 2, p.transpose(1, 2))      # [B,A,L]
    load = torch.zeros(B, A, device=dev())
    worst = torch.zeros(B, A, device=dev())
    for k in range(L):
        at_depot = p[:, k] == 0
        load = torch.where(at_depot, torch.zeros_like(load), load + dm[:, :, k])
        worst = torch.maximum(worst, load)
    assert float(worst.max()) <= cap
    # no two consecutive depot visits before the ant is done
    inner = (p[:, :-1] == 0) & (p[:, 1:] == 0)
    ks = torch.arange(L - 1, device=dev()).view(1, L - 1, 1)
    assert bool((~inner | (ks >= (lens.unsqueeze(1) - 1))).all())
    costs = engine.tour_costs(D, p.contiguous(), closed=False)
    ref = torch.stack([D[b][p[b, :-1], p[b, 1:]].double().sum(0) for b in range(B)])
    torch.testing.assert_close(costs.double(), ref, rtol=1e-5, atol=0)
    t2 = TA.clone().contiguous()
    engine.pheromone_update_(t2, p.contiguous(), costs, 0.9, symmetric=False, floor=1e-10)
    b = 2
    rt = oracle.pheromone_update_cvrp(tau[b].numpy(), p[b].cpu().numpy(), costs[b].cpu().numpy(), 0.9)
    assert np.array_equal(t2[b].cpu().numpy().view(np.uint32), rt.view(np.uint32))
    # one instance bit-exact against the oracle at full size
    P = oracle.prob_matrix(tau[b].numpy(), eta[b].numpy())
    rp, _, Lb = oracle.cvrp_sample_rng(P, demand[b].numpy(), cap, 64, mode, 3, 0, b * A)
    assert np.array_equal(paths[b, :Lb, :64].cpu().numpy(), rp)


@pytest.mark.parametrize("n,A,mode,elitist", [(100, 33, "scan", False), (100, 16, "scan", True), (40, 9, "scan", False),
                                              (100, 12, "race", False), (300, 7, "scan_wave", False)])
def test_cvrp_fused_costs_and_table_equal_separate_passes(n, A, mode, elitist):
    """The sampler's fused route costs and successor table give the same costs and the same pheromone as
    daco_tour_costs + the table rebuilt from the paths (both kernel layouts, any ant count)."""
    from deepaco_amd import engine
    B = 3
    d, demand, tau, eta = cvrp_instance(n, 9 * n + A, B)
    D, DM, TA, ET = d.to(dev()), demand.to(dev()), tau.to(dev()), eta.to(dev())
    paths, _, _, lens, flags, costs, table = engine.cvrp_sample(TA, ET, DM, 50.0, A, mode=mode, seed=5, it=3, dist=D,
                                                                want_table=True)
    assert int(flags.sum()) == 0
    p2, _, _, lens2, _ = engine.cvrp_sample(TA, ET, DM, 50.0, A, mode=mode, seed=5, it=3)
    assert torch.equal(paths, p2) and torch.equal(lens, lens2)
    pt = paths[:, :int(lens.max())].contiguous()
    assert torch.equal(costs, engine.tour_costs(D, pt, closed=False))
    t1 = TA.clone().contiguous()
    engine.pheromone_update_(t1, paths, costs, 0.9, elitist, False, floor=1e-10, nbr=table)
    for b in range(B):       # the reference pads each colony's routes to that colony's longest one
        t2 = TA[b:b + 1].clone().contiguous()
        pb = paths[b:b + 1, :int(lens[b].max())].contiguous()
        engine.pheromone_update_(t2, pb, costs[b:b + 1], 0.9, elitist, False, floor=1e-10)
        assert torch.equal(t1[b], t2[0]), b
    col = engine.BatchedCVRP(D, DM, n_ants=A, capacity=50, seed=11, sampler=mode, elitist=elitist)
    col.run(3)
    col.check_feasible()
    assert bool((col.lowest_cost > 0).all())
    # the recorded best route really has the recorded cost
    best = engine.tour_costs(D, col.shortest_path.unsqueeze(2).contiguous(), closed=False)[:, 0]
    torch.testing.assert_close(best, col.lowest_cost, rtol=1e-6, atol=0)


def test_cvrp_class_run():
    from deepaco_amd.cvrp.aco import ACO
    d, demand, _, _ = cvrp_instance(50, 9)
    aco = ACO(d[0].to(dev()), demand[0].to(dev()), n_ants=32, device="cuda:0")
    costs, logp = aco.sample()
    assert costs.shape == (32,) and logp.shape[1] == 32
    first = float(costs.min())
    low = aco.run(5)
    assert float(low) <= first * 1.2 and aco.shortest_path[0] == 0
    with pytest.raises(NotImplementedError):
        ACO(d[0].to(dev()), demand[0].to(dev()), adaptive=True)


@pytest.mark.parametrize("kw", [{}, {"elitist": True}, {"min_max": True}])
@pytest.mark.parametrize("n", [30, 100])
def test_cvrp_sync_free_run_equals_plain_sequence(kw, n):
    """ACO.run (fused costs / successor table, device-side bookkeeping) == the reference's call sequence
    gen_path -> gen_path_costs -> update_pheronome, pheromone bit for bit."""
    from deepaco_amd.cvrp.aco import ACO
    d, demand, _, _ = cvrp_instance(n, 5 * n)
    a1 = ACO(d[0].to(dev()), demand[0].to(dev()), n_ants=24, device="cuda:0", seed=8, **kw)
    a2 = ACO(d[0].to(dev()), demand[0].to(dev()), n_ants=24, device="cuda:0", seed=8, **kw)
    r1, r2 = a1.run(6), a2._run_plain(6)
    assert float(r1) == float(r2)
    assert torch.equal(a1.pheromone, a2.pheromone)
    s1, s2 = a1.shortest_path.tolist(), a2.shortest_path.tolist()
    while len(s2) > 1 and s2[-1] == 0 and s2[-2] == 0:      # the reference keeps the iteration's padding
        s2.pop()
    assert s1 == s2


def test_cvrp_nls_surface_float64_data():
    """cvrp_nls instances are float64 with capacity 1.0 (cvrp_nls/utils.py:19-30); same sampler."""
    from deepaco_amd.cvrp_nls.aco import ACO
    d, demand, _, _ = cvrp_instance(40, 31)
    cap = 1.0
    dem = (demand[0] / 50.0).double()
    aco = ACO(d[0].double().to(dev()), dem.to(dev()), n_ants=16, device="cuda:0", capacity=cap, seed=2)
    costs, logp, paths = aco.sample()
    assert costs.shape == (16,) and paths.shape[1] == 16 and logp.shape[0] == paths.shape[0] - 1
    p = paths.cpu().numpy()
    for a in range(16):
        load, seen = 0.0, set()
        for v in p[:, a]:
            load = 0.0 if v == 0 else load + float(dem[v])
            assert load <= cap + 1e-6
            if v:
                assert v not in seen
                seen.add(int(v))
        assert len(seen) == 40
    assert float(aco.run(3)) <= float(costs.max())
    # swapstar=True: the local search runs on the device (tests/test_gpu_09_cvrp_ls.py); float64 instance data are cast
    ls = ACO(d[0].double().to(dev()), dem.to(dev()), n_ants=16, device="cuda:0", capacity=cap, seed=2, swapstar=True,
             positions=torch.zeros(41, 2))
    c_ls, _, c_raw = ls.sample_nls()
    assert bool((c_ls <= c_raw + 1e-5).all()) and bool((c_ls < c_raw - 1e-4).any())


@pytest.mark.parametrize("name", ["g1f64_cvrp_nls_n20_a8", "g1f64_cvrp_nls_n50_a8", "g1f64_cvrp_nls_n100_a6"])
def test_cvrp_nls_float64_instances_reproduce_the_reference_routes(name):
    """cvrp_nls/ keeps demands and distances in float64 and its load bookkeeping (cvrp_nls/aco.py:254-272) therefore
    runs in double; with demands k / 50 an exactly fitting customer is common and float32 bookkeeping decides 5-10 % of
    the steps of these fixtures differently.  With the float64 demands handed to the drop-in the routes drawn on the
    recorded noise are the reference's, log-probabilities and costs to float32 rounding; and the gradient path replays
    the same rule."""
    from deepaco_amd.cvrp_nls.aco import ACO
    g = load_golden(name)
    A = g["paths"].shape[1]
    heu = T(g["heuristic"]).requires_grad_(True)
    aco = ACO(T(g["distances"]), T(g["demand"]), n_ants=A, heuristic=heu, pheromone=T(g["pheromone"]), device="cuda:0")
    assert aco.demand.dtype == torch.float64
    noise = T(g["noise"].astype(np.float32))
    paths, logp = aco.gen_path(True, _noise=noise)
    assert np.array_equal(paths.cpu().numpy(), g["paths"])
    np.testing.assert_allclose(logp.detach().cpu().numpy(), g["log_probs"], atol=3e-6, rtol=2e-5)
    np.testing.assert_allclose(aco.gen_path_costs(paths).cpu().numpy(), g["costs"], rtol=1e-5)
    # the float32 image of the same instance takes other routes (what the float64 path is for)
    aco32 = ACO(T(g["distances"]), T(g["demand"].astype(np.float32)), n_ants=A, heuristic=heu.detach(),
                pheromone=T(g["pheromone"]), device="cuda:0")
    p32 = aco32.gen_path(False, _noise=noise)
    L = min(p32.shape[0], paths.shape[0])
    assert not torch.equal(p32[:L], paths[:L])
    # gradient: closed form on the reference's routes with the float64 capacity rule (oracle), float32 elsewhere
    loss = (logp.sum(0) * torch.linspace(-1.0, 1.0, A, device=logp.device)).sum()
    loss.backward()
    assert bool(torch.isfinite(heu.grad).all()) and float(heu.grad.abs().max()) > 0
    assert heu.grad.shape == (len(g["demand"]), len(g["demand"]))


@pytest.mark.parametrize("n,A,B", [(21, 18, 1), (51, 65, 1), (101, 33, 2), (128, 9, 1), (201, 12, 1), (256, 5, 1), (301, 4, 1), (512, 3, 1), (600, 2, 1)])
def test_cvrp_float64_bookkeeping_in_the_scan_kernels_equals_the_oracle(n, A, B):
    """cvrp_nls/ keeps demands in float64 (k / 50 of capacity 1: exact fits are common): with a float64 `demand` the capacity
    mask is decided in double (cvrp_nls/aco.py:254-272) -- on every packed scan layout (4 / 8 / 16 lanes per ant, n <= 512; 600:
    the one-ant-per-wavefront kernel).  Routes and log-probabilities against the oracle's float64 variant; the float32 image of the
    same demands takes other routes somewhere in the batch (what the float64 path is for)."""
    from deepaco_amd import engine
    g = torch.Generator().manual_seed(900 + n)
    tau = torch.rand(B, n, n, generator=g) + 0.1
    eta = torch.rand(B, n, n, generator=g) + 1e-10
    dem64 = torch.cat((torch.zeros(B, 1, dtype=torch.float64),
                       torch.randint(1, 10, (B, n - 1), generator=g).double() / 50.0), 1)
    seed, it = 24680, 3
    paths, logp, _, lens, flags = engine.cvrp_sample(tau.to(dev()), eta.to(dev()), dem64.to(dev()), 1.0, A, mode="scan",
                                                     seed=seed, it=it, require_prob=True)
    assert int(flags.sum()) == 0
    differs = False
    for b in range(B):
        P = oracle.prob_matrix(tau[b].numpy(), eta[b].numpy())
        rp, rl, L = oracle.cvrp_sample_rng(P, dem64[b].numpy(), 1.0, A, "scan", seed, it, ant_gid0=b * A, require_prob=True)
        assert L == int(lens[b].max())
        assert np.array_equal(paths[b, :L].cpu().numpy(), rp), (n, b)
        np.testing.assert_allclose(logp[b, :L - 1].cpu().numpy(), rl, atol=3e-6, rtol=1e-5)
        p32, _, L32 = oracle.cvrp_sample_rng(P, dem64[b].numpy().astype(np.float32), 1.0, A, "scan", seed, it, ant_gid0=b * A)
        differs = differs or L32 != L or not np.array_equal(p32, rp)
    assert differs or n < 30
