"""GPU parity tests for the sibling problems (S1-S6): with the reference's recorded noise every class
rebuilds the reference's solutions exactly; objectives and the pheromone update match."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev())


def noise_list(g):
    return [T(q) for q in g["noise"]]


def check_update(aco, fn, sols, g, key="pheromone_as"):
    fn()
    np.testing.assert_allclose(aco.pheromone.cpu().numpy(), g[key], rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize("fix", ["s4_smtwtp_n20", "s4_smtwtp_n50"])
def test_smtwtp(fix):
    from deepaco_amd.smtwtp.aco import ACO
    g = load_golden(fix)
    A = g["paths"].shape[1]
    aco = ACO(T(g["due_time"]), T(g["weights"]), T(g["processing_time"]), n_ants=A, pheromone=T(g["pheromone"]),
              device="cuda:0")
    np.testing.assert_array_equal(aco.heuristic.cpu().numpy(), g["heuristic"])
    paths, logp = aco.gen_path(True, _noise=noise_list(g))
    assert np.array_equal(paths.cpu().numpy(), g["paths"])
    np.testing.assert_allclose(logp.cpu().numpy(), g["log_probs"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(aco.gen_path_costs(paths).cpu().numpy(), g["costs"], rtol=1e-5)
    aco.update_pheronome(paths, T(g["costs"]))
    assert np.array_equal(aco.pheromone.cpu().numpy().view(np.uint32), g["pheromone_as"].view(np.uint32))
    el = ACO(T(g["due_time"]), T(g["weights"]), T(g["processing_time"]), n_ants=A, pheromone=T(g["pheromone"]),
             elitist=True, device="cuda:0")
    el.update_pheronome(paths, T(g["costs"]))
    assert np.array_equal(el.pheromone.cpu().numpy().view(np.uint32), g["pheromone_elitist"].view(np.uint32))
    low = aco.run(3)
    assert float(low) <= float(g["costs"].max())
    # fused (one launch) and draw-by-draw paths share the Philox counters: identical sequences
    for sampler in ("scan", "race"):
        a1 = ACO(T(g["due_time"]), T(g["weights"]), T(g["processing_time"]), n_ants=32, device="cuda:0",
                 sampler=sampler, seed=11)
        a2 = ACO(T(g["due_time"]), T(g["weights"]), T(g["processing_time"]), n_ants=32, device="cuda:0",
                 sampler=sampler, seed=11)
        p1, l1 = a1.gen_path(True)
        p2, l2 = a2.gen_path(True, _stepwise=True)
        assert torch.equal(p1, p2)
        torch.testing.assert_close(l1, l2, rtol=1e-5, atol=2e-6)
    fresh = ACO(T(g["due_time"]), T(g["weights"]), T(g["processing_time"]), n_ants=A, pheromone=T(g["pheromone"]),
                device="cuda:0")
    pn, _ = fresh.gen_path(True, _noise=noise_list(g), _stepwise=True)
    assert np.array_equal(pn.cpu().numpy(), g["paths"])


@pytest.mark.parametrize("fix", ["s3_sop_n20", "s3_sop_n50"])
def test_sop(fix):
    from deepaco_amd.sop.aco import ACO
    g = load_golden(fix)
    A = g["paths"].shape[1]
    aco = ACO(T(g["distances"]), T(g["prec_cons"]), n_ants=A, pheromone=T(g["pheromone"]), device="cuda:0")
    paths, logp = aco.gen_path(True, _noise=noise_list(g))
    assert np.array_equal(paths.cpu().numpy(), g["paths"])
    np.testing.assert_allclose(logp.cpu().numpy(), g["log_probs"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(aco.gen_path_costs(paths).cpu().numpy(), g["costs"], rtol=1e-5)
    aco.update_pheronome(paths, T(g["costs"]))
    assert np.array_equal(aco.pheromone.cpu().numpy().view(np.uint32), g["pheromone_as"].view(np.uint32))
    # precedence holds for freshly sampled paths in both samplers
    for sampler in ("scan", "race"):
        a2 = ACO(T(g["distances"]), T(g["prec_cons"]), n_ants=64, device="cuda:0", sampler=sampler, seed=4)
        p = a2.gen_path().cpu().numpy()
        pos = np.argsort(p, axis=0)                       # pos[node, ant]
        jj, kk = np.nonzero(g["prec_cons"])
        assert (pos[kk] < pos[jj]).all() and (np.sort(p, axis=0) == np.arange(p.shape[0])[:, None]).all()


@pytest.mark.parametrize("fix", ["s2_pctsp_n20", "s2_pctsp_n100"])
def test_pctsp(fix):
    from deepaco_amd.pctsp.aco import ACO
    g = load_golden(fix)
    A = g["sols"].shape[1]
    aco = ACO(T(g["distances"]), T(g["prizes"]), T(g["penalties"]), n_ants=A, device="cuda:0")
    np.testing.assert_allclose(aco.heuristic.cpu().numpy(), g["heuristic"], rtol=1e-6)
    aco.heuristic, aco.pheromone = T(g["heuristic"]), T(g["pheromone"])
    sols, logp = aco.gen_sol(True, _noise=noise_list(g))
    assert np.array_equal(sols.cpu().numpy(), g["sols"])
    np.testing.assert_allclose(logp.cpu().numpy(), g["log_probs"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(aco.gen_sol_obj(sols).cpu().numpy(), g["objs"], rtol=1e-5)
    objs = T(g["objs"])
    best_obj, best_idx = objs.max(dim=0)
    aco.update_pheronome(sols.T, objs, best_obj, best_idx)
    assert np.array_equal(aco.pheromone.cpu().numpy().view(np.uint32), g["pheromone_as"].view(np.uint32))
    best, sol = aco.run(2)
    assert sol[0] == 0 and float(best) > 0


@pytest.mark.parametrize("fix", ["s1_op_n30", "s1_op_n100"])
def test_op(fix):
    from deepaco_amd.op.aco import ACO
    g = load_golden(fix)
    A = g["sols"].shape[1]
    aco = ACO(T(g["distances_in"]), T(g["prizes_in"]), float(g["max_len"]), n_ants=A, k_sparse=int(g["k_sparse"]),
              device="cuda:0")
    np.testing.assert_array_equal(aco.distances.cpu().numpy(), g["distances"])
    np.testing.assert_array_equal(aco.prizes.cpu().numpy(), g["prizes"])
    np.testing.assert_allclose(aco.heuristic.cpu().numpy(), g["heuristic"], rtol=1e-6)
    aco.heuristic = T(g["heuristic"])
    np.testing.assert_allclose(float(aco.Q), float(g["Q"]), rtol=1e-6)
    aco.Q = T(g["Q"])                                     # (a GPU sum may round differently by one ulp)
    sols, logp = aco.gen_sol(True, _noise=noise_list(g))
    assert np.array_equal(sols.cpu().numpy(), g["sols"])
    np.testing.assert_allclose(logp.cpu().numpy(), g["log_probs"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(aco.gen_sol_obj(sols).cpu().numpy(), g["objs"], rtol=1e-6)
    objs = T(g["objs"])
    best_obj, best_idx = objs.max(dim=0)
    aco.update_pheronome(sols.T, objs, best_obj, best_idx)
    assert np.array_equal(aco.pheromone.cpu().numpy().view(np.uint32), g["pheromone_as"].view(np.uint32))
    # fresh solutions respect the length budget (route + way back to the depot)
    a2 = ACO(T(g["distances_in"]), T(g["prizes_in"]), float(g["max_len"]), n_ants=64, k_sparse=int(g["k_sparse"]),
             device="cuda:0", seed=3)
    s = a2.gen_sol()
    d = a2.distances
    real = s.clone()
    length = torch.zeros(64, device=dev())
    last = torch.zeros(64, dtype=torch.long, device=dev())
    for k in range(1, s.shape[0]):
        step = real[k]
        move = step != a2.n
        length = length + torch.where(move, d[last, step], torch.zeros_like(length))
        last = torch.where(move, step, last)
    back = torch.where(last != 0, d[last, torch.zeros_like(last)], torch.zeros_like(length))
    assert float((length + back).max()) <= float(g["max_len"]) + 1e-4


@pytest.mark.parametrize("fix", ["s5_bpp_n24", "s5_bpp_n120"])
def test_bpp(fix):
    from deepaco_amd.bpp.aco import ACO
    g = load_golden(fix)
    A = g["paths"].shape[1]
    aco = ACO(T(g["demand"]), n_ants=A, pheromone=T(g["pheromone"]), device="cuda:0")
    np.testing.assert_array_equal(aco.heuristic.cpu().numpy(), g["heuristic"])
    paths, logp = aco.gen_path(True, _noise=noise_list(g))
    assert np.array_equal(paths.cpu().numpy(), g["paths"])
    np.testing.assert_allclose(logp.cpu().numpy(), g["log_probs"], atol=2e-6, rtol=1e-5)
    costs = aco.gen_path_costs(paths)
    np.testing.assert_allclose(costs.cpu().numpy(), g["costs"], rtol=1e-12)
    aco.update_pheronome(paths, -T(g["costs"]))
    np.testing.assert_allclose(aco.pheromone.cpu().numpy(), g["pheromone_as"], rtol=1e-6)
    fit = aco.run(3)
    assert 0 < float(fit) <= 1


@pytest.mark.parametrize("fix", ["s6_mkp_n20", "s6_mkp_n50"])
def test_mkp(fix):
    from deepaco_amd.mkp.aco import ACO
    g = load_golden(fix)
    A = g["sols"].shape[1]
    aco = ACO(T(g["prize_in"]), T(g["weight_in"]), n_ants=A, pheromone=T(g["pheromone"]), device="cuda:0")
    np.testing.assert_allclose(aco.heuristic.cpu().numpy(), g["heuristic"], rtol=1e-6)
    aco.heuristic = T(g["heuristic"])
    aco.Q = T(g["Q"])
    sols, logp = aco.gen_sol(True, _noise=noise_list(g), _start=T(g["start"]))
    assert np.array_equal(sols.cpu().numpy(), g["sols"])
    np.testing.assert_allclose(logp.cpu().numpy(), g["log_probs"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(aco.gen_sol_obj(sols).cpu().numpy(), g["objs"], rtol=1e-6)
    objs = T(g["objs"])
    best_obj, best_idx = objs.max(dim=0)
    aco.update_pheronome(sols.T, objs, best_obj.item(), best_idx.item())
    assert np.array_equal(aco.pheromone.cpu().numpy().view(np.uint32), g["pheromone_as"].view(np.uint32))
    # fresh solutions are feasible in every knapsack dimension
    a2 = ACO(T(g["prize_in"]), T(g["weight_in"]), n_ants=64, device="cuda:0", seed=8)
    s = a2.gen_sol().T
    used = a2.weight[s].sum(dim=1)
    assert float(used.max()) <= a2.n // 2 + 1e-5
    best, _ = a2.run(2)
    assert float(best) > 0


def test_sibling_gradients_flow():
    from deepaco_amd.smtwtp.aco import ACO
    g = load_golden("s4_smtwtp_n20")
    heu = T(g["heuristic"]).requires_grad_(True)
    aco = ACO(T(g["due_time"]), T(g["weights"]), T(g["processing_time"]), n_ants=8, heuristic=heu, device="cuda:0")
    costs, logp = aco.sample()
    loss = torch.sum((costs - costs.mean()) * logp.sum(dim=0)) / 8
    loss.backward()
    assert heu.grad is not None and bool(torch.isfinite(heu.grad).all()) and float(heu.grad.abs().sum()) > 0


# ------------------------------------------------------------------ fused one-launch constructions
def _both_paths(make, gen_name, A, **kw):
    """The fused kernel and the draw-by-draw service share Philox counters -> identical solutions."""
    for sampler in ("scan", "race"):
        a1, a2 = make(sampler), make(sampler)
        s1, l1 = getattr(a1, gen_name)(True, **kw)
        s2, l2 = getattr(a2, gen_name)(True, _stepwise=True, **kw)
        assert s1.shape == s2.shape, (sampler, s1.shape, s2.shape)
        assert torch.equal(s1, s2), sampler
        torch.testing.assert_close(l1, l2, rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("fx", [("s3_sop_n20", "s2_pctsp_n20", "s1_op_n30", "s6_mkp_n20"), ("s3_sop_n50", "s2_pctsp_n100", "s1_op_n100", "s6_mkp_n50")])
def test_stepwise_service_still_matches_reference(fx):
    """The draw-by-draw path (daco_pick_move) against the fixtures, now that the default is fused."""
    from deepaco_amd.sop.aco import ACO as SOP
    from deepaco_amd.pctsp.aco import ACO as PCTSP
    from deepaco_amd.op.aco import ACO as OP
    from deepaco_amd.mkp.aco import ACO as MKP
    g = load_golden(fx[0])
    a = SOP(T(g["distances"]), T(g["prec_cons"]), n_ants=8, pheromone=T(g["pheromone"]), device="cuda:0")
    assert np.array_equal(a.gen_path(True, _noise=noise_list(g), _stepwise=True)[0].cpu().numpy(), g["paths"])
    g = load_golden(fx[1])
    a = PCTSP(T(g["distances"]), T(g["prizes"]), T(g["penalties"]), n_ants=8, device="cuda:0")
    a.heuristic, a.pheromone = T(g["heuristic"]), T(g["pheromone"])
    assert np.array_equal(a.gen_sol(True, _noise=noise_list(g), _stepwise=True)[0].cpu().numpy(), g["sols"])
    g = load_golden(fx[2])
    a = OP(T(g["distances_in"]), T(g["prizes_in"]), float(g["max_len"]), n_ants=8, k_sparse=int(g["k_sparse"]),
           device="cuda:0")
    a.heuristic = T(g["heuristic"])
    assert np.array_equal(a.gen_sol(True, _noise=noise_list(g), _stepwise=True)[0].cpu().numpy(), g["sols"])
    g = load_golden(fx[3])
    a = MKP(T(g["prize_in"]), T(g["weight_in"]), n_ants=8, pheromone=T(g["pheromone"]), device="cuda:0")
    a.heuristic = T(g["heuristic"])
    assert np.array_equal(a.gen_sol(True, _noise=noise_list(g), _start=T(g["start"]), _stepwise=True)[0].cpu().numpy(),
                          g["sols"])


@pytest.mark.parametrize("n", [20, 70, 150])
def test_fused_sop_equals_stepwise(n):
    from deepaco_amd.sop.aco import ACO
    gen = torch.Generator().manual_seed(n)
    dist = torch.rand(n, n, generator=gen) + 0.05
    prec = torch.zeros(n, n)
    order = torch.randperm(n - 1, generator=gen) + 1                 # a hidden feasible order
    for _ in range(2 * n):
        i, j = sorted(torch.randint(0, n - 1, (2,), generator=gen).tolist())
        if i != j:
            prec[order[j], order[i]] = 1                              # order[i] precedes order[j]
    prec[1:, 0] = 1
    _both_paths(lambda s: ACO(dist.to(dev()), prec.to(dev()), n_ants=48, device="cuda:0", sampler=s, seed=3),
                "gen_path", 48)


@pytest.mark.parametrize("n", [21, 101])
def test_fused_pctsp_equals_stepwise(n):
    from deepaco_amd.pctsp.aco import ACO
    gen = torch.Generator().manual_seed(n)
    coor = torch.rand(n, 2, generator=gen)
    dist = torch.cdist(coor, coor)
    prizes = torch.cat((torch.zeros(1), torch.rand(n - 1, generator=gen)))
    pen = torch.cat((torch.zeros(1), torch.rand(n - 1, generator=gen) * 0.3))
    _both_paths(lambda s: ACO(dist.to(dev()), prizes.to(dev()), pen.to(dev()), n_ants=40, device="cuda:0", sampler=s,
                              seed=5), "gen_sol", 40)


@pytest.mark.parametrize("n,max_len", [(30, 3.0), (100, 4.0)])
def test_fused_op_equals_stepwise(n, max_len):
    from deepaco_amd.op.aco import ACO
    from deepaco_amd.tsp.utils import gen_distance_matrix
    gen = torch.Generator().manual_seed(n)
    coor = torch.rand(n, 2, generator=gen)
    dist = gen_distance_matrix(coor)
    dd = (coor - coor[0]).norm(dim=-1)
    prizes = 1 + torch.floor(99 * dd / dd.max())
    prizes = prizes / prizes.max()
    _both_paths(lambda s: ACO(dist.to(dev()), prizes.to(dev()), max_len, n_ants=40, k_sparse=max(5, n // 5),
                              device="cuda:0", sampler=s, seed=6), "gen_sol", 40)


@pytest.mark.parametrize("n,m", [(20, 3), (50, 5)])
def test_fused_mkp_equals_stepwise(n, m):
    from deepaco_amd.mkp.aco import ACO
    gen = torch.Generator().manual_seed(n)
    prize = torch.rand(n, generator=gen)
    w = torch.rand(n, m, generator=gen)
    cons = w.max(0).values + torch.rand(m, generator=gen) * (w.sum(0) - w.max(0).values)
    w = w * (n // 2) / cons.unsqueeze(0)
    start = torch.randint(0, n, (40,), generator=gen).to(dev())
    _both_paths(lambda s: ACO(prize.to(dev()), w.to(dev()), n_ants=40, device="cuda:0", sampler=s, seed=7), "gen_sol",
                40, _start=start)
    # Philox start nodes inside the kernel are valid items
    a = ACO(prize.to(dev()), w.to(dev()), n_ants=256, device="cuda:0", seed=1)
    s = a.gen_sol()
    assert int(s[0].min()) >= 0 and int(s[0].max()) < n and len(set(s[0].tolist())) > 5
