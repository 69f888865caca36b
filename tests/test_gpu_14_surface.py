"""The helper methods of the reference's class surface (VERDICT r4 missing #4), each against a reference-captured fixture:
ACO.pick_move (tsp/aco.py:165-177), the CVRP step-wise helpers (cvrp/aco.py:167-205), tsp_nls's numpy helpers and
inference_batch_sample (tsp_nls/aco.py:171-182, 222-228, 260-297)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev())


def test_tsp_gen_path_rebuilt_from_pick_move_calls():
    """The reference's gen_path loop (tsp/aco.py:134-163) written with the drop-in's pick_move and the reference's recorded
    noise: the reference's tours and log-probabilities."""
    from deepaco_amd.tsp.aco import ACO
    g = load_golden("g1_tsp_n20_a8_learned")
    n, A = g["paths"].shape
    aco = ACO(T(g["distances"]), n_ants=A, pheromone=T(g["pheromone"]), heuristic=T(g["heuristic"]), device="cuda:0")
    prev = T(g["start"])
    mask = torch.ones(A, n, device=dev())
    idx = torch.arange(A, device=dev())
    mask[idx, prev] = 0
    tour, lps = [prev], []
    for t in range(n - 1):
        actions, log_probs = aco.pick_move(prev, mask, True, _noise=T(g["noise"][t]))
        tour.append(actions)
        lps.append(log_probs)
        prev = actions
        mask[idx, actions] = 0
    assert np.array_equal(torch.stack(tour).cpu().numpy(), g["paths"])
    np.testing.assert_allclose(torch.stack(lps).cpu().numpy(), g["log_probs"], atol=2e-6, rtol=1e-5)
    a2, lp2 = aco.pick_move(T(g["start"]), torch.ones(A, n, device=dev()), False)          # own random stream, no log-probs
    assert lp2 is None and a2.shape == (A,) and int(a2.min()) >= 0 and int(a2.max()) < n


def test_cvrp_gen_path_rebuilt_from_the_stepwise_helpers():
    """cvrp/aco.py:138-165 written with pick_move / update_visit_mask / update_capacity_mask / check_done on the recorded noise."""
    from deepaco_amd.cvrp.aco import ACO
    g = load_golden("g1_cvrp_n20_a8")
    L, A = g["paths"].shape
    n = g["distances"].shape[0]
    aco = ACO(T(g["distances"]), T(g["demand"]), n_ants=A, pheromone=T(g["pheromone"]), heuristic=T(g["heuristic"]),
              capacity=float(g["capacity"]), device="cuda:0")
    actions = torch.zeros(A, dtype=torch.long, device=dev())
    visit_mask = torch.ones(A, n, device=dev())
    visit_mask = aco.update_visit_mask(visit_mask, actions)
    used = torch.zeros(A, device=dev())
    used, cap_mask = aco.update_capacity_mask(actions, used)
    paths, lps, done, t = [actions], [], False, 0
    while not done:
        actions, lp = aco.pick_move(actions, visit_mask, cap_mask, True, _noise=T(g["noise"][t]))
        paths.append(actions)
        lps.append(lp)
        visit_mask = aco.update_visit_mask(visit_mask, actions)
        used, cap_mask = aco.update_capacity_mask(actions, used)
        done = bool(aco.check_done(visit_mask, actions))
        t += 1
    got = torch.stack(paths).cpu().numpy()
    assert got.shape[0] == L and np.array_equal(got, g["paths"])
    np.testing.assert_allclose(torch.stack(lps).cpu().numpy(), g["log_probs"][: len(lps)], atol=2e-6, rtol=1e-5)


def test_tsp_nls_numpy_helpers_and_inference_sampler():
    from deepaco_amd.tsp_nls.aco import ACO, inference_batch_sample, _inference_sample
    g = load_golden("g1_nls_n20_a8")
    n, A = g["paths"].shape
    aco = ACO(T(g["distances"]), n_ants=A, heuristic=T(g["heuristic"]), device="cuda:0")
    assert aco.distances_numpy.dtype == np.float32 and aco.distances_numpy.shape == (n, n)
    np.testing.assert_array_equal(aco.heuristic_numpy, g["heuristic"].astype(np.float32))
    costs = aco.gen_numpy_path_costs(g["paths"].T, aco.distances_numpy)
    np.testing.assert_allclose(costs, g["costs"], rtol=1e-5)
    prob = (g["pheromone"] * g["heuristic"]).astype(np.float32)
    routes = inference_batch_sample(prob, count=6, startnode=0, seed=5)
    assert routes.dtype == np.uint16 and routes.shape == (6, n) and (routes[:, 0] == 0).all()
    assert (np.sort(routes, axis=1) == np.arange(n)).all()
    again = inference_batch_sample(prob, count=6, startnode=0, seed=5)
    np.testing.assert_array_equal(routes, again)
    rnd = inference_batch_sample(prob, count=32)                                # random start per tour
    assert (np.sort(rnd, axis=1) == np.arange(n)).all() and len(set(rnd[:, 0].tolist())) > 1
    one = _inference_sample(prob, 3)
    assert one.shape == (n,) and one[0] == 3


def test_cvrp_nls_host_copies_of_the_instance():
    """cvrp_nls/aco.py:273-287: distances_cpu / demand_cpu / positions_cpu -- numpy copies in the caller's dtype (float64 instance
    data, cvrp_nls/utils.py:12-32), cached; positions_cpu is None without positions (VERDICT r5 missing 6)."""
    from deepaco_amd.cvrp_nls.aco import ACO
    from deepaco_amd.cvrp_nls.utils import gen_instance
    torch.manual_seed(3)
    demand, dist, pos = gen_instance(20, dev(), position=True)
    aco = ACO(dist, demand, n_ants=4, swapstar=True, positions=pos, device="cuda:0")
    assert aco.distances_cpu.dtype == np.float64 and np.array_equal(aco.distances_cpu, dist.cpu().numpy())
    assert aco.demand_cpu.dtype == np.float64 and np.array_equal(aco.demand_cpu, demand.cpu().numpy())
    assert np.array_equal(aco.positions_cpu, pos.cpu().numpy())
    assert aco.distances_cpu is aco.distances_cpu                   # cached
    assert ACO(dist, demand, n_ants=4, device="cuda:0").positions_cpu is None
