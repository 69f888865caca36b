"""GPU tests of the training path: ACO.sample() returns log-probs that carry gradient to the
heuristic (tsp/train.ipynb:32-49, cvrp/train.ipynb:32-51), via daco_sample_backward."""
import numpy as np
import pytest
import torch

import oracle
from oracle import grad as ograd
from conftest import load_golden

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev())


@pytest.mark.parametrize("name", ["g3_grad_tsp_n20_a8", "g3_grad_tsp_n40_a10_beta2"])
def test_reinforce_grad_matches_reference_tsp(name):
    from deepaco_amd.tsp.aco import ACO
    g = load_golden(name)
    A = g["paths"].shape[1]
    heu = T(g["heuristic"]).requires_grad_(True)
    aco = ACO(T(g["distances"]), n_ants=A, heuristic=heu, pheromone=T(g["pheromone"]), beta=float(g["beta"]),
              device="cuda:0")
    paths, logp = aco.gen_path(True, _start=T(g["start"]), _noise=T(g["noise"]))
    assert np.array_equal(paths.cpu().numpy(), g["paths"])
    costs = aco.gen_path_costs(paths)
    loss = torch.sum((costs - costs.mean()) * logp.sum(dim=0)) / A        # tsp/train.ipynb:45-47
    loss.backward()
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=2e-4, atol=1e-5)
    scale = np.abs(g["grad"]).max()
    np.testing.assert_allclose(heu.grad.cpu().numpy(), g["grad"], rtol=3e-4, atol=3e-6 * scale)


def test_reinforce_grad_matches_reference_cvrp():
    from deepaco_amd.cvrp.aco import ACO
    g = load_golden("g3_grad_cvrp_n20_a8")
    A = g["paths"].shape[1]
    heu = T(g["heuristic"]).requires_grad_(True)
    aco = ACO(T(g["distances"]), T(g["demand"]), n_ants=A, heuristic=heu, capacity=float(g["capacity"]),
              device="cuda:0")
    paths, logp = aco.gen_path(True, _noise=T(g["noise"]))
    assert np.array_equal(paths.cpu().numpy(), g["paths"])
    costs = aco.gen_path_costs(paths)
    loss = torch.sum((costs - costs.mean()) * logp.sum(dim=0)) / A
    loss.backward()
    scale = np.abs(g["grad"]).max()
    np.testing.assert_allclose(heu.grad.cpu().numpy(), g["grad"], rtol=3e-4, atol=3e-6 * scale)


@pytest.mark.parametrize("mode", ["scan", "race"])
@pytest.mark.parametrize("n,A,beta", [(30, 6, 1), (100, 8, 1), (150, 5, 2), (300, 4, 1)])
def test_grad_vs_closed_form_philox(mode, n, A, beta):
    from deepaco_amd.tsp.aco import ACO
    g = torch.Generator().manual_seed(n)
    c = torch.rand(n, 2, generator=g)
    d = torch.cdist(c, c)
    d[torch.arange(n), torch.arange(n)] = 1e9
    tau = torch.rand(n, n, generator=g) + 0.2
    eta = (torch.rand(n, n, generator=g) + 1e-2)
    heu = eta.to(dev()).requires_grad_(True)
    aco = ACO(d.to(dev()), n_ants=A, heuristic=heu, pheromone=tau.to(dev()), beta=beta, device="cuda:0",
              sampler=mode, seed=5)
    costs, logp = aco.sample()
    w = torch.linspace(-1, 1, A, device=dev())
    (logp.sum(0) * w).sum().backward()
    # the tours the forward drew (same seed / iteration), then the closed form on them
    paths = ACO(d.to(dev()), n_ants=A, heuristic=eta.to(dev()), pheromone=tau.to(dev()), beta=beta, device="cuda:0",
                sampler=mode, seed=5).gen_path()
    G = np.tile(w.cpu().numpy()[None, :], (n - 1, 1))
    ref = ograd.tsp_grad(tau.numpy(), eta.numpy(), 1, beta, paths.cpu().numpy(), G)
    scale = np.abs(ref).max()
    np.testing.assert_allclose(heu.grad.cpu().numpy(), ref, rtol=3e-4, atol=3e-6 * scale)


def test_no_grad_path_unchanged():
    """Without requires_grad (or under no_grad) sample() takes the plain path and returns the same tours."""
    from deepaco_amd.tsp.aco import ACO
    n, A = 60, 8
    g = torch.Generator().manual_seed(2)
    c = torch.rand(n, 2, generator=g)
    d = torch.cdist(c, c)
    d[torch.arange(n), torch.arange(n)] = 1e9
    eta = torch.rand(n, n, generator=g) + 1e-2
    a1 = ACO(d.to(dev()), n_ants=A, heuristic=eta.to(dev()).requires_grad_(True), device="cuda:0", seed=3)
    a2 = ACO(d.to(dev()), n_ants=A, heuristic=eta.to(dev()), device="cuda:0", seed=3)
    p1, l1 = a1.gen_path(True)
    p2, l2 = a2.gen_path(True)
    assert torch.equal(p1, p2) and torch.equal(l1.detach(), l2) and l1.requires_grad and not l2.requires_grad
