"""torch-CPU port of the reference's TSP colony iteration -- the reference CPU path as it would
run on this host.  TEST INFRASTRUCTURE / bench cpu_baseline only (never imported by deepaco_amd).

The reference's Python cannot travel to the GPU box, so bench.py times this port next to the
HIP path.  It issues the same aten op sequence per step as tsp/aco.py:134-177 (row gathers of
pheromone and heuristic by `prev`, two pows, two muls, Categorical normalise + validate,
multinomial -> exponential_ + div + argmax, mask index_put), :121-132 (costs) and :95-118
(sequential per-ant deposit), so under the same torch build and torch.manual_seed it returns
the reference's tours bit for bit (checked in tests/test_torch_port.py against the golden
fixtures, which record the seed).
"""
import torch
from torch.distributions import Categorical


def rollout(pheromone, heuristic, n_ants, alpha=1, beta=1, require_prob=False):
    """tsp/aco.py:134-177 gen_path + pick_move."""
    n = pheromone.shape[0]
    ants = torch.arange(n_ants)
    cur = torch.randint(low=0, high=n, size=(n_ants,))
    open_mask = torch.ones(size=(n_ants, n))
    open_mask[ants, cur] = 0
    tour, logps = [cur], []
    for _ in range(n - 1):
        weights = (pheromone[cur] ** alpha) * (heuristic[cur] ** beta) * open_mask
        law = Categorical(weights)
        nxt = law.sample()
        if require_prob:
            logps.append(law.log_prob(nxt))
            open_mask = open_mask.clone()
        tour.append(nxt)
        cur = nxt
        open_mask[ants, cur] = 0
    paths = torch.stack(tour)
    return (paths, torch.stack(logps)) if require_prob else paths


def tour_lengths(distances, paths):
    """tsp/aco.py:121-132."""
    u = paths.T
    v = torch.roll(u, shifts=1, dims=1)
    return torch.sum(distances[u, v], dim=1)


def deposit(pheromone, paths, costs, decay=0.9, elitist=False):
    """tsp/aco.py:95-114 (AS / elitist), returns the new pheromone matrix."""
    tau = pheromone * decay
    if elitist:
        c, i = costs.min(dim=0)
        t = paths[:, i]
        tau[t, torch.roll(t, shifts=1)] += 1.0 / c
        tau[torch.roll(t, shifts=1), t] += 1.0 / c
    else:
        for a in range(paths.shape[1]):
            t, c = paths[:, a], costs[a]
            tau[t, torch.roll(t, shifts=1)] += 1.0 / c
            tau[torch.roll(t, shifts=1), t] += 1.0 / c
    return tau


@torch.no_grad()
def colony_iterations(distances, heuristic, n_ants, iterations, decay=0.9):
    """tsp/aco.py:75-92 run() for AS; returns (lowest_cost, pheromone)."""
    tau = torch.ones_like(distances)
    lowest = float("inf")
    for _ in range(iterations):
        paths = rollout(tau, heuristic, n_ants)
        costs = tour_lengths(distances, paths)
        lowest = min(lowest, float(costs.min()))
        tau = deposit(tau, paths, costs, decay)
    return lowest, tau
