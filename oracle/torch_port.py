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


# ------------------------------------------------------------------ CVRP (cvrp/aco.py:138-205, :107-136)
def cvrp_rollout(pheromone, heuristic, demand, capacity, n_ants, alpha=1, beta=1):
    """cvrp/aco.py:138-205 gen_path: depot start, visit mask with the depot rule (:176-180), capacity mask rebuilt per
    step (:182-202), the all-done test with its host sync (:204-205).  Returns paths [L, n_ants]."""
    n = pheromone.shape[0]
    ants = torch.arange(n_ants)

    def visit_update(mask, cur):
        mask[ants, cur] = 0
        mask[:, 0] = 1
        mask[(cur == 0) * (mask[:, 1:] != 0).any(dim=1), 0] = 0
        return mask

    def capacity_update(cur, used):
        cap_mask = torch.ones(size=(n_ants, n))
        used[cur == 0] = 0
        used = used + demand[cur]
        room = (capacity - used).unsqueeze(-1).repeat(1, n)
        cap_mask[demand.unsqueeze(0).repeat(n_ants, 1) > room] = 0
        return used, cap_mask

    cur = torch.zeros((n_ants,), dtype=torch.long)
    open_mask = visit_update(torch.ones(size=(n_ants, n)), cur)
    used, cap_mask = capacity_update(cur, torch.zeros(size=(n_ants,)))
    route = [cur]
    while not ((open_mask[:, 1:] == 0).all() and (cur == 0).all()):
        weights = (pheromone[cur] ** alpha) * (heuristic[cur] ** beta) * open_mask * cap_mask
        cur = Categorical(weights).sample()
        route.append(cur)
        open_mask = visit_update(open_mask, cur)
        used, cap_mask = capacity_update(cur, used)
    return torch.stack(route)


def route_lengths(distances, paths):
    """cvrp/aco.py:133-136."""
    u = paths.permute(1, 0)
    return torch.sum(distances[u[:, :-1], u[:, 1:]], dim=1)


def deposit_directed(pheromone, paths, costs, decay=0.9):
    """cvrp/aco.py:107-130 (AS): directed, non-accumulating index_put per ant, floor 1e-10."""
    tau = pheromone * decay
    for a in range(paths.shape[1]):
        t, c = paths[:, a], costs[a]
        tau[t[:-1], torch.roll(t, shifts=-1)[:-1]] += 1.0 / c
    tau[tau < 1e-10] = 1e-10
    return tau
