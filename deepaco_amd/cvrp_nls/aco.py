"""ACO for CVRP with the constructor, sampler and local-search surface of the reference's cvrp_nls/aco.py.

The tour construction, costing and pheromone update of cvrp_nls/aco.py:35-272 are the same code
as cvrp/aco.py on float64 instance data with the capacity normalised to 1.0.  They run on the same HIP
kernels here; probabilities, costs and pheromone are float32, the vehicle-load bookkeeping (which customer still
fits -- with demands k / capacity an exact fit is common and float32 decides 5-10 % of the steps differently,
tests/golden/gen_g1_cvrp_nls.py) is float64 as in the reference whenever `demand` is a float64 tensor
(fixtures g1f64_cvrp_nls_*: the reference's float64 routes on recorded noise).  `sample()` returns
`(costs, log_probs, paths)` as in cvrp_nls/aco.py:100-104.

Local search (`swapstar=True`; cvrp_nls/aco.py:106-128, 443-448).  The reference hands every ant's routes to the
vendored HGS-CVRP C++ (one thread-pool task per ant, /tmp files, ctypes).  Here `multiple_swap_star` improves all
selected ants in ONE launch per stage of daco_cvrp_local_search (csrc/daco_cvrp_ls.hip: best improvement over HGS's move
families 1-9 -- relocate 1 / 2 / 2 reversed, swap 1-1 / 2-1 / 2-2, 2-opt, 2-opt* both ways -- and SWAP*, with hard capacity) and
keeps the reference's three-stage schedule `neural_swapstar`: search on the distances, `disturb` = 10 moves on the
heuristic-derived matrix, search on the distances again.  HGS's own LocalSearch (first improvement in a shuffled order,
load penalties) is not reproduced move for move; parity is pinned on cost: on solutions sampled by the reference
the schedule reaches 0.986-0.999 of the mean cost of the reference's own neural_swapstar (gate: not more than 0.5 % above; tests/golden/g8_*,
tests/test_gpu_09_cvrp_ls.py), every result feasible, never worse than its input, a local optimum of the move set.
"""
import os
import sys

import torch

try:
    from deepaco_amd.cvrp.aco import ACO as _CvrpACO
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd.cvrp.aco import ACO as _CvrpACO
from deepaco_amd import engine

CAPACITY = 1.0


def get_subroutes(route, end_with_zero=True):
    """cvrp_nls/aco.py:12-20: the non-empty depot-to-depot pieces of a route sequence."""
    x = torch.nonzero(route == 0).flatten()
    subroutes = []
    for i, j in zip(x, x[1:]):
        if j - i > 1:
            subroutes.append(route[i:j + 1] if end_with_zero else route[i:j])
    return subroutes


def merge_subroutes(subroutes, length, device):
    """cvrp_nls/aco.py:22-33: back to one zero-padded sequence of `length` entries."""
    route = torch.zeros(length, dtype=torch.long, device=device)
    i = 0
    for r in subroutes:
        if len(r) > 2:
            r = torch.as_tensor(r[:-1], device=device)
            route[i: i + len(r)] = r
            i += len(r)
    return route


class ACO(_CvrpACO):

    def __init__(self, distances, demand, n_ants=20, decay=0.9, alpha=1, beta=1, elitist=False, min_max=False,
                 pheromone=None, heuristic=None, min=None, device='cpu', adaptive=False, capacity=CAPACITY,
                 swapstar=False, positions=None, inference=False, *, sampler='scan', seed=None):
        # demand keeps its dtype: float64 demands (cvrp_nls/utils.py:12-26) select the float64 load bookkeeping of the
        # sampler (cvrp_nls/aco.py:254-272 runs it in double; with demands k / capacity the last bit decides exact fits)
        super().__init__(distances.float(), demand, n_ants, decay, alpha, beta, elitist, min_max,
                         None if pheromone is None else pheromone.float(),
                         None if heuristic is None else heuristic.float(), min, device, adaptive, float(capacity),
                         sampler=sampler, seed=seed)
        self.swapstar, self.positions, self.inference = swapstar, positions, inference
        self._heuristic_dist = None

    def sample(self, inference=False):
        paths, log_probs = self.gen_path(require_prob=True)
        costs = self.gen_path_costs(paths)
        return costs, log_probs, paths

    # ------------------------------------------------------------------ cvrp_nls/aco.py:106-112
    def sample_nls(self):
        paths, log_probs = self.gen_path(require_prob=True)
        costs_raw = self.gen_path_costs(paths).detach()
        paths = self.multiple_swap_star(paths.clone())     # (the sampled routes stay as drawn: the backward pass replays them)
        costs = self.gen_path_costs(paths).detach()
        return costs, log_probs, costs_raw

    # ------------------------------------------------------------------ cvrp_nls/aco.py:128-132
    @property
    def heuristic_dist(self):
        if self._heuristic_dist is None:
            heu = self.heuristic.detach().float()
            self._heuristic_dist = (1 / (heu / heu.max(-1, keepdim=True).values + 1e-5)).contiguous()
        return self._heuristic_dist

    # ------------------------------------------------------------------ cvrp_nls/aco.py:114-126, 443-448
    @torch.no_grad()
    def multiple_swap_star(self, paths, indexes=None, disturb=10):
        """Improve the ants' solutions (all, or the columns `indexes`) in place and return `paths` ([L, A] int64).
        The reference's `count` (cvrp_nls/aco.py:443-448: limit / disturb / limit) bounds LOOPS of HGS's LocalSearch::run
        (LocalSearch.cpp:17: up to count + 1 passes over all nodes, each applying many moves), not moves: the searches on
        the distances run until no move improves, the perturbation on the heuristic-derived matrix applies `disturb` moves."""
        limit = 100000
        sel = paths if indexes is None else paths[:, indexes]
        work = sel.contiguous().unsqueeze(0).clone()
        dist = self.distances.detach().float().contiguous()
        for matrix, count in ((dist, limit), (self.heuristic_dist, disturb), (dist, limit)):
            engine.cvrp_local_search_(matrix, self.demand, self.capacity, work, count)
        if indexes is None:
            paths.copy_(work[0])
        else:
            paths[:, indexes] = work[0]
        return paths

    @torch.no_grad()
    def run(self, n_iterations, inference=False):
        if not self.swapstar:
            return super().run(n_iterations)
        for _ in range(n_iterations):                          # cvrp_nls/aco.py:135-171 (non-adaptive branch)
            paths = self.gen_path(require_prob=False)
            costs = self.gen_path_costs(paths)
            indexes = costs.topk(min(8, self.n_ants), largest=False).indices
            self.multiple_swap_star(paths, indexes=indexes)
            costs = self.gen_path_costs(paths)
            best_cost, best_idx = costs.min(dim=0)
            if best_cost < self.lowest_cost:
                self.shortest_path = paths[:, best_idx].clone()
                self.lowest_cost = best_cost
                if self.min_max:
                    max_ = self.problem_size / self.lowest_cost
                    if self.max is None:
                        self.pheromone *= max_ / self.pheromone.max()
                    self.max = max_
            self.update_pheronome(paths, costs)
        return self.lowest_cost
