"""ACO for CVRP with the constructor, sampler and local-search surface of the reference's cvrp_nls/aco.py.

The tour construction, costing and pheromone update of cvrp_nls/aco.py:35-272 are the same code
as cvrp/aco.py on float64 instance data with the capacity normalised to 1.0.  They run on the same HIP
kernels here; probabilities, costs and pheromone are float32, the vehicle-load bookkeeping (which customer still
fits -- with demands k / capacity an exact fit is common and float32 decides 5-10 % of the steps differently,
tests/golden/gen_g1_cvrp_nls.py) is float64 as in the reference whenever `demand` is a float64 tensor
(fixtures g1f64_cvrp_nls_*: the reference's float64 routes on recorded noise).  `sample()` returns
`(costs, log_probs, paths)` as in cvrp_nls/aco.py:100-104.

Local search (`swapstar=True`; cvrp_nls/aco.py:106-128, 443-448).  The reference hands every ant's routes to the
vendored HGS-CVRP C++ (one thread-pool task per ant, /tmp files, ctypes): `neural_swapstar` = LocalSearch::run on the
distances, 10 loops on the heuristic-derived matrix, LocalSearch::run on the distances again.  Here `multiple_swap_star`
improves all selected ants in ONE launch of daco_hgs_local_search (csrc/daco_hgs_ls.hip) that reproduces those three calls
ROUTE FOR ROUTE: moves 1-9 under the 20-nearest granular restriction, first improvement in the order libstdc++'s
std::shuffle over std::minstd_rand fixes, penalised loads, float64 -- and no SWAP*, because the reference's ctypes structure
(swapstar.py:62-74: 10 fields of AlgorithmParameters.h's 15) makes HGS read useSwapStar beyond it (tests/golden/
gen_g11_hgs_ls.py asserts that on every solution).  Pinned on the reference's own outputs: fixtures g8 / g11
(tests/test_gpu_09_cvrp_ls.py, tests/test_gpu_13_hgs_ls.py).  `local_search="best_improvement"` selects round 3's
deterministic best-improvement kernel instead (daco_cvrp_local_search: moves 1-9 + SWAP*, hard capacity; cost-pinned only).
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
                 swapstar=False, positions=None, inference=False, *, sampler='scan', seed=None, local_search='hgs'):
        # demand keeps its dtype: float64 demands (cvrp_nls/utils.py:12-26) select the float64 load bookkeeping of the
        # sampler (cvrp_nls/aco.py:254-272 runs it in double; with demands k / capacity the last bit decides exact fits)
        super().__init__(distances.float(), demand, n_ants, decay, alpha, beta, elitist, min_max,
                         None if pheromone is None else pheromone.float(),
                         None if heuristic is None else heuristic.float(), min, device, adaptive, float(capacity),
                         sampler=sampler, seed=seed)
        self.swapstar, self.positions, self.inference = swapstar, positions, inference
        assert positions is not None if swapstar else True                      # cvrp_nls/aco.py:73
        assert local_search in ('hgs', 'best_improvement')
        self.local_search = local_search
        self._heuristic_dist = None
        # the local search works on the caller's own numbers (the reference hands HGS distances_cpu / heuristic_dist in the
        # dtype they came in, float64 from cvrp_nls/utils.py): kept next to the float32 copies the sampler uses
        self._dist_src = engine.stage_to_hip(distances.detach(), like=self.distances)
        self._heu_src = None if heuristic is None else engine.stage_to_hip(heuristic.detach(), like=self.distances)
        self._demand_src = engine.stage_to_hip(demand.detach(), like=self.distances)
        self._hgs = None
        self._host = {}                                  # distances_cpu / demand_cpu / positions_cpu, on first use
        self._heu_at_init = self.heuristic               # (a heuristic assigned later replaces the constructor's in the local search too)

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

    # ------------------------------------------------------------------ cvrp_nls/aco.py:273-287
    # Host copies of the instance, as numpy arrays in the caller's dtype: what the reference hands to its C++ local search
    # (swapstar(self.demand_cpu, self.distances_cpu, ...)).  Nothing here computes with them (the local search reads the device
    # tensors); they exist for scripts that read them off the colony.  Cached on first use, like the reference's cached_property.
    @property
    def distances_cpu(self):
        if "distances_cpu" not in self._host:
            self._host["distances_cpu"] = self._dist_src.detach().cpu().numpy()
        return self._host["distances_cpu"]

    @property
    def demand_cpu(self):
        if "demand_cpu" not in self._host:
            self._host["demand_cpu"] = self._demand_src.detach().cpu().numpy()
        return self._host["demand_cpu"]

    @property
    def positions_cpu(self):
        if "positions_cpu" not in self._host:
            self._host["positions_cpu"] = self.positions.detach().cpu().numpy() if self.positions is not None else None
        return self._host["positions_cpu"]

    def _hgs_stage_tables(self):
        """(tables of the distances, tables of the heuristic-derived matrix): what HGS's Params derives from a matrix, once per
        colony.  The perturbation matrix is cvrp_nls/aco.py:128-132 in the heuristic's own dtype (numpy there, torch here:
        the same IEEE divisions), the default heuristic 1 / distances (cvrp_nls/aco.py:92)."""
        if self._hgs is None:
            if self.heuristic is not self._heu_at_init:      # assigned after construction (the reference reads self.heuristic at first use)
                heu = self.heuristic.detach()
            else:
                heu = self._heu_src if self._heu_src is not None else 1 / self._dist_src
            hd = 1 / (heu / heu.max(-1, keepdim=True).values + 1e-5)
            self._hgs = (engine.HgsTables(self._dist_src), engine.HgsTables(hd))
        return self._hgs

    # ------------------------------------------------------------------ cvrp_nls/aco.py:114-126, 443-448
    @torch.no_grad()
    def multiple_swap_star(self, paths, indexes=None, disturb=10):
        """Improve the ants' solutions (all, or the columns `indexes`) in place and return `paths` ([L, A] int64):
        neural_swapstar (cvrp_nls/aco.py:443-448) on every selected column, limit = 100000 with `inference`, else
        max(problem_size, 50) (cvrp_nls/aco.py:123) -- the loop bound of LocalSearch::run (LocalSearch.cpp:17)."""
        sel = paths if indexes is None else paths[:, indexes]
        work = sel.contiguous().unsqueeze(0).clone()
        if self.local_search == 'hgs':
            limit = 100000 if self.inference else max(self.problem_size, 50)
            td, th = self._hgs_stage_tables()
            engine.hgs_local_search_(work, [(td, limit), (th, disturb), (td, limit)], self._demand_src,
                                     capacity=1000.001 * self.capacity, demand_scale=1000.0)
        else:
            # best improvement to convergence / `disturb` moves on the perturbation matrix / to convergence
            dist = self.distances.detach().float().contiguous()
            for matrix, count in ((dist, 100000), (self.heuristic_dist, disturb), (dist, 100000)):
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
