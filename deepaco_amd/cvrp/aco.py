"""ACO for CVRP with the class surface of the reference's cvrp/aco.py (and the sampler half of
cvrp_nls/aco.py, which is the same code), running on MI355X.

Node 0 is the depot.  Constructor arguments, method names and return layouts follow
cvrp/aco.py:9-205: `gen_path` returns `paths` of shape (L, n_ants) where L is the length of the
longest ant's route sequence (shorter ones are padded with the depot), `gen_path_costs` sums
the open sequence, `update_pheronome` deposits on directed edges and floors at 1e-10.  The
"adaptive elitist" baseline that shares the reference file (cvrp/aco.py:207-383, declared
unrelated to DeepACO there) is out of scope: adaptive=True raises.
Extra keyword-only arguments `sampler`, `seed` as in deepaco_amd/tsp/aco.py.
"""
import os
import sys

import torch

try:
    from deepaco_amd import engine
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd import engine

CAPACITY = 50


class ACO():

    def __init__(self,  # 0: depot
                 distances,  # (n, n)
                 demand,  # (n, )
                 n_ants=20,
                 decay=0.9,
                 alpha=1,
                 beta=1,
                 elitist=False,
                 min_max=False,
                 pheromone=None,
                 heuristic=None,
                 min=None,
                 device='cpu',
                 adaptive=False,
                 capacity=CAPACITY,
                 *,
                 sampler='scan',
                 seed=None,
                 ):
        if adaptive:
            raise NotImplementedError("the adaptive elitist baseline (cvrp/aco.py:207-383) is out of scope")
        # device='cpu' + host tensors (cvrp/test.py): staged to the HIP device, see engine.stage_to_hip
        distances = engine.stage_to_hip(distances)
        demand = engine.stage_to_hip(demand, distances)
        pheromone = engine.stage_to_hip(pheromone, distances)
        heuristic = engine.stage_to_hip(heuristic, distances)
        self.problem_size = len(distances)
        self.distances = distances
        self.capacity = capacity
        self.demand = demand

        self.n_ants = n_ants
        self.decay = decay
        self.alpha = alpha
        self.beta = beta
        self.elitist = elitist
        self.min_max = min_max
        self.adaptive = False

        if min_max:
            if min is not None:
                assert min > 1e-9
            else:
                min = 0.1
            self.min = min
            self.max = None

        if pheromone is None:
            self.pheromone = torch.ones_like(self.distances)
            if min_max:
                self.pheromone = self.pheromone * self.min
        else:
            self.pheromone = pheromone

        self.heuristic = 1 / distances if heuristic is None else heuristic

        self.shortest_path = None
        self.lowest_cost = float('inf')

        self.device = distances.device
        self.sampler = sampler
        self.seed = torch.initial_seed() if seed is None else seed
        self._calls = 0

    # ------------------------------------------------------------------ cvrp/aco.py:66-69
    def sample(self):
        paths, log_probs = self.gen_path(require_prob=True)
        costs = self.gen_path_costs(paths)
        return costs, log_probs

    # ------------------------------------------------------------------ cvrp/aco.py:72-104
    @torch.no_grad()
    def run(self, n_iterations):
        """The reference's loop (cvrp/aco.py:72-104) without its per-iteration host round trips: route costs and
        the deposit's successor table come fused out of the construction kernel and the `if best_cost <
        self.lowest_cost` bookkeeping is daco_track_best.  One sync at the end trims `shortest_path` to its route.
        (If `gen_path` / `gen_path_costs` were replaced on the instance, the plain call sequence runs instead.)"""
        if "gen_path" in self.__dict__ or "gen_path_costs" in self.__dict__ or type(self).gen_path is not ACO.gen_path \
                or type(self).gen_path_costs is not ACO.gen_path_costs:
            return self._run_plain(n_iterations)
        dev = self.distances.device
        dist = self.distances.detach().float().contiguous()
        n = self.problem_size
        lowest = torch.as_tensor(self.lowest_cost, dtype=torch.float32, device=dev).reshape(1).clone()
        shortest = torch.zeros((1, 2 * n + 1), dtype=torch.int64, device=dev)
        if self.shortest_path is not None:
            shortest[0, :self.shortest_path.numel()] = self.shortest_path
        flags_seen = torch.zeros(1, dtype=torch.int32, device=dev)
        # one private copy for the whole loop (see tsp/aco.py run): updated in place, rebound at the end
        tau = self.pheromone.detach().to(torch.float32).clone().contiguous().unsqueeze(0)
        eta = self.heuristic.detach()
        cmin_t = torch.full((1,), float(self.min), device=dev) if self.min_max else None
        # (loop invariants formed once: at CVRP-20 / 100 with 20 ants an iteration is three library calls whose HOST time is the
        # iteration time -- tools/host_overhead_small.py -- so the demands are laid out [1, n] here, not copied per call, and the
        # flag words are the loop's own, OR-ed into by every construction)
        dem = self.demand.detach()
        dem = (dem if dem.dtype == torch.float64 else dem.to(torch.float32)).reshape(1, -1).contiguous()
        for _ in range(n_iterations):
            paths, _, _, lens, _, costs, table = engine.cvrp_sample(
                tau, eta, dem, self.capacity, self.n_ants, self.alpha,
                self.beta, mode=self.sampler, seed=self.seed, it=self._calls, batch=1, dist=dist, want_table=True, flags=flags_seen)
            self._calls += 1
            new_max = engine.track_best_(costs, paths, lowest, shortest,
                                         mmas_scale=self.problem_size if self.min_max else None)
            cmin = cmax = None
            if self.min_max:
                if self.max is None:
                    tau *= new_max[0] / tau.max()
                self.max = new_max[0]
                cmin, cmax = cmin_t, new_max
            engine.pheromone_update_(tau, paths, costs, self.decay, self.elitist, False, cmin, cmax, floor=1e-10,
                                     nbr=table)
        self.pheromone = tau[0]
        fl = int(flags_seen[0])                      # the only host sync of the loop
        if fl & 1:
            raise ValueError("ACO.run: a transition row had no feasible candidate")
        if fl & 2:
            raise RuntimeError("ACO.run: route buffer too short")
        route = shortest[0]
        last = int(torch.nonzero(route).max()) if bool((route != 0).any()) else 0
        self.lowest_cost, self.shortest_path = lowest[0], route[:last + 2].clone()
        return self.lowest_cost

    @torch.no_grad()
    def _run_plain(self, n_iterations):
        for _ in range(n_iterations):
            paths = self.gen_path(require_prob=False)
            costs = self.gen_path_costs(paths)

            best_cost, best_idx = costs.min(dim=0)
            if best_cost < self.lowest_cost:
                self.shortest_path = paths[:, best_idx]
                self.lowest_cost = best_cost
                if self.min_max:
                    max = self.problem_size / self.lowest_cost
                    if self.max is None:
                        self.pheromone *= max / self.pheromone.max()
                    self.max = max

            self.update_pheronome(paths, costs)

        return self.lowest_cost

    # ------------------------------------------------------------------ cvrp/aco.py:106-130
    @torch.no_grad()
    def update_pheronome(self, paths, costs):
        '''
        Args:
            paths: torch tensor with shape (seq_len, n_ants)
            costs: torch tensor with shape (n_ants,)
        '''
        tau = self.pheromone.detach().to(torch.float32).clone().contiguous().unsqueeze(0)
        cmin = cmax = None
        if self.min_max:
            cmin = torch.full((1,), float(self.min), device=tau.device)
            cmax = torch.as_tensor(self.max, dtype=torch.float32, device=tau.device).reshape(1).contiguous()
        engine.pheromone_update_(tau, paths.contiguous().unsqueeze(0), costs.unsqueeze(0), self.decay, self.elitist,
                                 False, cmin, cmax, floor=1e-10)
        self.pheromone = tau[0]

    update_pheromone = update_pheronome

    # ------------------------------------------------------------------ cvrp/aco.py:132-136
    @torch.no_grad()
    def gen_path_costs(self, paths):
        return engine.tour_costs(self.distances, paths.contiguous().unsqueeze(0), closed=False)[0]

    # ------------------------------------------------------------------ cvrp/aco.py:138-205
    def gen_path(self, require_prob=False, *, _noise=None):
        mode = "race_noise" if _noise is not None else self.sampler
        noise = None if _noise is None else _noise.unsqueeze(0)
        it = self._calls
        self._calls += 1
        if require_prob and torch.is_grad_enabled() and self.heuristic.requires_grad:
            from deepaco_amd.autograd import CvrpSampleFn
            p_, lp_, lens, flags = CvrpSampleFn.apply(self.heuristic, self.pheromone.detach(), self.demand,
                                                      float(self.capacity), self.n_ants, self.alpha, self.beta, mode,
                                                      noise, self.seed, it)
            paths, logp, lens = p_.unsqueeze(0), lp_.unsqueeze(0), lens.unsqueeze(0)
        else:
            paths, logp, _, lens, flags = engine.cvrp_sample(
                self.pheromone.detach(), self.heuristic.detach(), self.demand, self.capacity, self.n_ants,
                self.alpha, self.beta, mode=mode, noise=noise, seed=self.seed, it=it, require_prob=require_prob,
                batch=1)
        L = int(lens.max())                      # host sync (the reference syncs every step: check_done)
        fl = int(flags[0])
        if fl & 1:
            raise ValueError("ACO.gen_path: a transition row had no feasible candidate")
        if fl & 2:
            raise RuntimeError("ACO.gen_path: route buffer / noise tensor too short")
        if require_prob:
            return paths[0, :L], logp[0, :L - 1]
        return paths[0, :L]

    # ------------------------------------------------------------------ cvrp/aco.py:167-205 (the step-wise helpers)
    @torch.no_grad()
    def pick_move(self, prev, visit_mask, capacity_mask, require_prob, *, _noise=None):
        """cvrp/aco.py:167-174: one draw per ant from tau[prev]^alpha * eta[prev]^beta * visit_mask * capacity_mask
        (engine.PickService).  gen_path() is one fused launch and does not call this; see tsp.ACO.pick_move."""
        key = (id(self.pheromone), self.pheromone._version, id(self.heuristic), self.heuristic._version)
        if getattr(self, "_pick_key", None) != key:
            self._pick_svc = engine.PickService(self.pheromone.detach(), self.heuristic.detach(), self.n_ants, self.alpha,
                                                self.beta, mode="scan", seed=self.seed, it=self._calls)
            self._pick_key, self._pick_step = key, 0
            self._calls += 1
        self._pick_step += 1
        dev = self.pheromone.device
        mask = visit_mask.to(dev) * capacity_mask.to(dev)
        actions, log_probs, _ = self._pick_svc.pick(prev.to(dev), mask, self._pick_step, require_prob=bool(require_prob), noise=_noise)
        if bool(self._pick_svc.flags.any()):
            raise ValueError("ACO.pick_move: a transition row had no feasible candidate")
        return actions, log_probs

    def update_visit_mask(self, visit_mask, actions):
        """cvrp/aco.py:176-180: the visited customer closes, the depot is open unless the ant stands on it with customers left."""
        visit_mask[torch.arange(self.n_ants, device=visit_mask.device), actions] = 0
        visit_mask[:, 0] = 1
        visit_mask[(actions == 0) * (visit_mask[:, 1:] != 0).any(dim=1), 0] = 0
        return visit_mask

    def update_capacity_mask(self, cur_nodes, used_capacity):
        """cvrp/aco.py:182-202: (used capacity after serving cur_nodes -- reset at the depot --, mask of the nodes whose demand
        still fits: demand > capacity - used closes a node)."""
        used_capacity[cur_nodes == 0] = 0
        used_capacity = used_capacity + self.demand[cur_nodes]
        remaining = (self.capacity - used_capacity).unsqueeze(-1)
        capacity_mask = torch.ones((self.n_ants, self.problem_size), device=used_capacity.device)
        capacity_mask[self.demand.unsqueeze(0) > remaining] = 0
        return used_capacity, capacity_mask

    def check_done(self, visit_mask, actions):
        """cvrp/aco.py:204-205."""
        return (visit_mask[:, 1:] == 0).all() and (actions == 0).all()
