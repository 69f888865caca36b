"""ACO for TSP with the class surface of the reference's tsp/aco.py, running on MI355X.

Import it either as `from deepaco_amd.tsp.aco import ACO` or, like the reference's scripts
do (tsp/train.ipynb:14-16 `from aco import ACO`), by putting this directory on sys.path.

Constructor arguments, method names (including the reference's spelling `update_pheronome`),
return layouts and attributes follow tsp/aco.py:4-177.  What differs is underneath: tour
construction, costing and the pheromone update are single launches of hand-written HIP
kernels (libdeepaco_hip.so) instead of ~20 aten ops per step.  Extra keyword-only arguments:

  sampler  'auto' (default): the scan draw (roulette by wavefront prefix scan, one Philox uniform per step) -- on head / tail
           rows ('scan_sparse': the same categorical from 384 / 768 bytes per step instead of a whole row) wherever they apply:
           after sparsify(k), or when 63 / 127 entries hold >= 98 % of every heuristic row (the learned heuristic is k-sparse),
           and 129 <= n <= 1024; 'scan' / 'scan_sparse' force one of them
           'race': the exponential race torch.multinomial runs, noise from Philox in-kernel
  seed     Philox key (default: torch.initial_seed(), so torch.manual_seed() governs the run)

Both samplers draw from exactly the reference's categorical distribution
p_k ~ tau^alpha * eta^beta * mask.  For bit-exact comparison with the reference pass the
reference's own noise: gen_path(..., _start=start, _noise=q) with q of shape [n-1, A, n].
"""
import os
import sys

import torch

try:
    from deepaco_amd import engine
except ImportError:  # imported as a bare module from this directory, like the reference's layout
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd import engine


class ACO():

    NORM_PASSES = 1      # Categorical(dist) normalises once (tsp/aco.py:174)
    FIXED_START = -1     # random start node per ant (tsp/aco.py:141)

    def __init__(self,
                 distances,
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
                 *,
                 sampler='auto',
                 seed=None,
                 ):
        # device='cpu' + host tensors (the reference's test scripts): staged to the HIP device, see engine.stage_to_hip
        distances = engine.stage_to_hip(distances)
        pheromone = engine.stage_to_hip(pheromone, distances)
        heuristic = engine.stage_to_hip(heuristic, distances)
        self.problem_size = len(distances)
        self.distances = distances
        self.n_ants = n_ants
        self.decay = decay
        self.alpha = alpha
        self.beta = beta
        self.elitist = elitist
        self.min_max = min_max

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

    # ------------------------------------------------------------------ tsp/aco.py:52-67
    @torch.no_grad()
    def sparsify(self, k_sparse):
        '''Heuristic = 1/dist on each node's k nearest neighbours, 1e-10 elsewhere
        (vanilla-ACO baseline).'''
        _, topk_indices = torch.topk(self.distances, k=k_sparse, dim=1, largest=False)
        sparse_distances = torch.full_like(self.distances, 1e10)
        sparse_distances.scatter_(1, topk_indices, torch.gather(self.distances, 1, topk_indices))
        self.heuristic = 1 / sparse_distances
        self._head_k = min(int(k_sparse), 127)              # sampler='scan_sparse': the head of a row = these k entries

    def resolved_sampler(self):
        """(sampler the next construction runs, head size): engine.resolve_sampler."""
        cache = self.__dict__.setdefault("_auto", {})
        return engine.resolve_sampler(self.sampler, self.problem_size, self.__dict__.get("_head_k"), self.heuristic, cache)

    def _head_table(self, k=None):
        """[1, n, 64 | 128] head ids for sampler='scan_sparse' (engine.sparse_head), once per heuristic object."""
        hit = self.__dict__.get("_head")
        if hit is None or hit[0] is not self.heuristic or (k is not None and hit[2] != k):
            k = k or self.__dict__.get("_head_k") or max(1, min(127, self.problem_size // 10))
            top = engine.take_auto_top(self.__dict__.get("_auto"), self.heuristic)
            hit = (self.heuristic, engine.sparse_head(self.heuristic.detach().float().contiguous(), k, top=top), k)
            self._head = hit
        return hit[1]

    # ------------------------------------------------------------------ tsp/aco.py:69-72
    def sample(self):
        paths, log_probs = self.gen_path(require_prob=True)
        costs = self.gen_path_costs(paths)
        return costs, log_probs

    # ------------------------------------------------------------------ tsp/aco.py:75-92
    @torch.no_grad()
    def run(self, n_iterations):
        """The reference's loop (tsp/aco.py:75-92) without its per-iteration host sync: the
        `if best_cost < self.lowest_cost` bookkeeping is done with device-side selects, and tour costs and
        the neighbour table of the deposit come fused out of the construction kernel.  `lowest_cost` /
        `shortest_path` are tensors from the first iteration on.  (If `gen_path` has been replaced on the
        instance -- noise injection in the parity tests -- the plain call sequence is used instead.)"""
        if "gen_path" in self.__dict__ or type(self).gen_path_costs is not ACO.gen_path_costs \
                or getattr(self, "local_search_type", None) is not None:
            return self._run_plain(n_iterations)
        sampler, hk = self.resolved_sampler()               # head / tail rows where they apply (a k-sparse heuristic, 129 <= n <= 1024)
        if sampler == "scan_sparse":
            return self._run_colony(n_iterations, hk)
        dev = self.distances.device
        dist = self.distances.detach().float().contiguous()
        lowest = torch.as_tensor(self.lowest_cost, dtype=torch.float32, device=dev).reshape(1).clone()
        shortest = (self.shortest_path.clone() if self.shortest_path is not None else
                    torch.zeros(self.problem_size, dtype=torch.int64, device=dev)).reshape(1, -1).contiguous()
        # one private copy for the whole loop, updated in place and rebound at the end (the reference rebinds
        # self.pheromone every iteration, tsp/aco.py:101: a tensor the caller still holds is never modified)
        tau = self.pheromone.detach().to(torch.float32).clone().contiguous().unsqueeze(0)
        eta = self.heuristic.detach()
        cmin_t = torch.full((1,), float(self.min), device=dev) if self.min_max else None
        flags = torch.zeros(1, dtype=torch.int32, device=dev)        # the loop's own flag words (OR-ed into by every construction)
        for _ in range(n_iterations):
            paths, _, _, _, costs, nbr = engine.tsp_sample(
                tau, eta, self.n_ants, self.alpha, self.beta, mode=sampler,
                norm_passes=self.NORM_PASSES, fixed_start=self.FIXED_START, seed=self.seed, it=self._calls, batch=1,
                dist=dist, want_nbr=True, flags=flags)
            self._calls += 1
            self._last_flags = flags
            # `if best_cost < self.lowest_cost: ...` (tsp/aco.py:78-88) on the device
            new_max = engine.track_best_(costs, paths, lowest, shortest,
                                         mmas_scale=self.problem_size if self.min_max else None)
            cmin = cmax = None
            if self.min_max:
                if self.max is None:
                    tau *= new_max[0] / tau.max()
                self.max = new_max[0]
                cmin, cmax = cmin_t, new_max
            engine.pheromone_update_(tau, paths, costs, self.decay, self.elitist, True, cmin, cmax, nbr=nbr)
        self.pheromone = tau[0]
        self.lowest_cost, self.shortest_path = lowest[0], shortest[0]
        return self.lowest_cost

    @torch.no_grad()
    def _run_colony(self, n_iterations, hk):
        """run() on head / tail rows = a one-instance engine.BatchedTSP kept with this object (round 6): the colony owns the
        sampler's workspace, so the pheromone update leaves the next iteration's head rows there (no pre-pass launch, tau read once
        per iteration), a launch of few ants keeps the instance's head rows in LDS (this IS the reference's call
        pattern -- one instance, 20-50 ants, tsp/test.ipynb:31-70) and the tours stay compact.  Same seed and iteration counter as
        the loop above: the same tours, costs, records and pheromone (tests/test_gpu_00_tsp.py)."""
        dev = self.distances.device
        key = (self.distances, self.heuristic, hk, self.n_ants, self.decay, self.alpha, self.beta, bool(self.elitist), bool(self.min_max), self.seed)
        hit = self.__dict__.get("_colony")
        if hit is None or any((a is not b) if torch.is_tensor(a) else (a != b) for a, b in zip(hit[0], key)):
            eta = self.heuristic.detach().to(torch.float32).contiguous().unsqueeze(0)
            col = engine.BatchedTSP(self.distances.detach().float().contiguous().unsqueeze(0), n_ants=self.n_ants, decay=self.decay,
                                    alpha=self.alpha, beta=self.beta, elitist=self.elitist, min_max=self.min_max, heuristic=eta,
                                    min=self.min if self.min_max else None, sampler="scan_sparse", seed=self.seed,
                                    fixed_start=self.FIXED_START)
            col.head_k = hk
            col._head = (eta, self._head_table(hk), hk)         # (this object's table: built once, from the sorted top values if 'auto' left them)
            hit = self._colony = (key, col)
        col = hit[1]
        # one private copy for the whole loop, rebound at the end (the reference rebinds self.pheromone every iteration,
        # tsp/aco.py:101: a tensor the caller still holds is never modified)
        col.pheromone = self.pheromone.detach().to(torch.float32).clone().contiguous().unsqueeze(0)
        col.lowest_cost = torch.as_tensor(self.lowest_cost, dtype=torch.float32, device=dev).reshape(1).clone()
        col.shortest_path = (self.shortest_path.clone() if self.shortest_path is not None else
                             torch.zeros(self.problem_size, dtype=torch.int64, device=dev)).reshape(1, -1).contiguous()
        col.iteration = self._calls
        if self.min_max:
            col.min = self.min
            col._cmin = None
            col.max = None if self.max is None else torch.as_tensor(self.max, dtype=torch.float32, device=dev).reshape(1).clone()
        col._flags.zero_()
        for _ in range(n_iterations):
            col.step(want_paths=False)
        self._calls = col.iteration
        self._last_flags = col._flags
        if self.min_max and col.max is not None:
            self.max = col.max[0]
        self.pheromone = col.pheromone[0]
        self.lowest_cost, self.shortest_path = col.lowest_cost[0], col.shortest_path[0]
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

    # ------------------------------------------------------------------ tsp/aco.py:95-118
    @torch.no_grad()
    def update_pheronome(self, paths, costs):
        '''
        Args:
            paths: torch tensor with shape (problem_size, n_ants)
            costs: torch tensor with shape (n_ants,)
        '''
        tau = self.pheromone.detach().to(torch.float32).clone().contiguous().unsqueeze(0)
        cmin = cmax = None
        if self.min_max:
            cmin = torch.full((1,), float(self.min), device=tau.device)
            cmax = torch.as_tensor(self.max, dtype=torch.float32, device=tau.device).reshape(1).contiguous()
        engine.pheromone_update_(tau, paths.unsqueeze(0), costs.unsqueeze(0), self.decay, self.elitist, True,
                                 cmin, cmax)
        self.pheromone = tau[0]          # the reference rebinds too (tsp/aco.py:101)

    update_pheromone = update_pheronome  # correctly spelled alias

    # ------------------------------------------------------------------ tsp/aco.py:121-132
    @torch.no_grad()
    def gen_path_costs(self, paths):
        '''
        Args:
            paths: torch tensor with shape (problem_size, n_ants)
        Returns:
            Lengths of paths: torch tensor with shape (n_ants,)
        '''
        assert paths.shape == (self.problem_size, self.n_ants)
        return engine.tour_costs(self.distances, paths.unsqueeze(0))[0]

    # ------------------------------------------------------------------ tsp/aco.py:134-177
    def gen_path(self, require_prob=False, *, _start=None, _noise=None):
        '''
        Tour construction for all ants
        Returns:
            paths: torch tensor with shape (problem_size, n_ants), paths[:, i] is the tour of ant i
            log_probs: torch tensor with shape (problem_size-1, n_ants) (only if require_prob)
        '''
        if _noise is not None:
            mode, start = "race_noise", _start
        else:
            mode, start = self.sampler, _start
        hk = None
        if mode in ("auto", "scan_sparse"):
            # head / tail rows draw tours only: with log-probabilities (training) or prescribed start nodes the dense scan --
            # the same categorical -- builds them
            mode, hk = self.resolved_sampler()
            if mode == "scan_sparse" and (require_prob or start is not None):
                mode = "scan"
        fixed = self.FIXED_START
        if _noise is not None and start is None and fixed < 0:
            raise ValueError("noise injection needs the start nodes too (_start)")
        start = None if start is None else start.view(1, -1)
        noise = None if _noise is None else _noise.unsqueeze(0)
        it = self._calls
        self._calls += 1
        if require_prob and torch.is_grad_enabled() and self.heuristic.requires_grad:
            # log_probs carry gradient to the heuristic (the reference: autograd through Categorical)
            from deepaco_amd.autograd import TspSampleFn
            paths, logp, flags = TspSampleFn.apply(self.heuristic, self.pheromone.detach(), self.n_ants, self.alpha,
                                                   self.beta, mode, self.NORM_PASSES, start, fixed, noise,
                                                   self.seed, it)
            self._last_flags = flags
            return paths, logp
        if mode == "scan_sparse":
            paths, flags, _, _ = engine.tsp_sample_sparse(self.pheromone.detach(), self.heuristic.detach(), self.n_ants,
                                                          self._head_table(hk), self.alpha, self.beta, fixed_start=fixed,
                                                          seed=self.seed, it=it, batch=1)
            self._last_flags = flags
            return paths[0]
        paths, logp, rowsum, flags = engine.tsp_sample(
            self.pheromone.detach(), self.heuristic.detach(), self.n_ants, self.alpha, self.beta, mode=mode,
            norm_passes=self.NORM_PASSES, start=start, fixed_start=fixed, noise=noise,
            seed=self.seed, it=it, require_prob=require_prob, batch=1)
        self._last_flags = flags
        if require_prob:
            return paths[0], logp[0]
        return paths[0]

    # ------------------------------------------------------------------ tsp/aco.py:165-177
    @torch.no_grad()
    def pick_move(self, prev, mask, require_prob, *, _noise=None):
        """One draw per ant from tau[prev]^alpha * eta[prev]^beta * mask (daco_prob_matrix + daco_pick_move through
        engine.PickService): `prev` [n_ants] int64, `mask` [n_ants, problem_size] (0 = closed).  Returns (actions,
        log_probs | None).  gen_path() does NOT call this (its n - 1 draws are one fused kernel launch), so overriding it in
        a subclass does not change gen_path; log-probabilities that carry gradient come from gen_path(require_prob=True) /
        sample().  `_noise` [n_ants, problem_size]: the reference's Exp(1) variates for a bit-exact replay of its draw."""
        key = (id(self.pheromone), self.pheromone._version, id(self.heuristic), self.heuristic._version)
        if getattr(self, "_pick_key", None) != key:
            self._pick_svc = engine.PickService(self.pheromone.detach(), self.heuristic.detach(), self.n_ants, self.alpha,
                                                self.beta, mode="race" if self.sampler == "race" else "scan", seed=self.seed,
                                                it=self._calls)
            self._pick_key, self._pick_step = key, 0
            self._calls += 1
        self._pick_step += 1
        actions, log_probs, _ = self._pick_svc.pick(prev.to(self.pheromone.device), mask.to(self.pheromone.device),
                                                    self._pick_step, require_prob=bool(require_prob), noise=_noise)
        self._last_flags = self._pick_svc.flags
        return actions, log_probs

    def check_feasible(self):
        """Raise like torch.distributions.Categorical does when some draw had no candidate
        (host sync; the kernels only set a flag)."""
        if getattr(self, "_last_flags", None) is not None and bool(self._last_flags.any()):
            raise ValueError("ACO.gen_path: a transition row had no feasible candidate "
                             "(all probabilities zero)")
