"""ACO for TSP + neural-guided local search with the class surface of the reference's
tsp_nls/aco.py, running on MI355X.

Differences from deepaco_amd/tsp/aco.py follow the reference: every ant starts at node 0
(tsp_nls/aco.py:191), the transition row is renormalised explicitly before Categorical
normalises it again (:206-207 -> norm_passes = 2 in parity mode), `sample()` returns
`(costs, log_probs, paths)` (:80-90), `run()` applies the local search before costing (:114)
and keeps `lowest_cost` as a Python float (:120).  The local search (2-opt and the NLS driver,
:234-258) runs on the device: tours never make the reference's `.cpu().numpy()` round trip
(:236-237).  `inference=True` selects, like the reference, the roulette sampler (:260-297) --
here the wavefront prefix-scan kernel -- and `maxt = 10000` sweeps.
"""
import os
import sys

import torch

try:
    from deepaco_amd import engine
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd import engine
from deepaco_amd.tsp.aco import ACO as _TspACO
from deepaco_amd.tsp_nls.two_opt import two_opt_device


class ACO(_TspACO):

    NORM_PASSES = 2
    FIXED_START = 0

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
                 two_opt=False,  # for compatibility
                 device='cpu',
                 local_search='nls',
                 *,
                 sampler='auto',
                 seed=None,
                 ):
        if not distances.is_cuda and str(device) != 'cpu':
            distances = distances.to(device)            # the reference moves them too (:29)
        super().__init__(distances, n_ants, decay, alpha, beta, elitist, min_max, pheromone, heuristic, min,
                         device, sampler=sampler, seed=seed)
        assert local_search in [None, "2opt", "nls"]
        self.local_search_type = '2opt' if two_opt else local_search
        self._heuristic_dist = None

    # ------------------------------------------------------------------ tsp_nls/aco.py:80-95
    def sample(self, inference=False):
        if inference:
            paths = self.gen_path(require_prob=False, _sampler='auto')
            costs = self.gen_path_costs(paths)
            return costs, None, paths
        paths, log_probs = self.gen_path(require_prob=True)
        costs = self.gen_path_costs(paths)
        return costs, log_probs, paths

    def sample_2opt(self, paths):
        paths = self.local_search(paths)
        costs = self.gen_path_costs(paths)
        return costs, paths

    def local_search(self, paths, inference=False):
        if self.local_search_type == "2opt":
            paths = self.two_opt(paths, inference)
        elif self.local_search_type == "nls":
            paths = self.nls(paths, inference)
        return paths

    # ------------------------------------------------------------------ tsp_nls/aco.py:104-129
    @torch.no_grad()
    def run(self, n_iterations, inference=False):
        for _ in range(n_iterations):
            if inference:
                paths = self.gen_path(require_prob=False, _sampler='auto')
            else:
                paths = self.gen_path(require_prob=False)

            paths = self.local_search(paths, inference)
            costs = self.gen_path_costs(paths)

            best_cost, best_idx = costs.min(dim=0)
            if best_cost < self.lowest_cost:
                self.shortest_path = paths[:, best_idx]
                self.lowest_cost = best_cost.item()
                if self.min_max:
                    max = self.problem_size / self.lowest_cost
                    if self.max is None:
                        self.pheromone *= max / self.pheromone.max()
                    self.max = max

            self.update_pheronome(paths, costs)

        return self.lowest_cost

    def gen_path(self, require_prob=False, *, _start=None, _noise=None, _sampler=None):
        if _sampler is not None and _noise is None:
            keep, self.sampler = self.sampler, _sampler
            try:
                return super().gen_path(require_prob, _start=_start, _noise=_noise)
            finally:
                self.sampler = keep
        return super().gen_path(require_prob, _start=_start, _noise=_noise)

    # ------------------------------------------------------------------ tsp_nls/aco.py:222-258
    @property
    def heuristic_dist(self):
        """1 / (eta / rowmax(eta) + 1e-5): the perturbation matrix of the NLS (tsp_nls/aco.py:230-232)."""
        if self._heuristic_dist is None:
            h = self.heuristic.detach().to(torch.float32)
            self._heuristic_dist = (1 / (h / h.max(-1, keepdim=True).values + 1e-5)).contiguous()
        return self._heuristic_dist

    def _transposed(self, name):
        """engine.two_opt_'s dist_t for self.<name>, computed once per matrix object ("symmetric" or a transposed copy)."""
        m = getattr(self, name).detach().to(torch.float32)
        cache = self.__dict__.setdefault("_t_cache", {})
        hit = cache.get(name)
        if hit is None or hit[0] is not getattr(self, name):
            hit = (getattr(self, name), engine.transposed_for_two_opt(m))
            cache[name] = hit
        return hit[1]

    def _tables(self, name):
        """engine.TwoOptTables of self.<name> (neighbour lists for the candidate-list 2-opt kernel), once per matrix object."""
        cache = self.__dict__.setdefault("_tab_cache", {})
        hit = cache.get(name)
        if hit is None or hit[0] is not getattr(self, name):
            m = getattr(self, name).detach().to(torch.float32)
            hit = (getattr(self, name), engine.two_opt_tables(m, self._transposed(name)))
            cache[name] = hit
        return hit[1]

    def _tours(self, paths):
        return paths.T.contiguous().to(torch.int16)

    def _paths(self, tours):
        return tours.T.contiguous().to(torch.int64)

    def _tour_costs(self, tours):
        return engine.tour_costs(self.distances, self._paths(tours).unsqueeze(0))[0]

    @torch.no_grad()
    def two_opt(self, paths, inference=False):
        maxt = 10000 if inference else self.problem_size // 4
        best = two_opt_device(self.distances, self._tours(paths), maxt, self._transposed("distances"),
                              self._tables("distances"))
        return self._paths(best)

    @torch.no_grad()
    def nls(self, paths, inference=False, T_nls=10, T_p=20):
        """tsp_nls/aco.py:241-258.  One launch for the whole search of every tour (engine.nls_: daco_tsp_nls) where the
        candidate tables exist (n <= 1024), the pass-by-pass driver otherwise; the same tours either way."""
        maxt = 10000 if inference else self.problem_size // 4
        dist = self.distances.to(torch.float32)
        tabs, hd_tabs = self._tables("distances"), self._tables("heuristic_dist")
        # (the cached transposes too: without tables -- n > 1024 -- nls_ would otherwise rebuild them, a host sync and an n^2
        # copy per call)
        dt, hdt = self._transposed("distances"), self._transposed("heuristic_dist")
        best = engine.nls_(dist.unsqueeze(0), self.heuristic_dist.unsqueeze(0), self._tours(paths).unsqueeze(0), maxt,
                           T_nls=T_nls, T_p=T_p, dist_t=dt.unsqueeze(0) if torch.is_tensor(dt) else dt,
                           heuristic_dist_t=hdt.unsqueeze(0) if torch.is_tensor(hdt) else hdt, tables=tabs, heuristic_tables=hd_tabs)
        return self._paths(best[0])

    # ------------------------------------------------------------------ tsp_nls/aco.py:171-182, 222-228
    def gen_numpy_path_costs(self, paths, numpy_distances):
        """Closed-tour lengths of `paths` [n_ants, problem_size] (one ROW per ant -- the transposed layout of the local search)
        on a numpy matrix, as tsp_nls/aco.py:171-182 sums them."""
        import numpy as np
        assert paths.shape == (self.n_ants, self.problem_size)
        return np.sum(numpy_distances[paths, np.roll(paths, shift=1, axis=1)], axis=1)

    @property
    def distances_numpy(self):
        """tsp_nls/aco.py:222-224: the distances as a float32 numpy matrix (a host copy; the local search here reads the device one)."""
        if getattr(self, "_distances_numpy", None) is None:
            import numpy as np
            self._distances_numpy = self.distances.detach().cpu().numpy().astype(np.float32)
        return self._distances_numpy

    @property
    def heuristic_numpy(self):
        """tsp_nls/aco.py:226-228."""
        if getattr(self, "_heuristic_numpy", None) is None:
            import numpy as np
            self._heuristic_numpy = self.heuristic.detach().cpu().numpy().astype(np.float32)
        return self._heuristic_numpy


def inference_batch_sample(probmat, count=1, startnode=None, *, seed=None):
    """tsp_nls/aco.py:276-297: `count` tours drawn from the transition matrix `probmat` [n, n] (numpy or tensor) by roulette
    selection, fixed start node (random per tour if None) -> routes [count, n] uint16 numpy.  The reference runs a numba loop
    per tour in a thread pool; here it is one launch of the scan sampler (the same categorical per step: the cumulative sum
    against u * total; fixtures g6 / g6w pin the arithmetic on injected uniforms), its own random stream."""
    import numpy as np
    from deepaco_amd import engine
    p = torch.as_tensor(np.asarray(probmat, dtype=np.float32)) if not torch.is_tensor(probmat) else probmat.detach().float()
    p = engine.stage_to_hip(p).contiguous()
    n = p.shape[0]
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)))
    start = None
    if startnode is None:
        start = torch.randint(0, n, (1, count), device=p.device)
    paths, _, _, flags = engine.tsp_sample(torch.ones_like(p), p, count, mode="scan", start=start,
                                           fixed_start=-1 if startnode is None else int(startnode), seed=seed, batch=1)
    return paths[0].T.contiguous().cpu().numpy().astype(np.uint16)


def _inference_sample(probmat, startnode=0):
    """tsp_nls/aco.py:260-274: one tour."""
    return inference_batch_sample(probmat, 1, startnode)[0]
