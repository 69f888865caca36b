"""ACO for CVRP with the constructor and sampler surface of the reference's cvrp_nls/aco.py.

The tour construction, costing and pheromone update of cvrp_nls/aco.py:35-272 are the same code
as cvrp/aco.py (float64 instance data, capacity normalised to 1.0) and run on the same HIP
kernels here (instance data are cast to float32 on the device).  `sample()` returns
`(costs, log_probs, paths)` as in cvrp_nls/aco.py:100-104.  The SWAP* local search
(`swapstar=True`, cvrp_nls/aco.py:106-128,443-448 -> ctypes into the vendored HGS-CVRP C++ via
/tmp files) is a CPU pointer-chasing solver outside this path's scope:
requesting it raises NotImplementedError instead of silently skipping it.
"""
import os
import sys

import torch

try:
    from deepaco_amd.cvrp.aco import ACO as _CvrpACO
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd.cvrp.aco import ACO as _CvrpACO

CAPACITY = 1.0


class ACO(_CvrpACO):

    def __init__(self, distances, demand, n_ants=20, decay=0.9, alpha=1, beta=1, elitist=False, min_max=False,
                 pheromone=None, heuristic=None, min=None, device='cpu', adaptive=False, capacity=CAPACITY,
                 swapstar=False, positions=None, inference=False, *, sampler='scan', seed=None):
        if swapstar:
            raise NotImplementedError("SWAP* (HGS-CVRP C++ local search) is outside the rollout hot path; "
                                      "use swapstar=False")
        super().__init__(distances.float(), demand.float(), n_ants, decay, alpha, beta, elitist, min_max,
                         None if pheromone is None else pheromone.float(),
                         None if heuristic is None else heuristic.float(), min, device, adaptive, float(capacity),
                         sampler=sampler, seed=seed)
        self.swapstar, self.positions, self.inference = False, positions, inference

    def sample(self, inference=False):
        paths, log_probs = self.gen_path(require_prob=True)
        costs = self.gen_path_costs(paths)
        return costs, log_probs, paths

    @torch.no_grad()
    def run(self, n_iterations, inference=False):
        return super().run(n_iterations)
