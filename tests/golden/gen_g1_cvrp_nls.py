#!/usr/bin/env python3
"""g1f64_cvrp_nls: the reference's cvrp_nls sampler on its own float64 instance data, with the Exp(1) noise recorded
(this container only).  cvrp_nls/aco.py:205-272 as it is run by cvrp_nls/train.py: demands, distances float64
(cvrp_nls/utils.py:12-32), heuristic float32 (the network's output + 1e-5), so the load bookkeeping
(used = used + demand[cur]; demand > capacity - used) and the probabilities are float64.  torch.multinomial is replaced
by the arithmetic of its one-sample path (argmax(probs / q), q ~ Exp(1) recorded), as in gen_golden.py, and the tapped
run is checked against the untapped one.

Run:  make -C oracle ref && python tests/golden/gen_g1_cvrp_nls.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(os.environ.get("DEEPACO_REFERENCE", "/root/reference"), "cvrp_nls")
LIB = os.path.join(ROOT, "oracle", "_ref", "libhgscvrp.so")
scratch = tempfile.mkdtemp(prefix="g1n_")
os.makedirs(os.path.join(scratch, "HGS-CVRP-main", "build"))
os.symlink(LIB, os.path.join(scratch, "HGS-CVRP-main", "build", "libhgscvrp.so"))     # (aco.py imports swapstar.py)
os.chdir(scratch)
sys.path.insert(0, os.path.join(HERE, "shims"))
sys.path.insert(0, REF)
import aco as ref_aco  # noqa: E402
import utils as ref_utils  # noqa: E402


class NoiseTap:
    def __init__(self):
        self.q, self.orig = [], torch.multinomial

    def __call__(self, probs, num_samples, replacement=False, *, generator=None):
        assert num_samples == 1
        q = torch.empty_like(probs).exponential_(1)
        self.q.append(q.clone())
        return torch.argmax(probs / q, dim=-1, keepdim=True)

    def __enter__(self):
        torch.multinomial = self
        return self

    def __exit__(self, *a):
        torch.multinomial = self.orig


def main():
    for n, A, seed in ((20, 8, 301), (50, 8, 302), (100, 6, 303)):
        torch.manual_seed(seed)
        demands, distances = ref_utils.gen_instance(n, "cpu")                  # float64
        g = torch.Generator().manual_seed(seed + 1)
        heu = (torch.rand(n + 1, n + 1, generator=g) + 1e-5).float()            # float32, like Net's output + EPS
        phe = (torch.rand(n + 1, n + 1, generator=g, dtype=torch.double) + 0.1)
        torch.manual_seed(seed)
        ref = ref_aco.ACO(distances, demands, n_ants=A, heuristic=heu, pheromone=phe.clone())
        ref_paths, ref_logp = ref.gen_path(True)
        torch.manual_seed(seed)
        aco = ref_aco.ACO(distances, demands, n_ants=A, heuristic=heu, pheromone=phe.clone())
        with NoiseTap() as tap:
            paths, logp = aco.gen_path(True)
        assert torch.equal(paths, ref_paths) and torch.equal(logp, ref_logp)
        assert logp.dtype == torch.float64 and tap.q[0].dtype == torch.float64
        costs = aco.gen_path_costs(paths)
        # how many of these draws does float32 bookkeeping decide differently?  (the float32 image of the same instance)
        exact_fit = 0
        used = torch.zeros(A, dtype=torch.double)
        for t in range(paths.shape[0] - 1):
            cur = paths[t]
            used[cur == 0] = 0
            used = used + demands[cur]
            rem64 = 1.0 - used
            rem32 = (torch.tensor(1.0) - used.float())
            m64 = demands.unsqueeze(0) > rem64.unsqueeze(1)
            m32 = demands.float().unsqueeze(0) > rem32.unsqueeze(1)
            exact_fit += int((m64 != m32).any(dim=1).sum())
        name = os.path.join(HERE, f"g1f64_cvrp_nls_n{n}_a{A}.npz")
        np.savez_compressed(name, distances=distances.numpy(), demand=demands.numpy(), capacity=np.float64(ref_aco.CAPACITY),
                            heuristic=heu.numpy(), pheromone=phe.numpy(), noise=torch.stack(tap.q).numpy(),
                            paths=paths.numpy(), log_probs=logp.numpy(), costs=costs.numpy())
        print(f"{name}: L = {paths.shape[0]}, ant-steps whose capacity mask differs between float64 and float32 bookkeeping: "
              f"{exact_fit} of {A * (paths.shape[0] - 1)}")


if __name__ == "__main__":
    main()
