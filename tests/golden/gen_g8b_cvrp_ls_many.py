#!/usr/bin/env python3
"""g8b: the cost pin of the CVRP local search on MANY instances (this container only).

Like gen_g8_cvrp_ls.py -- the reference's own Python (cvrp_nls/aco.py, swapstar.py, utils.py imported from
/root/reference) over HGS-CVRP compiled from the reference's sources (`make -C oracle ref`) -- but ten instances per
size (four at n = 200, two at n = 500: the other pretrained size) with eight sampled solutions each, heuristic 1/d, and only what a cost comparison needs
is kept: positions, demands, the sampled route sequences, and the route costs of
  in      the solutions sampled by the reference's ACO.gen_path
  ls10    swapstar(count = 10)          HGS LocalSearch::run, 10 loops
  ls100   swapstar(count = 100)
  nls     neural_swapstar(limit = max(n, 50))   the training-time schedule (cvrp_nls/aco.py:100-126, inference = False)
  nls_inf neural_swapstar(limit = 10000)        the inference schedule
One small .npz for all sizes.  Run:  make -C oracle ref && python tests/golden/gen_g8b_cvrp_ls_many.py
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
assert os.path.isfile(LIB), "build it first: make -C oracle ref"
scratch = tempfile.mkdtemp(prefix="g8b_")
os.makedirs(os.path.join(scratch, "HGS-CVRP-main", "build"))
os.symlink(LIB, os.path.join(scratch, "HGS-CVRP-main", "build", "libhgscvrp.so"))
os.chdir(scratch)
sys.path.insert(0, os.path.join(HERE, "shims"))
sys.path.insert(0, REF)
import swapstar as ref_swapstar  # noqa: E402
ref_swapstar.HGS_LIBRARY_FILEPATH = LIB
import aco as ref_aco  # noqa: E402
import utils as ref_utils  # noqa: E402


def cost64(dist, routes):
    return float(sum(dist[r[:-1], r[1:]].sum() for r in routes))


def main():
    out = {}
    A = 8
    for n, instances in ((20, 10), (50, 10), (100, 10), (200, 4), (500, 2)):
        L = 2 * n + 3
        pos_all, dem_all, paths_all = [], [], []
        costs = {k: [] for k in ("in", "ls10", "ls100", "nls", "nls_inf")}
        for inst in range(instances):
            torch.manual_seed(7000 + 13 * n + inst)
            np.random.seed(n + inst)
            demands, distances, positions = ref_utils.gen_instance(n, "cpu", True)
            heu = 1.0 / distances
            colony = ref_aco.ACO(distances, demands, n_ants=A, heuristic=heu, swapstar=True, positions=positions)
            paths = colony.gen_path(require_prob=False)
            d_np, dem_np, pos_np, hd_np = colony.distances_cpu, colony.demand_cpu, colony.positions_cpu, colony.heuristic_dist
            col = np.zeros((L, A), dtype=np.int16)
            col[:paths.shape[0]] = paths.numpy()
            for a in range(A):
                p0 = ref_aco.get_subroutes(paths[:, a])
                outs = {"in": p0,
                        "ls10": ref_swapstar.swapstar(dem_np, d_np, pos_np, p0, count=10),
                        "ls100": ref_swapstar.swapstar(dem_np, d_np, pos_np, p0, count=100),
                        "nls": ref_aco.neural_swapstar(dem_np, d_np, hd_np, pos_np, p0, limit=max(n, 50)),
                        "nls_inf": ref_aco.neural_swapstar(dem_np, d_np, hd_np, pos_np, p0, limit=10000)}
                for k, routes in outs.items():
                    costs[k].append(cost64(d_np, [np.asarray(r) for r in routes]))
            pos_all.append(pos_np); dem_all.append(dem_np); paths_all.append(col)
        out[f"n{n}_positions"] = np.stack(pos_all)                     # [I, n+1, 2] float64
        out[f"n{n}_demands"] = np.stack(dem_all)                       # [I, n+1] float64 (normalised: capacity 1.0)
        out[f"n{n}_paths_in"] = np.stack(paths_all)                    # [I, L, A] int16, zero-padded route sequences
        for k, v in costs.items():
            out[f"n{n}_costs_{k}"] = np.asarray(v, dtype=np.float64).reshape(instances, A)
        print(f"n = {n}: " + " | ".join(f"{k} {np.mean(v):.4f}" for k, v in costs.items()))
    np.savez_compressed(os.path.join(HERE, "g8b_cvrp_ls_many.npz"), **out)


if __name__ == "__main__":
    main()
