#!/usr/bin/env python3
"""g8: routes-in / routes-out of the REFERENCE's CVRP local search (this container only).

cvrp_nls/aco.py:114-126,443-448 -> cvrp_nls/swapstar.py:240-271 -> HGS-CVRP-main/Program/C_Interface.cpp:128-172
(LocalSearch::run with AlgorithmParameters.seed = 0: deterministic).  The C++ is compiled by `make -C oracle ref`
(plain g++ on the reference's own files, output oracle/_ref/libhgscvrp.so); the reference's Python (swapstar.py, aco.py,
utils.py) is IMPORTED from /root/reference and run as it is: instances from cvrp_nls/utils.gen_instance, solutions
sampled by the reference's ACO.gen_path, improved by the reference's swapstar() / neural_swapstar().  What is committed
is data: instance, sampled routes, improved routes, route costs.

swapstar.py resolves the library through a path relative to the current directory and then loads it from its own
directory, HGS-CVRP-main/build/libhgscvrp.so -- the reference ships that binary (an earlier version of this comment said it
does not); the fixtures are nevertheless generated through oracle/_ref, the build of the reference's own sources that
`make -C oracle ref` reproduces, and tests/test_hgs_ls_oracle.py::test_oracle_against_the_reference_library_live holds the
restatement against BOTH libraries where the reference checkout is present.  This script runs from a scratch directory
holding the relative path and points the module's HGS_LIBRARY_FILEPATH at oracle/_ref after import.

Run:  make -C oracle ref && python tests/golden/gen_g8_cvrp_ls.py   (writes tests/golden/g8_cvrp_ls_n*.npz)
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

scratch = tempfile.mkdtemp(prefix="g8_")
os.makedirs(os.path.join(scratch, "HGS-CVRP-main", "build"))
os.symlink(LIB, os.path.join(scratch, "HGS-CVRP-main", "build", "libhgscvrp.so"))
os.chdir(scratch)
sys.path.insert(0, os.path.join(HERE, "shims"))          # torch_geometric stand-in (utils.py imports Data)
sys.path.insert(0, REF)
import swapstar as ref_swapstar  # noqa: E402
ref_swapstar.HGS_LIBRARY_FILEPATH = LIB
import aco as ref_aco  # noqa: E402
import utils as ref_utils  # noqa: E402


def cost64(dist, routes):
    """sum of d[u_k][u_{k+1}] over every route (float64)."""
    return float(sum(dist[r[:-1], r[1:]].sum() for r in routes))


def as_column(routes, length):
    return ref_aco.merge_subroutes(routes, length, "cpu").numpy()


def main():
    for n, A in ((20, 24), (50, 24), (100, 16)):
        torch.manual_seed(1000 + n)
        np.random.seed(n)
        demands, distances, positions = ref_utils.gen_instance(n, "cpu", True)          # float64, as the reference keeps them
        # a heuristic that is not 1/d (the perturbation stage works on 1/(heu/rowmax + 1e-5)): 1/d with a smooth random factor
        heu = (1.0 / distances) * (0.25 + torch.rand(n + 1, n + 1, dtype=torch.double))
        colony = ref_aco.ACO(distances, demands, n_ants=A, heuristic=heu, swapstar=True, positions=positions)
        paths = colony.gen_path(require_prob=False)                                        # [L, A] int64
        L = paths.shape[0] + 2                                                             # room for the closing depots
        d_np, dem_np, pos_np = colony.distances_cpu, colony.demand_cpu, colony.positions_cpu
        hd_np = colony.heuristic_dist
        limit = max(colony.problem_size, 50)
        cols = {k: [] for k in ("in", "ls10", "ls100", "nls")}
        costs = {k: [] for k in cols}
        for a in range(A):
            p0 = ref_aco.get_subroutes(paths[:, a])
            outs = {"in": p0,
                    "ls10": ref_swapstar.swapstar(dem_np, d_np, pos_np, p0, count=10),
                    "ls100": ref_swapstar.swapstar(dem_np, d_np, pos_np, p0, count=100),
                    "nls": ref_aco.neural_swapstar(dem_np, d_np, hd_np, pos_np, p0, limit=limit)}
            for k, routes in outs.items():
                routes = [torch.as_tensor(r) for r in routes]
                cols[k].append(as_column(routes, L))
                costs[k].append(cost64(d_np, [r.numpy() for r in routes]))
        out = {"demands": dem_np, "distances": d_np, "positions": pos_np, "heuristic": heu.numpy(), "heuristic_dist": hd_np,
               "capacity": np.float64(ref_aco.CAPACITY), "limit": np.int64(limit)}
        for k in cols:
            out["paths_" + k] = np.stack(cols[k], axis=1).astype(np.int64)                 # [L, A], zero-padded route sequences
            out["costs_" + k] = np.asarray(costs[k], dtype=np.float64)
        name = os.path.join(HERE, f"g8_cvrp_ls_n{n}.npz")
        np.savez_compressed(name, **out)
        print(f"{name}: mean cost sampled {np.mean(costs['in']):.4f} | HGS count=10 {np.mean(costs['ls10']):.4f} | "
              f"count=100 {np.mean(costs['ls100']):.4f} | neural_swapstar {np.mean(costs['nls']):.4f}")


if __name__ == "__main__":
    main()
