#!/usr/bin/env python3
"""g11: routes-in / routes-out of the REFERENCE's CVRP local search, for the route-exact restatement (this container only).

Same set-up as gen_g8_cvrp_ls.py (the reference's Python -- swapstar.py, aco.py, utils.py -- imported from /root/reference
over HGS built by `make -C oracle ref` from the reference's own C++), but recorded for MOVE-FOR-MOVE parity:

  as_run_c{0,1,2,100}   swapstar(demand, distances, positions, routes, count)        cvrp_nls/swapstar.py:324-346
  as_run_hd_c10         the same on the heuristic-derived matrix, count = 10          cvrp_nls/aco.py:446
  as_run_nls            neural_swapstar(..., limit = max(n, 50))                      cvrp_nls/aco.py:443-448
  ss{0,1}_c10 / _hd     local_search called with a CORRECTLY laid out AlgorithmParameters (AlgorithmParameters.h:10-28),
                        seed 1, useSwapStar 0 / 1: what the sources mean.  "as run" is what the reference's 10-field ctypes
                        structure (swapstar.py:62-74) makes of it: it equals ss0 on every solution here (the C side reads
                        useSwapStar beyond the structure); this script asserts that.

Committed: instance (float64), sampled route sequences, result sequences (merge_subroutes layout).  Data only.

Run:  make -C oracle ref && python tests/golden/gen_g11_hgs_ls.py
"""
import os
import random
import sys
import tempfile
from ctypes import CDLL, POINTER, Structure, byref, c_char, c_double, c_int

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(os.environ.get("DEEPACO_REFERENCE", "/root/reference"), "cvrp_nls")
LIB = os.path.join(ROOT, "oracle", "_ref", "libhgscvrp.so")
assert os.path.isfile(LIB), "build it first: make -C oracle ref"

scratch = tempfile.mkdtemp(prefix="g11_")
os.makedirs(os.path.join(scratch, "HGS-CVRP-main", "build"))
os.symlink(LIB, os.path.join(scratch, "HGS-CVRP-main", "build", "libhgscvrp.so"))
os.chdir(scratch)
sys.path.insert(0, os.path.join(HERE, "shims"))
sys.path.insert(0, REF)
import swapstar as ref_swapstar  # noqa: E402
ref_swapstar.HGS_LIBRARY_FILEPATH = LIB
import aco as ref_aco  # noqa: E402
import utils as ref_utils  # noqa: E402


class FullAP(Structure):            # the 15 fields of AlgorithmParameters.h:10-28, in order
    _fields_ = [("nbGranular", c_int), ("mu", c_int), ("lambda_", c_int), ("nbElite", c_int), ("nbClose", c_int),
                ("nbIterPenaltyManagement", c_int), ("targetFeasible", c_double), ("penaltyDecrease", c_double),
                ("penaltyIncrease", c_double), ("seed", c_int), ("nbIter", c_int), ("nbIterTraces", c_int),
                ("timeLimit", c_double), ("useSwapStar", c_int)]


_lib = CDLL(LIB)
_dp = POINTER(c_double)
_lib.local_search.argtypes = [c_int, _dp, _dp, _dp, _dp, _dp, c_double, c_double, c_char, c_int, POINTER(FullAP), c_char,
                              c_int, c_int]


def ls_full(dem, mat, pos, routes, count, seed, use_swap_star):
    """C_Interface.cpp:128-172 with the structure laid out as the header declares it."""
    n = len(dem)
    ap = FullAP(20, 25, 40, 4, 5, 100, 0.2, 0.85, 1.2, seed, 20000, 500, 0.0, use_swap_star)
    callid = random.randint(0, 2 ** 30)
    ref_swapstar.write_routes(routes, f"/tmp/route-{callid}")
    x = np.ascontiguousarray(pos[:, 0], dtype=np.float64)
    y = np.ascontiguousarray(pos[:, 1], dtype=np.float64)
    m = np.ascontiguousarray(mat, dtype=np.float64).reshape(-1)
    s = np.zeros(n)
    d = np.ascontiguousarray(dem * 1000, dtype=np.float64)
    _lib.local_search(n, x.ctypes.data_as(_dp), y.ctypes.data_as(_dp), m.ctypes.data_as(_dp), s.ctypes.data_as(_dp),
                      d.ctypes.data_as(_dp), 1000.001, sys.float_info.max, b'\0', len(routes), byref(ap), b'\0', callid, count)
    res = ref_swapstar.read_routes(f"/tmp/swapstar-result-{callid}")
    os.remove(f"/tmp/swapstar-result-{callid}")
    os.remove(f"/tmp/route-{callid}")
    return res


def main():
    for n, A, seed in ((20, 16, 0), (50, 12, 1), (100, 10, 2), (200, 4, 3)):
        torch.manual_seed(1100 + n + seed)
        np.random.seed(n)
        demands, distances, positions = ref_utils.gen_instance(n, "cpu", True)
        heu = (1.0 / distances) * (0.25 + torch.rand(n + 1, n + 1, dtype=torch.double))
        colony = ref_aco.ACO(distances, demands, n_ants=A, heuristic=heu, swapstar=True, positions=positions)
        paths = colony.gen_path(require_prob=False)
        L = paths.shape[0] + 2
        d, de, po, hd = colony.distances_cpu, colony.demand_cpu, colony.positions_cpu, colony.heuristic_dist
        limit = max(colony.problem_size, 50)
        keys = ["as_run_c0", "as_run_c1", "as_run_c2", "as_run_c100", "as_run_hd_c10", "as_run_nls",
                "ss0_c10", "ss1_c10", "ss0_hd_c10", "ss1_hd_c10"]
        cols = {k: [] for k in keys}
        for a in range(A):
            p0 = ref_aco.get_subroutes(paths[:, a])
            outs = {f"as_run_c{c}": ref_swapstar.swapstar(de, d, po, p0, count=c) for c in (0, 1, 2, 100)}
            outs["as_run_hd_c10"] = ref_swapstar.swapstar(de, hd, po, p0, count=10)
            outs["as_run_nls"] = ref_aco.neural_swapstar(de, d, hd, po, p0, limit=limit)
            for sw in (0, 1):
                outs[f"ss{sw}_c10"] = ls_full(de, d, po, p0, 10, 1, sw)
                outs[f"ss{sw}_hd_c10"] = ls_full(de, hd, po, p0, 10, 1, sw)
            for k in keys:
                cols[k].append(ref_aco.merge_subroutes([torch.as_tensor(r) for r in outs[k]], L, "cpu").numpy())
        out = {"demands": de, "distances": d, "positions": po, "heuristic_dist": hd, "limit": np.int64(limit),
               "paths_in": paths.numpy().astype(np.int16)}
        for k in keys:
            out["paths_" + k] = np.stack(cols[k], axis=1).astype(np.int16)
        # "as run" is "no SWAP*" (the structure mismatch): asserted, not assumed
        assert np.array_equal(out["paths_ss0_hd_c10"], out["paths_as_run_hd_c10"])
        c10 = np.stack([ref_aco.merge_subroutes([torch.as_tensor(r) for r in
                                                 ref_swapstar.swapstar(de, d, po, ref_aco.get_subroutes(paths[:, a]), count=10)],
                                                L, "cpu").numpy() for a in range(A)], axis=1)
        assert np.array_equal(out["paths_ss0_c10"], c10)
        differs = int((out["paths_ss1_c10"] != out["paths_ss0_c10"]).any(axis=0).sum())
        path = os.path.join(HERE, f"g11_hgs_ls_n{n}.npz")
        np.savez_compressed(path, **out)
        print(f"n={n}: {A} solutions, L={L}, SWAP* changes {differs} of them -> {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    main()
