#!/usr/bin/env python3
"""g6 at n = 30 / 100 / 200 / 300 / 600: the reference's roulette sampler (tsp_nls/aco.py:260-275, `_inference_sample`) pinned where the
kernels' candidate order differs from the index order (this container only: imports the reference, numba replaced by the
identity shim).

The scan kernels walk a row lane by lane (candidate k sits in lane (k / VEC) % lanes: DESIGN.md section 2), the reference
walks it in index order -- with the same uniform the two inverse-CDF draws pick different candidates.  The statement
"the kernels draw the reference's roulette" is therefore pinned on a RELABELLED instance: with sigma = the order in which
a layout walks the candidates, the reference is run on P'[a][b] = P[sigma(a)][sigma(b)] from start sigma^-1(0); its index
order is the kernel's lane order, so its route r' is the kernel's route sigma(r') -- provided no draw falls on a rounding
boundary (the reference accumulates in float64, the kernels add float32 partial sums).  Rows carry exact zeros (10 % of
the entries) and one row is k-sparse (support 40); uniform streams whose route runs into a dead end (every open
candidate has probability 0 -- the reference then returns a non-permutation) are not recorded, and the number of
streams dropped because the scan specification (oracle) disagrees with the reference is printed.

Run:  python tests/golden/gen_g6_wide.py [n ...]   (writes tests/golden/g6w_roulette_n*.npz; default: all five sizes.
n = 30, 100 and 200 were recorded in round 3 on the four- and eight-lane layouts that serve n <= 128 and n <= 256 since then)
"""
import importlib.util
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("DEEPACO_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "shims"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(REF, "tsp_nls"))
import oracle  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_tsp_nls_aco", os.path.join(REF, "tsp_nls", "aco.py"))
nls_aco = importlib.util.module_from_spec(spec)
spec.loader.exec_module(nls_aco)


def layout_order(n, lanes):
    """sigma: candidate ids in the order a layout walks a row -- (lane, chunk, slot); 64 lanes: VEC by n."""
    vec = 4 if lanes < 64 else (4 if n > 128 else (2 if n > 64 else 1))
    k = np.arange(n)
    key = ((k // vec) % lanes) * 1_000_000 + (k // (lanes * vec)) * 100 + (k % vec)
    return np.argsort(key, kind="stable")


def reference_route(prob, start, uniforms):
    it = iter(uniforms.tolist())
    orig = random.random
    nls_aco.random.random = lambda: np.float64(next(it))
    try:
        return np.asarray(nls_aco._inference_sample(prob, start)).astype(np.int64)
    finally:
        nls_aco.random.random = orig


def main():
    for n in ([int(a) for a in sys.argv[1:]] or [30, 100, 200, 300, 600]):
        rng = np.random.default_rng(7000 + n)
        c = rng.random((n, 2)).astype(np.float32)
        d = np.sqrt(((c[:, None] - c[None]) ** 2).sum(-1)).astype(np.float32)
        np.fill_diagonal(d, 1e9)
        P = (1.0 / d).astype(np.float32) * (0.2 + rng.random((n, n))).astype(np.float32)
        P[rng.random((n, n)) < 0.10] = 0.0                                   # exact zeros
        sparse_row = 7
        keep = rng.choice(np.delete(np.arange(n), sparse_row), size=min(40, n // 3), replace=False)
        row = np.zeros(n, dtype=np.float32)
        row[keep] = P[sparse_row, keep] + np.float32(0.01)
        P[sparse_row] = row
        out = {"probmat": P}
        for lanes in ((4 if n <= 128 else 8 if n <= 256 else 32), 64):
            sigma = layout_order(n, lanes)
            rho = np.argsort(sigma)
            Pp = np.ascontiguousarray(P[sigma][:, sigma])
            us, routes, dead, rounding = [], [], 0, 0
            while len(routes) < 6:
                u = rng.random(n - 1).astype(np.float32)                      # float32-representable: the kernels read f32
                rp = reference_route(Pp, int(rho[0]), u.astype(np.float64))
                route = sigma[rp]
                if not np.array_equal(np.sort(route), np.arange(n)):
                    dead += 1
                    continue
                spec_route, _, rc = oracle.tsp_sample_scan_injected(P, u[:, None], fixed_start=0, wave=(lanes == 64))
                if rc != 0 or not np.array_equal(spec_route[:, 0], route):
                    rounding += 1
                    continue
                us.append(u)
                routes.append(route.astype(np.uint16))
            out[f"uniforms_l{lanes}"] = np.stack(us)
            out[f"routes_l{lanes}"] = np.stack(routes)
            print(f"n = {n}, {lanes}-lane order: 6 routes recorded; streams dropped: {dead} dead ends, {rounding} where the scan "
                  f"specification differs from the reference (rounding boundary)")
        np.savez_compressed(os.path.join(HERE, f"g6w_roulette_n{n}.npz"), **out)


if __name__ == "__main__":
    main()
