"""CPU: the candidate-list 2-opt SPECIFICATION (oracle/two_opt_candidates.py, the numpy restatement of what
csrc/daco_two_opt_nbr.hip does) against the oracle's full evaluation (the reference's algorithm): same tours, same sweep
counts -- symmetric (single walk per node list, closing edge) and general matrices, ties, duplicates, sweep caps."""
import numpy as np
import pytest

import oracle
from oracle import two_opt_candidates as spec


def cases(seed):
    rng = np.random.default_rng(seed)
    for kind in ("euclid", "grid", "rowscaled", "asym", "signed_sym", "tiny"):
        n = int(rng.integers(4, 7)) if kind == "tiny" else int(rng.integers(8, 45))
        c = rng.random((n, 2)).astype(np.float32)
        d = np.sqrt(((c[:, None] - c[None]) ** 2).sum(-1)).astype(np.float32)
        if kind == "grid":
            g = rng.integers(0, 4, size=(n, 2)).astype(np.float32)
            d = np.sqrt(((g[:, None] - g[None]) ** 2).sum(-1)).astype(np.float32)
        elif kind == "rowscaled":
            d = (d * rng.uniform(1, 300, size=(n, 1))).astype(np.float32)
        elif kind == "asym":
            d = (rng.random((n, n)) * 10 ** rng.uniform(-2, 3)).astype(np.float32)
        elif kind == "signed_sym":
            d = rng.uniform(-1, 1, size=(n, n)).astype(np.float32)
            d = ((d + d.T) / 2).astype(np.float32)
        np.fill_diagonal(d, 0.0 if kind == "signed_sym" else 1e9)
        yield kind, d, rng.permutation(n), int(rng.choice([1, 3, 40]))


@pytest.mark.parametrize("seed", range(5))
def test_candidate_list_specification_equals_full_evaluation(seed):
    for kind, d, tour, maxit in cases(seed):
        ref, rs = oracle.two_opt_batch(d, tour[None].astype(np.uint16), maxit)
        out, sweeps = spec.two_opt(d, tour, maxit)
        assert np.array_equal(out, ref[0]) and sweeps == int(rs[0]), (kind, len(tour), maxit)
