"""Randomised parity soak through the C ABI (tests/soak_parity.py): random sizes across the three lane layouts,
ant counts, seeds and value distributions (uniform, heavy-tailed, sparse with exact zeros, denormal-scale), TSP
and CVRP, infeasible draws included -- every tour must equal the oracle's."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [3, 20240917])
def test_random_cases_match_the_oracle(seed):
    import soak_parity
    assert soak_parity.run(250, seed, save_failures=False) == 0


def test_random_updates_and_two_opt_match_the_oracle():
    import soak_parity
    assert soak_parity.run_updates(60, 11) == 0


def test_random_multi_instance_launches_match_the_oracle():
    import soak_parity
    assert soak_parity.run_batches(60, 5) == 0


def test_random_fused_siblings_equal_their_stepwise_paths():
    import soak_parity
    assert soak_parity.run_siblings(40, 9) == 0


def test_random_gradients_match_the_closed_form():
    import soak_parity
    assert soak_parity.run_grads(30, 4) == 0


def test_random_fused_sibling_gradients_equal_the_stepwise_autograd():
    import soak_parity
    assert soak_parity.run_siblings(30, 17, grad=True) == 0


def test_regression_soak_case_1971():
    """Found by `python tests/soak_parity.py 6000 20260927` in round 2: a draw whose threshold no running sum of the
    chosen lane reaches (rounding of the scan) AND whose lane ends in a positive weight small enough to be absorbed by
    the sum before it.  The rule is "the lane's last open candidate with p > 0"; the first in-lane search took "where
    the running sum reaches its final value", which is a different slot here (n = 501, two ants per wavefront)."""
    import numpy as np
    import torch
    import oracle
    from conftest import load_golden
    from deepaco_amd import engine
    g = load_golden("r2_soak_case_1971")
    P, A, seed, it = g["P"], int(g["A"]), int(g["seed"]), int(g["it"])
    n = P.shape[0]
    dev = torch.device("cuda:0")
    for wave in (False, True):
        paths, _, _, flags = engine.tsp_sample(torch.from_numpy(P)[None].to(dev), torch.ones(1, n, n, device=dev), A,
                                               mode="scan_wave" if wave else "scan", seed=seed, it=it)
        rp, _, rc = oracle.tsp_sample_scan(P, A, seed, it, wave=wave)
        assert (int(flags[0]) != 0) == (rc != 0)
        assert np.array_equal(paths[0].cpu().numpy(), rp), wave
