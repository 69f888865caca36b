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
