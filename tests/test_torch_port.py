"""The torch-CPU port used as bench.py's cpu_baseline reproduces the reference bit for bit.

The G1 fixtures record the torch seed the reference ran under.  If this torch build replays
the recorded exponential_ noise (same RNG stream), the port must return the reference's tours
exactly; on a build whose stream differs the comparison is meaningless and the test skips."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import torch_port


@pytest.mark.parametrize("name", ["g1_tsp_n20_a8_inv", "g1_tsp_n20_a8_learned", "g1_tsp_n37_a5_inv",
                                  "g1_tsp_n100_a16_learned"])
def test_port_replays_reference(name):
    g = load_golden(name)
    n, A = g["paths"].shape
    seed = int(g["seed"])
    torch.manual_seed(seed)
    start = torch.randint(low=0, high=n, size=(A,))
    q0 = torch.empty(A, n).exponential_(1)
    if not (np.array_equal(start.numpy(), g["start"]) and np.array_equal(q0.numpy(), g["noise"][0])):
        pytest.skip("this torch build's CPU RNG stream differs from the one the fixtures were recorded with")
    torch.manual_seed(seed)
    paths, logp = torch_port.rollout(torch.from_numpy(g["pheromone"]), torch.from_numpy(g["heuristic"]), A,
                                     require_prob=True)
    assert np.array_equal(paths.numpy(), g["paths"])
    assert np.array_equal(logp.numpy(), g["log_probs"])
    costs = torch_port.tour_lengths(torch.from_numpy(g["distances"]), paths)
    assert np.array_equal(costs.numpy(), g["costs"])


def test_port_deposit_matches_reference():
    for name in ("g2_tsp_as_n50_a64", "g2_tsp_elitist_n50_a16", "g2_tsp_as_n3_a4"):
        g = load_golden(name)
        out = torch_port.deposit(torch.from_numpy(g["pheromone_in"]), torch.from_numpy(g["paths"]),
                                 torch.from_numpy(g["costs"]), float(g["decay"]), bool(g["elitist"]))
        assert np.array_equal(out.numpy(), g["pheromone_out"])
