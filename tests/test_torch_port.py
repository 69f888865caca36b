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


@pytest.mark.parametrize("name,seed", [("g1_cvrp_n20_a8", 81), ("g1_cvrp_n50_a8", 82), ("g1_cvrp_n20_a8_cap20", 83)])
def test_cvrp_port_replays_reference(name, seed):
    """The CVRP port (bench.py's cpu_baseline for config 4) under the seed tests/golden/gen_golden.py ran the reference
    with (cvrp/aco.py:138-205 consumes the RNG through Categorical.sample only): the reference's routes, costs and
    directed deposit."""
    g = load_golden(name)
    A = g["paths"].shape[1]
    torch.manual_seed(seed)
    q0 = torch.empty(A, g["distances"].shape[0]).exponential_(1)
    if not np.array_equal(q0.numpy(), g["noise"][0]):
        pytest.skip("this torch build's CPU RNG stream differs from the one the fixtures were recorded with")
    torch.manual_seed(seed)
    paths = torch_port.cvrp_rollout(torch.from_numpy(g["pheromone"]), torch.from_numpy(g["heuristic"]),
                                    torch.from_numpy(g["demand"]), float(g["capacity"]), A)
    assert np.array_equal(paths.numpy(), g["paths"])
    costs = torch_port.route_lengths(torch.from_numpy(g["distances"]), paths)
    assert np.array_equal(costs.numpy(), g["costs"])
    tau = torch_port.deposit_directed(torch.from_numpy(g["pheromone"]), paths, costs, float(g["decay"]))
    assert np.array_equal(tau.numpy(), g["pheromone_as"])


def test_oracle_nls_driver_matches_reference():
    """oracle.nls_batch (bench.py's cpu_baseline for config 3: the C restatement of two_opt.py under the NLS schedule of
    tsp_nls/aco.py:241-258) against the reference's NLS output (o4)."""
    import oracle
    g = load_golden("o4_nls_n40_a6")
    out, sweeps = oracle.nls_batch(g["distances"], g["heuristic_dist"], g["paths"].T.astype(np.uint16), 40 // 4)
    assert np.array_equal(out.T.astype(np.int64), g["nls_paths"]) and sweeps >= 21 * g["paths"].shape[1]
