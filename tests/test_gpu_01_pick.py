"""GPU tests of the step-wise draw service (daco_prob_matrix + daco_pick_move) and the generalised
directed deposit that the sibling problems use."""
import numpy as np
import pytest
import torch

import oracle
from conftest import load_golden

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev())


@pytest.mark.parametrize("name", ["g1_tsp_n20_a8_learned", "g1_tsp_n50_a16_sparse", "g1_tsp_n100_a16_learned"])
def test_stepwise_tsp_reproduces_reference(name):
    """tsp/aco.py gen_path rebuilt from single pick_move calls with the reference's noise = same tours."""
    from deepaco_amd import engine
    g = load_golden(name)
    n, A = g["paths"].shape
    svc = engine.PickService(T(g["pheromone"]), T(g["heuristic"]), A)
    prev = T(g["start"])
    mask = torch.ones(A, n, device=dev())
    mask[torch.arange(A), prev] = 0
    tour, lps = [prev], []
    for t in range(1, n):
        act, lp, _ = svc.pick(prev, mask, t, require_prob=True, noise=T(g["noise"][t - 1]))
        tour.append(act)
        lps.append(lp)
        mask[torch.arange(A), act] = 0
        prev = act
    assert np.array_equal(torch.stack(tour).cpu().numpy(), g["paths"])
    np.testing.assert_allclose(torch.stack(lps).cpu().numpy(), g["log_probs"], atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("mode", ["scan", "race"])
@pytest.mark.parametrize("n,A,B", [(7, 5, 1), (64, 9, 2), (100, 33, 1), (300, 8, 2), (700, 4, 1)])
def test_pick_move_vs_oracle(mode, n, A, B):
    from deepaco_amd import engine
    g = torch.Generator().manual_seed(n)
    tau = torch.rand(B, n, n, generator=g) + 0.1
    eta = torch.rand(B, n, n, generator=g) + 1e-3
    prev = torch.randint(0, n, (B, A), generator=g)
    mask = (torch.rand(B, A, n, generator=g) > 0.4).float()
    mask[:, :, 0] = 1
    svc = engine.PickService(tau.to(dev()), eta.to(dev()), A, mode=mode, seed=99, it=4, ant_gid0=10)
    for step in (1, 5, 300):
        act, lp, _ = svc.pick(prev.to(dev()), mask.to(dev()), step, require_prob=True)
        for b in range(B):
            P = oracle.prob_matrix(tau[b].numpy(), eta[b].numpy())
            ra, rl, rc = oracle.pick_move(P, prev[b].numpy(), mask[b].numpy(), mode, seed=99, it=4, ant_gid0=10 + b * A,
                                          step=step)
            assert rc == 0 and np.array_equal(act[b].cpu().numpy(), ra), (mode, n, step)
            np.testing.assert_allclose(lp[b].cpu().numpy(), rl, atol=2e-6, rtol=1e-5)
    assert int(svc.flags.sum()) == 0
    # an ant with nothing open raises the flag
    empty = torch.zeros(B, A, n)
    svc.pick(prev.to(dev()), empty.to(dev()), 2)
    assert int(svc.flags.sum()) == B


@pytest.mark.parametrize("hub", [-1, 0, 11])
def test_directed_deposit_weights_and_hub(hub):
    """Directed deposit with explicit weights: node `hub` may be left many times (and (hub,hub) repeats
    collapse), the others at most once; ants may stop early (no successor)."""
    from deepaco_amd import engine
    n, A, L = 12, 9, 20
    rng = np.random.default_rng(hub + 5)
    paths = np.zeros((L, A), np.int64)
    for a in range(A):
        others = [v for v in rng.permutation(n) if v != hub]
        seq = []
        if hub >= 0:
            seq.append(hub)
            for v in others[: rng.integers(3, n - 1)]:
                seq.append(v)
                if rng.random() < 0.35:
                    seq.append(hub)
            seq += [hub] * (L - len(seq))
        else:
            seq = list(rng.permutation(n)) + [0] * (L - n)
        paths[:, a] = seq[:L]
    if hub < 0:
        paths = paths[:n]
    tau = (rng.random((n, n)) + 0.1).astype(np.float32)
    costs = (rng.random(A) + 0.5).astype(np.float32)
    w = (rng.random(A) * 0.3).astype(np.float32)
    for elitist in (False, True):
        for weights in (None, w):
            t = T(tau)[None].clone().contiguous()
            engine.pheromone_update_(t, T(paths)[None], T(costs)[None], 0.9, elitist, False, floor=1e-10,
                                     weights=None if weights is None else T(weights)[None], hub=hub)
            ref = oracle.pheromone_update_directed(tau, paths, costs, 0.9, weights, elitist, floor=1e-10)
            assert np.array_equal(t[0].cpu().numpy().view(np.uint32), ref.view(np.uint32)), (hub, elitist, weights is None)
