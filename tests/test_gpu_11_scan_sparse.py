"""sampler "scan_sparse" (csrc/daco_scan_sparse.hip, include/deepaco_hip.h daco_tsp_sample_sparse) through the C ABI against its
CPU restatement (oracle draw_scan_sparse, which tests/test_scan_sparse_oracle.py holds against the reference's categorical,
tsp/aco.py:165-177): tours bit for bit -- head steps, tail walks with rejections, dense steps --, the counts of the three
kinds of step, the fused tour lengths and the update's table; and the colony surface."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def instance(n, seed, kind, B=1):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(B, n, 2, generator=g)
    d = (c[:, :, None] - c[:, None]).norm(dim=-1)
    i = torch.arange(n)
    d[:, i, i] = 1e9
    tau = 0.5 + torch.rand(B, n, n, generator=g)
    if kind == "ksparse":                                  # tsp/aco.py:52-67
        k = max(5, n // 10)
        _, idx = torch.topk(d, k=k, dim=2, largest=False)
        eta = 1 / torch.full_like(d, 1e10).scatter_(2, idx, torch.gather(d, 2, idx))
        heads = [oracle.sparse_head_ids(eta[b].numpy(), min(k, 127)) for b in range(B)]      # (k >= 64: the 128-slot head)
    elif kind == "random_head":                            # every entry matters, the head an arbitrary subset: tail walks, rejections
        eta = 1 / d
        heads = []
        rng = np.random.default_rng(seed)
        for b in range(B):
            ids = np.zeros((n, 64), dtype=np.uint16)
            cnt = rng.integers(1, 40, n).astype(np.uint8)
            for r in range(n):
                ids[r, :cnt[r]] = np.sort(rng.choice(n, int(cnt[r]), replace=False))
            heads.append((ids, cnt))
    elif kind == "wide_random_head":                       # the same with 128 slots (eight per lane), 50 .. 126 of them live
        eta = 1 / d
        heads = []
        rng = np.random.default_rng(seed)
        for b in range(B):
            ids = np.zeros((n, 128), dtype=np.uint16)
            cnt = rng.integers(50, 127, n).astype(np.uint8)
            for r in range(n):
                ids[r, :cnt[r]] = np.sort(rng.choice(n, int(cnt[r]), replace=False))
            heads.append((ids, cnt))
    else:                                                  # tiny head: exhausted after a few steps -> dense steps
        eta = 1 / d
        heads = [oracle.sparse_head_ids(eta[b].numpy(), 3) for b in range(B)]
    return d, tau, eta.contiguous(), heads


def pack(heads):
    out = []
    for ids, cnt in heads:
        h = ids.astype(np.int64).copy()
        h[:, h.shape[1] - 1] = cnt
        out.append(h)
    return torch.from_numpy(np.stack(out)).to(torch.int16).contiguous().to(dev())


@pytest.mark.parametrize("n,A,B,kind,fixed", [(160, 40, 1, "random_head", 0), (200, 64, 2, "ksparse", -1), (300, 21, 1, "tiny_head", 3),
                                            (500, 32, 2, "ksparse", -1), (512, 16, 1, "random_head", -1), (513, 19, 1, "ksparse", 0),
                                            (1000, 12, 1, "ksparse", -1), (777, 9, 2, "random_head", 5), (129, 33, 1, "tiny_head", -1),
                                            (300, 20, 1, "wide_random_head", 0), (700, 16, 2, "ksparse", -1), (640, 9, 2, "wide_random_head", -1),
                                            (1024, 8, 1, "wide_random_head", 7)])
def test_scan_sparse_bit_exact_vs_oracle(n, A, B, kind, fixed):
    from deepaco_amd import engine
    d, tau, eta, heads = instance(n, 100 + n, kind, B)
    paths, flags, costs, nbr, stats = engine.tsp_sample_sparse(tau.to(dev()), eta.to(dev()), A, pack(heads), seed=77, it=3,
                                                               fixed_start=fixed, dist=d.to(dev()), want_nbr=True, want_stats=True)
    assert int(flags.sum()) == 0
    ref_stats = np.zeros(3, dtype=np.int64)
    for b in range(B):
        P = oracle.prob_matrix(tau[b].numpy(), eta[b].numpy())
        ref, rc, st = oracle.tsp_sample_scan_sparse(P, heads[b][0], heads[b][1], A, seed=77, it=3, ant_gid0=b * A, fixed_start=fixed)
        assert rc == 0
        got = paths[b].cpu().numpy()
        bad = np.nonzero((got != ref).any(axis=0))[0]
        assert bad.size == 0, (n, kind, b, bad[:5], [int(np.nonzero(got[:, a] != ref[:, a])[0][0]) for a in bad[:5]])
        ref_stats += st
        ref_costs = oracle.tour_costs(d[b].numpy(), ref, closed=True)
        assert np.array_equal(costs[b].cpu().numpy().view(np.int32), np.asarray(ref_costs, dtype=np.float32).view(np.int32))
        nb = nbr[b].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        for a in (0, A - 1):
            t = ref[:, a]
            assert np.array_equal(nb[t, a] & 0xFFFF, np.roll(t, 1)) and np.array_equal(nb[t, a] >> 16, np.roll(t, -1))
    assert np.array_equal(stats.cpu().numpy(), ref_stats), (stats.cpu().numpy(), ref_stats)
    if kind in ("random_head", "wide_random_head"):
        assert ref_stats[1] > 0 and ref_stats[2] > 0
    if kind == "tiny_head":
        assert ref_stats[0] > 0


def test_scan_sparse_headline_shape_properties_and_distribution():
    """TSP-500 x 512 ants x 4 instances, the headline heuristic (k = 50): permutations, costs equal an independent sum, the
    tail is never walked (its mass is 1e-11 of a row), the dense step is rare; and the tours are as good as the dense
    sampler's (same distribution: mean cost within 1 %)."""
    from deepaco_amd import engine
    B, n, A = 4, 500, 512
    d, _, eta, heads = instance(n, 9, "ksparse", B)
    dd, ee = d.to(dev()), eta.to(dev())
    tau = torch.ones_like(dd)
    head = engine.sparse_head(ee, 50)
    assert torch.equal(head.cpu(), pack(heads).cpu())                    # the device rule = the oracle's rule
    paths, flags, costs, nbr, stats = engine.tsp_sample_sparse(tau, ee, A, head, seed=5, dist=dd, want_nbr=True, want_stats=True)
    assert int(flags.sum()) == 0
    assert bool((paths.sort(dim=1).values == torch.arange(n, device=dev()).view(1, n, 1)).all())
    u = paths.transpose(1, 2)
    ref = torch.stack([dd[b][u[b], torch.roll(u[b], 1, dims=1)].double().sum(1) for b in range(B)])
    torch.testing.assert_close(costs.double(), ref, rtol=1e-5, atol=0)
    st = stats.cpu().numpy()
    assert st[1] == 0 and st[2] == 0 and 0 < st[0] < 0.05 * B * A * n, st
    _, _, _, _, dense_costs, _ = engine.tsp_sample(tau, ee, A, mode="scan", seed=5, dist=dd, want_nbr=True)
    assert abs(float(costs.mean()) / float(dense_costs.mean()) - 1) < 0.01


def test_scan_sparse_colony_surface():
    """BatchedTSP(sampler='scan_sparse') after sparsify(k): iterations run, the best cost improves, pheromone stays symmetric."""
    from deepaco_amd import engine
    B, n, A = 3, 200, 64
    d = instance(n, 3, "ksparse", B)[0].to(dev())
    col = engine.BatchedTSP(d, n_ants=A, seed=2, sampler="scan_sparse")
    col.sparsify(20)
    first = None
    for _ in range(6):
        _, costs = col.step()
        first = float(costs.mean()) if first is None else first
    assert float(col.lowest_cost.mean()) < first
    torch.testing.assert_close(col.pheromone, col.pheromone.transpose(1, 2))
    dense = engine.BatchedTSP(d, n_ants=A, seed=2, sampler="scan")
    dense.sparsify(20)
    for _ in range(6):
        dense.step()
    assert abs(float(col.lowest_cost.mean()) / float(dense.lowest_cost.mean()) - 1) < 0.05


@pytest.mark.parametrize("n,A,B,kind,fixed", [(200, 40, 2, "ksparse", -1), (160, 24, 1, "random_head", 0), (300, 21, 1, "tiny_head", 3),
                                            (500, 32, 2, "ksparse", -1), (1000, 10, 1, "ksparse", -1), (640, 9, 1, "random_head", 2),
                                            (400, 12, 1, "wide_random_head", -1)])
def test_race_on_head_rows_equals_the_dense_race(n, A, B, kind, fixed):
    """daco_tsp_sample_race_head: the exponential race of DACO_RACE_PHILOX (torch.multinomial's arithmetic, tsp/aco.py:174-175)
    generated for the 64 / 128 head slots only, the dense race for an ant whenever a tail candidate could still win.  Same noise
    indexing by node id: the tours must be the dense race's bit for bit -- against the oracle's race and against the dense
    kernel -- whatever the head is (k-sparse: a few dense steps late in the tours; an arbitrary subset of a dense heuristic:
    the bound fails at almost every step)."""
    from deepaco_amd import engine
    d, tau, eta, heads = instance(n, 300 + n, kind, B)
    td, ed, dd = tau.to(dev()), eta.to(dev()), d.to(dev())
    paths, flags, costs, nbr, stats = engine.tsp_sample_sparse(td, ed, A, pack(heads), seed=13, it=2, fixed_start=fixed, dist=dd,
                                                               want_nbr=True, want_stats=True, race=True)
    assert int(flags.sum()) == 0
    dense, _, _, dflags, dcosts, dnbr = engine.tsp_sample(td, ed, A, mode="race", seed=13, it=2, fixed_start=fixed, dist=dd, want_nbr=True)
    assert torch.equal(paths, dense) and torch.equal(costs.view(torch.int32), dcosts.view(torch.int32)) and torch.equal(nbr, dnbr)
    for b in range(B):
        P = oracle.prob_matrix(tau[b].numpy(), eta[b].numpy())
        ref, _, rc = oracle.tsp_sample_race(P, A, seed=13, it=2, ant_gid0=b * A, fixed_start=fixed)
        assert rc == 0 and np.array_equal(paths[b].cpu().numpy(), ref), (n, kind, b)
    st = int(stats[0])
    if kind == "ksparse":                               # (n = 1000: 100 live entries per row, the 128-slot head)
        assert 0 < st < 0.1 * B * A * n
    if kind in ("random_head", "wide_random_head"):
        assert st > 0.5 * B * A * n


def test_scan_sparse_config5_shape_with_the_wide_head():
    """TSP-1000, k = 100 (BASELINE config 5's heuristic) on the 128-slot head: the tail is never walked, dense steps are rare, the
    tours are permutations with the right lengths and as good as the dense sampler's."""
    from deepaco_amd import engine
    B, n, A = 2, 1000, 256
    d, _, eta, heads = instance(n, 21, "ksparse", B)
    dd, ee = d.to(dev()), eta.to(dev())
    tau = torch.ones_like(dd)
    head = engine.sparse_head(ee, 100)
    assert head.shape == (B, n, 128) and torch.equal(head.cpu(), pack(heads).cpu())
    paths, flags, costs, nbr, stats = engine.tsp_sample_sparse(tau, ee, A, head, seed=5, dist=dd, want_nbr=True, want_stats=True)
    assert int(flags.sum()) == 0
    assert bool((paths.sort(dim=1).values == torch.arange(n, device=dev()).view(1, n, 1)).all())
    u = paths.transpose(1, 2)
    ref = torch.stack([dd[b][u[b], torch.roll(u[b], 1, dims=1)].double().sum(1) for b in range(B)])
    torch.testing.assert_close(costs.double(), ref, rtol=1e-5, atol=0)
    st = stats.cpu().numpy()
    assert st[1] == 0 and st[2] == 0 and 0 <= st[0] < 0.05 * B * A * n, st
    _, _, _, _, dense_costs, _ = engine.tsp_sample(tau, ee, A, mode="scan", seed=5, dist=dd, want_nbr=True)
    assert abs(float(costs.mean()) / float(dense_costs.mean()) - 1) < 0.01


def test_race_colony_takes_the_head_rows_after_sparsify():
    """BatchedTSP(sampler='race') after sparsify(k) draws from the head rows: the same colony, iteration for iteration, as one
    that is kept on the dense race kernel."""
    from deepaco_amd import engine
    B, n, A = 2, 300, 64
    d = instance(n, 4, "ksparse", B)[0].to(dev())
    a = engine.BatchedTSP(d, n_ants=A, seed=9, sampler="race")
    a.sparsify(30)
    b = engine.BatchedTSP(d, n_ants=A, seed=9, sampler="race")
    b.sparsify(30)
    b.head_k = None                                      # (stays on the dense kernel)
    for _ in range(4):
        pa, ca = a.step()
        pb, cb = b.step()
        assert torch.equal(pa, pb) and torch.equal(ca, cb)
    assert torch.equal(a.pheromone, b.pheromone)


def test_class_surface_runs_on_head_rows():
    """tsp/test.ipynb's inference pattern on the drop-in class: ACO(..., sampler='scan_sparse'); sparsify(k); run(T) -- the
    colony runs on head / tail rows and lands where the dense sampler lands."""
    from deepaco_amd.tsp.aco import ACO
    n = 300
    d = instance(n, 8, "ksparse", 1)[0][0].to(dev())
    best = {}
    for smp in ("scan_sparse", "scan"):
        aco = ACO(d, n_ants=64, device="cuda:0", sampler=smp, seed=3)
        aco.sparsify(30)
        best[smp] = float(aco.run(8))
        assert aco.shortest_path.sort().values.tolist() == list(range(n))
    assert abs(best["scan_sparse"] / best["scan"] - 1) < 0.06
