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


def test_auto_sampler_picks_head_rows_where_they_apply():
    """sampler='auto' (the colonies' default): head / tail rows after sparsify(k) and on a k-sparse (learned-like) heuristic,
    the dense scan on plain 1/d and outside 129 <= n <= 1024; the auto colony IS the explicit scan_sparse colony, bit for bit."""
    import warnings
    from deepaco_amd import engine
    from deepaco_amd.tsp.aco import ACO
    g = torch.Generator().manual_seed(8)
    c = torch.rand(3, 300, 2, generator=g)
    d = (c[:, :, None] - c[:, None]).norm(dim=-1)
    d[:, torch.arange(300), torch.arange(300)] = 1e9
    d = d.to(dev())
    auto = engine.BatchedTSP(d, n_ants=64, seed=3)
    assert auto.resolved_sampler() == ("scan", None)                       # plain 1/d: a fifth of the mass is in the tail
    auto.sparsify(30)
    assert auto.resolved_sampler() == ("scan_sparse", 30)
    forced = engine.BatchedTSP(d, n_ants=64, seed=3, sampler="scan_sparse")
    forced.sparsify(30)
    dense = engine.BatchedTSP(d, n_ants=64, seed=3, sampler="scan")
    dense.sparsify(30)
    assert dense.resolved_sampler()[0] == "scan"
    for _ in range(4):
        auto.step(); forced.step(); dense.step()
    assert torch.equal(auto.pheromone, forced.pheromone) and torch.equal(auto.shortest_path, forced.shortest_path)
    assert not torch.equal(auto.pheromone, dense.pheromone)                # its own uniform stream
    # a k-sparse heuristic nobody announced (what Net.reshape + 1e-10 hands over): 40 live entries per row
    _, idx = torch.topk(d, k=40, dim=2, largest=False)
    heu = torch.full_like(d, 1e-10).scatter_(2, idx, torch.rand(3, 300, 40, device=dev()) + 0.05)
    learned = engine.BatchedTSP(d, n_ants=64, seed=3, heuristic=heu)
    assert learned.resolved_sampler() == ("scan_sparse", 62)       # (the largest head that leaves a slot free: engine.auto_head_k)
    learned.run(3)
    assert bool((learned.shortest_path.sort(dim=1).values == torch.arange(300, device=dev())).all())
    wide = torch.full_like(d, 1e-10).scatter_(2, torch.topk(d, k=100, dim=2, largest=False).indices, 1.0)
    assert engine.BatchedTSP(d, n_ants=8, heuristic=wide).resolved_sampler() == ("scan_sparse", 127)
    # sizes the head kernels do not cover: the dense scan, also when scan_sparse was asked for (one warning, no exception)
    small = d[:, :100, :100].contiguous()
    col = engine.BatchedTSP(small, n_ants=16, seed=1)
    col.sparsify(10)
    assert col.resolved_sampler() == ("scan", None)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        engine._warned_sparse_range = False
        f = engine.BatchedTSP(small, n_ants=16, seed=1, sampler="scan_sparse")
        f.sparsify(10)
        f.run(2)
        col.run(2)
    assert any("scan_sparse" in str(x.message) for x in w) and torch.equal(f.pheromone, col.pheromone)
    # the class surface: run() and gen_path() of the drop-in ACO follow the same rule
    aco = ACO(d[0], n_ants=32, device="cuda:0", seed=5)
    aco.sparsify(30)
    assert aco.resolved_sampler() == ("scan_sparse", 30)
    aco.run(3)
    p = aco.gen_path(require_prob=False)
    assert sorted(p[:, 0].tolist()) == list(range(300))
    p2, lp = aco.gen_path(require_prob=True)                                 # log-probabilities: the dense scan builds them
    assert lp.shape == (299, 32)


def test_sparse_head_equals_the_oracles_rule_with_ties():
    """engine.sparse_head (top-k selection, no row sort) against oracle.sparse_head_ids: ties at the k-th value go to the
    smaller ids."""
    import numpy as np
    import oracle
    from deepaco_amd import engine
    g = torch.Generator().manual_seed(2)
    w = torch.randint(0, 12, (2, 150, 150), generator=g).float() + 1e-3         # many equal values
    for k in (1, 17, 63, 64, 127):
        got = engine.sparse_head(w.to(dev()), k).cpu().numpy().view(np.uint16)
        for b in range(2):
            want, _ = oracle.sparse_head_ids(w[b].numpy(), k)
            np.testing.assert_array_equal(got[b][:, :k], want[:, :k], err_msg=f"k={k}")
            assert (got[b][:, -1] == k).all()
        # the same table from the sorted top values sampler='auto' already has (no torch.topk of its own), batched and [n, m]
        top = torch.topk(w.to(dev()), 127, dim=-1).values
        assert torch.equal(engine.sparse_head(w.to(dev()), k, top=top).cpu().view(torch.int16), torch.from_numpy(got.view(np.int16)))
        one = engine.sparse_head(w[0].to(dev()), k, top=top[0])
        assert torch.equal(one[0].cpu(), torch.from_numpy(got[0].view(np.int16)))


def _import_cpu_chi2():
    import importlib
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    return importlib.import_module("test_scan_sparse_oracle")


@pytest.mark.parametrize("kind", ["dense_random_head", "ksparse", "tiny_head", "dense_wide_head", "ksparse_wide"])
def test_hip_draws_follow_the_reference_categorical(kind):
    """the HIP kernel's OWN draws against the reference's distribution, not only against the restatement.
    30 000 ants from a fixed start on the five head kinds of tests/test_scan_sparse_oracle.py (arbitrary heads of a dense
    heuristic: tail walks and rejections; the k-sparse heuristic of tsp/aco.py:52-67; an exhausted head: dense steps; both head
    widths): chi-square of the first step and of the conditioned second step against Categorical(P[cur] * mask)
    (tsp/aco.py:165-177), with a seed the CPU suite does not use; and the same launch ant for ant against the restatement."""
    from deepaco_amd import engine
    cpu = _import_cpu_chi2()
    n, A, seed = 160, 30000, 20261
    tau, eta, (hid, cnt) = cpu._instance_parts(n, 5, kind)
    P = oracle.prob_matrix(tau, eta)
    head = pack([(hid, cnt)])
    paths, flags, _, _, stats = engine.tsp_sample_sparse(torch.from_numpy(tau)[None].to(dev()), torch.from_numpy(eta)[None].to(dev()), A,
                                                        head, seed=seed, fixed_start=0, want_stats=True)
    assert int(flags.sum()) == 0
    got = paths[0].cpu().numpy()
    cpu.check_first_two_steps(kind, P, got, stats.cpu().numpy())
    ref, rc, st = oracle.tsp_sample_scan_sparse(P, hid, cnt, A, seed=seed, fixed_start=0)
    assert rc == 0 and np.array_equal(got, ref) and np.array_equal(stats.cpu().numpy(), st)


@pytest.mark.parametrize("mode", ["scan", "race"])
def test_dense_samplers_follow_the_reference_categorical(mode):
    """The same chi-square for the two dense samplers the reference fixtures pin (scan: the roulette arithmetic of
    tsp_nls/aco.py:266-274; race: torch.multinomial's, tsp/aco.py:174-175) with in-kernel Philox noise."""
    from deepaco_amd import engine
    cpu = _import_cpu_chi2()
    n, A = 160, 30000
    tau, eta, _ = cpu._instance_parts(n, 5, "dense_random_head")
    P = oracle.prob_matrix(tau, eta)
    paths, _, _, flags = engine.tsp_sample(torch.from_numpy(tau)[None].to(dev()), torch.from_numpy(eta)[None].to(dev()), A, mode=mode,
                                           seed=31337, fixed_start=0)
    assert int(flags.sum()) == 0
    cpu.check_first_two_steps("dense", P, paths[0].cpu().numpy(), np.zeros(3, dtype=np.int64))


@pytest.mark.parametrize("n,A,B,k,kw", [(500, 48, 3, 50, {}), (333, 20, 2, 30, {"min_max": True}), (640, 17, 2, 100, {"elitist": True}),
                                        (300, 24, 2, 30, {"elitist": True}), (1000, 72, 1, 100, {}),
                                        (200, 33, 1, 20, {"sampler": "race"}), (513, 16, 2, 51, {"alpha": 2, "beta": 2})])
def test_head_rows_from_the_pheromone_update_equal_the_pre_pass(n, A, B, k, kw):
    """Round 6: daco_pheromone_update_heads leaves the NEXT iteration's head rows in the colony's workspace
    (tau is read once per iteration, the pre-pass launch is gone), and daco_tsp_sample_heads(heads_ready = 1) takes them.  The same
    colony with a pre-pass every iteration (fuse_head_rows = False: the round-5 path) must give the same tours, costs, pheromone
    and head rows bit for bit over several iterations -- AS / MMAS (clamp applied before the rows are formed) / elitist, sizes
    with and without 16-byte rows, both head widths, the race on head rows, alpha = beta = 2."""
    from deepaco_amd import engine
    d = instance(n, 40 + n, "ksparse", B)[0].to(dev())
    kw = dict(kw)
    sampler = kw.pop("sampler", "scan_sparse")
    cols = []
    for fuse in (True, False):
        col = engine.BatchedTSP(d, n_ants=A, seed=8, sampler=sampler, **kw)
        col.sparsify(k)
        col.fuse_head_rows = fuse
        cols.append(col)
    for it in range(4):
        (pa, ca), (pb, cb) = cols[0].step(), cols[1].step()
        assert torch.equal(pa, pb) and torch.equal(ca.view(torch.int32), cb.view(torch.int32)), it
        assert torch.equal(cols[0].pheromone.view(torch.int32), cols[1].pheromone.view(torch.int32)), it
        if it:                                             # (exponents other than 1: the update does not form the rows, both colonies pre-pass)
            assert (cols[0]._heads_for is not None) == (kw.get("alpha", 1) == 1) and cols[1]._heads_for is None
    # the rows the update left = the rows a pre-pass forms from the same pheromone (compared as bytes; every slot is written)
    rows = n * 16 * (24 if k <= 63 else 48)
    fused_rows = cols[0]._sparse_ws[:B * rows].clone()
    cols[1].step()                                                      # (its pre-pass runs on the pheromone both colonies hold)
    if cols[0]._heads_for is not None:
        assert torch.equal(fused_rows, cols[1]._sparse_ws[:B * rows])
    cols[0].step()
    # a pheromone the caller touched is not the one the rows were formed from: the version counter sends the step to the pre-pass
    cols[0].pheromone.mul_(1.5)
    cols[1].pheromone.mul_(1.5)
    (pa, _), (pb, _) = cols[0].step(), cols[1].step()
    assert torch.equal(pa, pb)
    cols[0].pheromone = cols[0].pheromone.clone()                      # another tensor object: likewise
    (pa, _), (pb, _) = cols[0].step(), cols[1].step()
    assert torch.equal(pa, pb) and torch.equal(cols[0].pheromone.view(torch.int32), cols[1].pheromone.view(torch.int32))


@pytest.mark.parametrize("n,A,B,kind,fixed", [(500, 512, 1, "ksparse", -1), (200, 40, 2, "ksparse", -1), (333, 21, 1, "ksparse", 0),
                                            (160, 24, 1, "random_head", 0), (512, 16, 1, "random_head", -1), (300, 21, 1, "tiny_head", 3),
                                            (129, 33, 1, "tiny_head", -1), (400, 100, 2, "random_head", 5)])
def test_few_ants_keep_the_head_rows_in_lds_same_tours(n, A, B, kind, fixed):
    """Round 6: a launch of few ants -- the reference's own call pattern, one instance per colony
    (tsp/test.ipynb:66-68) -- takes the LDS-heads variant of scan_sparse_kernel when the caller names the head's live bound
    (head_live_max): one wavefront of four ants per workgroup, the instance's head rows compressed into LDS.  Tours, step
    counters, fused costs and the update's table must be the oracle's AND those of the launch without the bound (the kernel
    that reads its head rows through L2), bit for bit."""
    from deepaco_amd import engine
    d, tau, eta, heads = instance(n, 300 + n, kind, B)
    live = int(max(int(h[1].max()) for h in heads))
    assert live <= 62
    T, E, D, H = tau.to(dev()), eta.to(dev()), d.to(dev()), pack(heads)
    paths, flags, costs, nbr, stats = engine.tsp_sample_sparse(T, E, A, H, seed=91, it=2, fixed_start=fixed, dist=D, want_nbr=True,
                                                               want_stats=True, head_live_max=live)
    assert int(flags.sum()) == 0
    p0, f0, c0, n0, s0 = engine.tsp_sample_sparse(T, E, A, H, seed=91, it=2, fixed_start=fixed, dist=D, want_nbr=True, want_stats=True)
    assert torch.equal(paths, p0) and torch.equal(costs.view(torch.int32), c0.view(torch.int32)) and torch.equal(nbr, n0) and torch.equal(stats, s0)
    ref_stats = np.zeros(3, dtype=np.int64)
    for b in range(B):
        P = oracle.prob_matrix(tau[b].numpy(), eta[b].numpy())
        ref, rc, st = oracle.tsp_sample_scan_sparse(P, heads[b][0], heads[b][1], A, seed=91, it=2, ant_gid0=b * A, fixed_start=fixed)
        assert rc == 0 and np.array_equal(paths[b].cpu().numpy(), ref), (n, kind, b)
        ref_stats += st
        assert np.array_equal(costs[b].cpu().numpy().view(np.int32), np.asarray(oracle.tour_costs(d[b].numpy(), ref), dtype=np.float32).view(np.int32))
    assert np.array_equal(stats.cpu().numpy(), ref_stats)
    # a bound below the table's live counts is reported (bit 2 of the flags), not silently wrong
    if live > 4:
        _, fl_bad, _, _ = engine.tsp_sample_sparse(T, E, A, H, seed=91, it=2, fixed_start=fixed, head_live_max=live - 1)
        assert int(fl_bad[0]) & 4


def test_single_instance_colony_takes_the_lds_heads_and_matches_the_batched_one():
    """BatchedTSP at B = 1 (what ACO(...).run does per instance) passes the head bound: the colony's iterations equal instance 0 of
    a B = 3 colony of the same seed (which is too large for the variant) -- tours, costs and pheromone, three iterations."""
    from deepaco_amd import engine
    n, A, k = 500, 512, 50
    d = instance(n, 77, "ksparse", 3)[0].to(dev())
    one = engine.BatchedTSP(d[:1].contiguous(), n_ants=A, seed=6)
    many = engine.BatchedTSP(d, n_ants=A, seed=6)
    one.sparsify(k)
    many.sparsify(k)
    for _ in range(3):
        (p1, c1), (p3, c3) = one.step(), many.step()
        assert torch.equal(p1[0], p3[0]) and torch.equal(c1[0], c3[0])
    assert torch.equal(one.pheromone[0], many.pheromone[0])


@pytest.mark.parametrize("n,A,B", [(500, 48, 2), (200, 8, 1), (700, 40, 2)])
def test_grouped_table_is_the_classic_table_rearranged(n, A, B):
    """nbr_grouped=True (include/deepaco_hip.h daco_tsp_sample_heads): the update's table as [B][A/8][n][8] -- the same entries as
    the classic [B][n][A] table of the same launch."""
    from deepaco_amd import engine
    d, tau, eta, heads = instance(n, 500 + n, "ksparse", B)
    T, E, D, H = tau.to(dev()), eta.to(dev()), d.to(dev()), pack(heads)
    p0, _, c0, n0 = engine.tsp_sample_sparse(T, E, A, H, seed=4, it=1, dist=D, want_nbr=True)
    p1, _, c1, n1 = engine.tsp_sample_sparse(T, E, A, H, seed=4, it=1, dist=D, want_nbr=True, nbr_grouped=True)
    assert torch.equal(p0, p1) and torch.equal(c0, c1)
    regrouped = n1.reshape(B, A // 8, n, 8).permute(0, 2, 1, 3).reshape(B, n, A)
    assert torch.equal(regrouped, n0)


def test_learned_heuristic_of_one_instance_takes_the_lds_heads():
    """The reference's own inference call (tsp/test.ipynb:31-70: one instance, the network's heuristic + 1e-10, 20-50 ants): sampler
    'auto' now picks the largest head that leaves a slot free and fits a CU's LDS (51 at TSP-500 for the graph's k = 50 live entries
    per row, engine.auto_head_k), so the single-instance colony runs the LDS-heads variant; its iterations equal instance 0 of a
    three-instance colony with the same seed (too many ants for the variant: head rows through L2) -- same tours, costs, pheromone."""
    from deepaco_amd import engine
    n, A, k = 500, 50, 50
    d = instance(n, 55, "ksparse", 3)[0].to(dev())
    _, idx = torch.topk(d, k=k, dim=2, largest=False)
    heu = torch.full_like(d, 1e-10).scatter_(2, idx, torch.rand(3, n, k, device=dev()) + 0.05)
    one = engine.BatchedTSP(d[:1].contiguous(), n_ants=A, seed=8, heuristic=heu[:1].contiguous())
    many = engine.BatchedTSP(d, n_ants=A, seed=8, heuristic=heu)
    assert one.resolved_sampler() == ("scan_sparse", 51) and many.resolved_sampler() == ("scan_sparse", 51)
    for _ in range(3):
        (p1, c1), (p3, c3) = one.step(), many.step()
        assert torch.equal(p1[0], p3[0]) and torch.equal(c1[0], c3[0])
    assert torch.equal(one.pheromone[0], many.pheromone[0])
    assert bool((one.shortest_path.sort(dim=1).values == torch.arange(n, device=dev())).all())


@pytest.mark.parametrize("n,A,B,k,kw", [(500, 64, 3, 50, {}), (200, 24, 2, 20, {"elitist": True}), (300, 40, 2, 30, {"min_max": True}),
                                         (700, 32, 2, 70, {}), (500, 20, 1, 50, {})])
def test_colony_iteration_without_int64_paths(n, A, B, k, kw):
    """BatchedTSP.step(want_paths=False) -- what run() passes: the construction kernel leaves the tours as compact u16 rows in its
    workspace (daco_tsp_sparse_tours_offset) instead of the int64 [B, n, A] tensor, the best tour is copied from those rows
    (daco_track_best_tours16), the deposit takes the table.  Against the colony that materialises the paths every step: the same
    costs, records, best tours and pheromone, iteration by iteration; the compact rows are the paths."""
    from deepaco_amd import engine
    d = instance(n, 7 + n, "ksparse", B)[0].to(dev())
    cols = []
    for _ in range(2):
        c = engine.BatchedTSP(d, n_ants=A, seed=13, sampler="scan_sparse", **kw)
        c.sparsify(k)
        cols.append(c)
    for it in range(5):
        paths, costs = cols[0].step()
        none, costs_c = cols[1].step(want_paths=False)
        assert none is None and torch.equal(costs, costs_c), it
        rows = engine.sparse_tours16(cols[1]._sparse_ws, B, n, A)
        assert torch.equal(rows[:, :, :n].permute(0, 2, 1).to(torch.int64), paths), it
        assert torch.equal(cols[0].lowest_cost, cols[1].lowest_cost) and torch.equal(cols[0].shortest_path, cols[1].shortest_path), it
        assert torch.equal(cols[0].pheromone, cols[1].pheromone), it
    # run() keeps the tours compact; a captured graph replays the same iteration
    a, b = cols
    a.run(4)
    for _ in range(4):
        b.step()
    assert torch.equal(a.lowest_cost, b.lowest_cost) and torch.equal(a.shortest_path, b.shortest_path) and torch.equal(a.pheromone, b.pheromone)
    a.run(5, graph=True)
    for _ in range(5):
        b.step()
    assert torch.equal(a.lowest_cost, b.lowest_cost) and torch.equal(a.shortest_path, b.shortest_path) and torch.equal(a.pheromone, b.pheromone)
