"""GPU parity tests for 2-opt / NLS (tsp_nls/two_opt.py, tsp_nls/aco.py:234-258) via the C ABI."""
import glob
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev())


def tsp_instance(n, seed, B=1):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(B, n, 2, generator=g)
    d = torch.cdist(c, c)
    i = torch.arange(n)
    d[:, i, i] = 1e9
    return d


@pytest.mark.parametrize("name", names("g4_twoopt"))
def test_two_opt_golden(name):
    from deepaco_amd import engine
    g = load_golden(name)
    d = T(g["dist"])
    tours = T(g["tours"].astype(np.int16))
    one = engine.two_opt_(d, tours.clone(), 1)
    assert np.array_equal(one.cpu().numpy().astype(np.uint16), g["after_one"])
    full, sweeps = engine.two_opt_(d, tours.clone(), 10000, want_sweeps=True)
    assert np.array_equal(full.cpu().numpy().astype(np.uint16), g["after_full"])
    assert np.array_equal(sweeps[0].cpu().numpy(), g["sweeps"])
    cap = engine.two_opt_(d, tours.clone(), 5)
    assert np.array_equal(cap.cpu().numpy().astype(np.uint16), g["after_cap5"])
    # the module-level drop-in entry point with numpy in / numpy out, as the reference is called
    from deepaco_amd.tsp_nls.two_opt import batched_two_opt_python
    out = batched_two_opt_python(g["dist"], g["tours"], max_iterations=10000)
    assert out.dtype == np.uint16 and np.array_equal(out, g["after_full"])


@pytest.mark.parametrize("n,Tn,B,maxit", [(4, 3, 1, 50), (5, 4, 2, 50), (33, 6, 1, 1000), (128, 5, 1, 7), (129, 5, 2, 1000),
                                           (257, 4, 1, 30), (500, 6, 1, 125)])
def test_two_opt_vs_oracle(n, Tn, B, maxit):
    from deepaco_amd import engine
    d = tsp_instance(n, 7 + n, B)
    rng = np.random.default_rng(n)
    tours = np.stack([[rng.permutation(n) for _ in range(Tn)] for _ in range(B)]).astype(np.int16)
    out, sweeps = engine.two_opt_(d.to(dev()), T(tours), maxit, want_sweeps=True)
    for b in range(B):
        ref, rs = oracle.two_opt_batch(d[b].numpy(), tours[b].astype(np.uint16), maxit)
        assert np.array_equal(out[b].cpu().numpy().astype(np.uint16), ref), (n, b)
        assert np.array_equal(sweeps[b].cpu().numpy(), rs)


def test_two_opt_asymmetric_matrix_and_properties():
    """The NLS runs 2-opt on a non-symmetric perturbation matrix (where the move evaluation is only a
    heuristic and the search may cycle until the sweep cap): parity and permutation validity there;
    on a symmetric matrix a converged tour is a fixed point."""
    from deepaco_amd import engine
    n, Tn = 200, 64
    g = torch.Generator().manual_seed(5)
    d = torch.rand(n, n, generator=g) + 0.01
    rng = np.random.default_rng(1)
    tours = np.stack([rng.permutation(n) for _ in range(Tn)]).astype(np.int16)
    out = engine.two_opt_(d.to(dev()), T(tours), 10000).cpu().numpy().astype(np.int64)
    ref, _ = oracle.two_opt_batch(d.numpy(), tours.astype(np.uint16), 10000)
    assert np.array_equal(out.astype(np.uint16), ref)
    assert np.array_equal(np.sort(out, axis=1), np.tile(np.arange(n), (Tn, 1)))
    ds = ((d + d.T) / 2).to(dev())
    conv, sweeps = engine.two_opt_(ds, T(tours), 10000, want_sweeps=True)
    assert int(sweeps.max()) < 10000
    again, s2 = engine.two_opt_(ds, conv.clone(), 10000, want_sweeps=True)
    assert torch.equal(again, conv) and bool((s2 == 1).all())


def test_nls_driver_matches_reference():
    """ACO.nls / ACO.two_opt (tsp_nls/aco.py:234-258) against outputs captured from the reference."""
    from deepaco_amd.tsp_nls.aco import ACO
    g = load_golden("o4_nls_n40_a6")
    aco = ACO(T(g["distances"]), n_ants=g["paths"].shape[1], heuristic=T(g["heuristic"]), device="cuda:0")
    np.testing.assert_array_equal(aco.heuristic_dist.cpu().numpy(), g["heuristic_dist"])
    out2 = aco.two_opt(T(g["paths"]))
    assert np.array_equal(out2.cpu().numpy(), g["twoopt_paths"])
    out = aco.nls(T(g["paths"]))
    assert np.array_equal(out.cpu().numpy(), g["nls_paths"])
    np.testing.assert_allclose(aco.gen_path_costs(out).cpu().numpy(), g["nls_costs"], rtol=1e-5)


def test_nls_class_surface_and_recorded_noise():
    """tsp_nls sampler through the class: start 0, double normalisation, (costs, log_probs, paths)."""
    from deepaco_amd.tsp_nls.aco import ACO
    g = load_golden("g1_nls_n50_a16")
    aco = ACO(T(g["distances"]), n_ants=16, heuristic=T(g["heuristic"]), pheromone=T(g["pheromone"]),
              device="cuda:0")
    paths, logp = aco.gen_path(True, _start=T(g["start"]), _noise=T(g["noise"]))
    assert np.array_equal(paths.cpu().numpy(), g["paths"])
    np.testing.assert_allclose(logp.cpu().numpy(), g["log_probs"], atol=2e-6, rtol=1e-5)
    costs, log_probs, p2 = aco.sample()
    assert costs.shape == (16,) and log_probs.shape == (49, 16) and p2.shape == (50, 16)
    assert bool((p2[0] == 0).all())
    c2, p3 = aco.sample_2opt(p2)
    assert bool((c2 <= costs + 1e-4).all())
    low = aco.run(2)
    assert isinstance(low, float) and low <= float(costs.mean())
    low_inf = aco.run(1, inference=True)
    assert low_inf <= low


def test_incremental_equals_full_sweep_kernels():
    """The default incremental kernel and the full-sweep kernels (staged / unstaged) choose the same moves:
    identical tours and sweep counts on ACO-sampled tours, symmetric and perturbation (asymmetric) matrices."""
    from deepaco_amd import engine
    B, n, A = 2, 300, 48
    d = tsp_instance(n, 99, B).to(dev())
    eta = 1 / d
    paths, _, _, _ = engine.tsp_sample(torch.ones_like(d), eta, A, mode="scan", seed=4, fixed_start=0)
    tours = paths.permute(0, 2, 1).to(torch.int16).contiguous()
    hd = (1 / (eta / eta.amax(dim=-1, keepdim=True) + 1e-5)).contiguous()
    results = {}
    try:
        for variant in ("17", "18", "9", "8", "2"):
            os.environ["DACO_TWO_OPT_VARIANT"] = variant
            t1, s1 = engine.two_opt_(d, tours.clone(), n // 4, want_sweeps=True)
            t2, s2 = engine.two_opt_(hd, t1.clone(), 20, want_sweeps=True)
            t3, s3 = engine.two_opt_(d, t2.clone(), 10000, want_sweeps=True)
            results[variant] = (t1, s1, t2, s2, t3, s3)
    finally:
        os.environ.pop("DACO_TWO_OPT_VARIANT", None)
    ref = results["9"]
    for variant, res in results.items():
        for x, y in zip(res, ref):
            assert torch.equal(x, y), variant
    # and the oracle agrees on the first instance
    o1, os1 = oracle.two_opt_batch(d[0].cpu().numpy(), tours[0].cpu().numpy().astype(np.uint16), n // 4)
    assert np.array_equal(ref[0][0].cpu().numpy().astype(np.uint16), o1) and np.array_equal(ref[1][0].cpu().numpy(), os1)


def test_config3_full_size_nls_iteration():
    """BASELINE config 3 at its own sizes (TSP-500, 256 ants, NLS with maxt = n//4, T_nls = 10, T_p = 20), two
    instances: one colony iteration through BatchedTSP(local_search='nls').  Properties: every tour a permutation
    starting at node 0, the NLS never makes a tour longer, costs equal an independent gather-sum; and for a few ants
    the first 2-opt pass (the one deterministic piece: maxt sweeps on the sampled tour) equals the oracle bit for bit."""
    from deepaco_amd import engine
    import oracle
    B, n, A = 2, 500, 256
    g = torch.Generator().manual_seed(33)
    c = torch.rand(B, n, 2, generator=g)
    d = torch.cdist(c, c)
    i = torch.arange(n)
    d[:, i, i] = 1e9
    dev = torch.device("cuda:0")
    dd = d.to(dev)
    col = engine.BatchedTSP(dd, n_ants=A, seed=4, local_search="nls", fixed_start=0)
    col.sparsify(50)
    raw, _, _, _ = engine.tsp_sample(col.pheromone, col.heuristic, A, seed=4, it=0, fixed_start=0, batch=B)
    raw_costs = engine.tour_costs(dd, raw)
    paths, costs = col.step()
    assert bool((paths.sort(dim=1).values == torch.arange(n, device=dev).view(1, n, 1)).all())
    assert bool((paths[:, 0] == 0).all())
    assert bool((costs <= raw_costs + 1e-4).all()) and float(costs.mean()) < 0.75 * float(raw_costs.mean())
    u = paths.transpose(1, 2)
    ref = torch.stack([dd[b][u[b], torch.roll(u[b], 1, dims=1)].double().sum(1) for b in range(B)])
    torch.testing.assert_close(costs.double(), ref, rtol=1e-5, atol=0)
    assert torch.equal(col.lowest_cost, costs.min(dim=1).values)
    # first pass of the local search on the sampled tours vs the oracle (4 ants of instance 1)
    tours = raw[1, :, :4].T.contiguous().to(torch.int16)
    out, sweeps = engine.two_opt_(dd[1], tours.clone(), n // 4, want_sweeps=True, dist_t="symmetric")
    ref_t, ref_s = oracle.two_opt_batch(d[1].numpy(), tours.cpu().numpy().astype(np.uint16), n // 4)
    assert np.array_equal(out.cpu().numpy().astype(np.uint16), ref_t) and np.array_equal(sweeps[0].cpu().numpy(), ref_s)


def test_reference_test_script_call_pattern_with_host_tensors():
    """tsp_nls/test.py:16-33 builds the colony from HOST tensors with device='cpu' (the reference's numba local search
    runs there): ACO(n_ants, heuristic=heu_mat.cpu(), distances=distances.cpu(), device='cpu', local_search='nls'),
    then sample(inference=True) and run(inference=True).  The same calls work unchanged: the inputs are staged to
    the HIP device once and everything runs there (there is no CPU compute path)."""
    from deepaco_amd.tsp_nls.aco import ACO
    n = 60
    g = torch.Generator().manual_seed(12)
    c = torch.rand(n, 2, generator=g)
    d = torch.cdist(c, c)
    d[torch.arange(n), torch.arange(n)] = 1e9
    heu = 1 / d + 1e-10
    aco = ACO(n_ants=20, heuristic=heu.cpu(), distances=d.cpu(), device='cpu', local_search='nls', seed=3)
    assert aco.distances.is_cuda and aco.heuristic.is_cuda
    costs = aco.sample(inference=True)[0]
    baseline, best_sample = costs.mean().item(), torch.min(costs).item()
    best_1 = aco.run(n_iterations=1, inference=True)
    best_T = aco.run(n_iterations=4, inference=True)
    assert isinstance(best_1, float) and best_T <= best_1 <= baseline + 1e-6 and best_sample <= baseline


@pytest.mark.parametrize("n,Tn,B,maxit", [(4, 3, 1, 50), (5, 4, 2, 50), (33, 6, 1, 1000), (129, 5, 2, 1000), (257, 4, 1, 30),
                                           (500, 6, 1, 125), (1000, 2, 1, 40)])
def test_candidate_list_kernel_vs_oracle(n, Tn, B, maxit):
    """daco_two_opt_nbr (neighbour-list pruning) makes the reference's moves: tours and sweep counts equal the oracle's
    full evaluation bit for bit, from random permutations (every pair is a candidate at first) to convergence."""
    from deepaco_amd import engine
    d = tsp_instance(n, 7 + n, B)
    rng = np.random.default_rng(n)
    tours = np.stack([[rng.permutation(n) for _ in range(Tn)] for _ in range(B)]).astype(np.int16)
    dd = d.to(dev())
    tabs = engine.TwoOptTables(dd)
    refs = [oracle.two_opt_batch(d[b].numpy(), tours[b].astype(np.uint16), maxit) for b in range(B)]
    for kernel, wide in (("nbr", "0"), ("nbr", "1"), ("auto", "0"), ("auto", "1")):     # 256 / 1024 threads per tour
        os.environ["DACO_TWO_OPT_WIDE"] = wide
        try:
            out, sweeps = engine.two_opt_(dd, T(tours), maxit, want_sweeps=True, tables=tabs, kernel=kernel)
        finally:
            os.environ.pop("DACO_TWO_OPT_WIDE")
        for b in range(B):
            ref, rs = refs[b]
            assert np.array_equal(out[b].cpu().numpy().astype(np.uint16), ref), (n, b, kernel, wide)
            assert np.array_equal(sweeps[b].cpu().numpy(), rs), (kernel, wide)


@pytest.mark.parametrize("wide", ["0", "1"])
def test_candidate_list_kernel_on_perturbation_and_asymmetric_matrices(wide, monkeypatch):
    """The NLS schedule (2-opt on dist, 20 sweeps on the asymmetric heuristic-derived matrix, 2-opt on dist again) with
    the candidate-list kernel equals the dense kernels at every stage; a random non-symmetric matrix with values of very
    different magnitude (tolerance ranks from the largest entry) and a shared [n,n] matrix as well."""
    from deepaco_amd import engine
    monkeypatch.setenv("DACO_TWO_OPT_WIDE", wide)                # 256 / 1024 threads per tour
    B, n, A = 2, 300, 48
    d = tsp_instance(n, 99, B).to(dev())
    eta = 1 / d
    paths, _, _, _ = engine.tsp_sample(torch.ones_like(d), eta, A, mode="scan", seed=4, fixed_start=0)
    tours = paths.permute(0, 2, 1).to(torch.int16).contiguous()
    hd = (1 / (eta / eta.amax(dim=-1, keepdim=True) + 1e-5)).contiguous()
    td, th = engine.TwoOptTables(d), engine.TwoOptTables(hd)
    assert th.tables_t is not th.tables
    t1, s1 = engine.two_opt_(d, tours.clone(), n // 4, want_sweeps=True)
    u1, r1 = engine.two_opt_(d, tours.clone(), n // 4, want_sweeps=True, tables=td, kernel="nbr")
    assert torch.equal(t1, u1) and torch.equal(s1, r1)
    v1, q1 = engine.two_opt_(d, tours.clone(), n // 4, want_sweeps=True, tables=td)          # auto: dense slices first
    assert torch.equal(t1, v1) and torch.equal(s1, q1)
    t2, s2 = engine.two_opt_(hd, t1.clone(), 20, want_sweeps=True)
    u2, r2 = engine.two_opt_(hd, t1.clone(), 20, want_sweeps=True, tables=th, kernel="nbr")
    assert torch.equal(t2, u2) and torch.equal(s2, r2)
    v2, q2 = engine.two_opt_(hd, t1.clone(), 20, want_sweeps=True, tables=th)
    assert torch.equal(t2, v2) and torch.equal(s2, q2)
    t3, s3 = engine.two_opt_(d, t2.clone(), 10000, want_sweeps=True)
    u3, r3 = engine.two_opt_(d, t2.clone(), 10000, want_sweeps=True, tables=td, kernel="nbr")
    assert torch.equal(t3, u3) and torch.equal(s3, r3)
    v3, q3 = engine.two_opt_(d, tours.clone(), 10000, want_sweeps=True, tables=td)           # sampled tours to convergence
    w3, p3 = engine.two_opt_(d, tours.clone(), 10000, want_sweeps=True)
    assert torch.equal(v3, w3) and torch.equal(q3, p3)
    for sw, back in (("3000", "2500"), ("60000", "59000"), ("1", "0")):        # hand-overs at other places / none / never back
        os.environ["DACO_TWO_OPT_SWITCH"], os.environ["DACO_TWO_OPT_BACK"] = sw, back
        try:
            v4, q4 = engine.two_opt_(d, tours.clone(), 10000, want_sweeps=True, tables=td)
            v5, q5 = engine.two_opt_(hd, t1.clone(), 20, want_sweeps=True, tables=th)
        finally:
            os.environ.pop("DACO_TWO_OPT_SWITCH"); os.environ.pop("DACO_TWO_OPT_BACK")
        assert torch.equal(v4, w3) and torch.equal(q4, p3), sw
        assert torch.equal(v5, t2) and torch.equal(q5, s2), sw
    # sparse learned-heuristic style matrix: a plateau of 1e5 off the k-NN graph
    k = 20
    _, idx = torch.topk(d, k=k, dim=2, largest=False)
    h = torch.zeros_like(d).scatter_(2, idx, torch.rand(B, n, k, device=d.device) + 0.05)
    hp = (1 / (h / h.amax(dim=-1, keepdim=True) + 1e-5)).contiguous()
    tp = engine.TwoOptTables(hp)
    t4, s4 = engine.two_opt_(hp, t3.clone(), 20, want_sweeps=True)
    u4, r4 = engine.two_opt_(hp, t3.clone(), 20, want_sweeps=True, tables=tp)
    assert torch.equal(t4, u4) and torch.equal(s4, r4)
    # random asymmetric matrix, one [n,n] matrix shared by a [B,T,n] batch of tours
    g = torch.Generator().manual_seed(5)
    m = (torch.rand(n, n, generator=g) * torch.logspace(-2, 2, n).view(n, 1) + 0.01).to(dev())
    rng = np.random.default_rng(1)
    rt = T(np.stack([[rng.permutation(n) for _ in range(8)] for _ in range(2)]).astype(np.int16))
    t5, s5 = engine.two_opt_(m, rt.clone(), 60, want_sweeps=True)
    u5, r5 = engine.two_opt_(m, rt.clone(), 60, want_sweeps=True, tables=engine.TwoOptTables(m))
    assert torch.equal(t5, u5) and torch.equal(s5, r5)


def test_candidate_list_kernel_many_small_cases_with_ties():
    """Sixty small searches, candidate-list kernel vs dense kernel vs (a third of them) the oracle: integer grid
    coordinates (many equal distances: tie-breaking on (i, j), tolerance ranks with ties), duplicate points (zero
    distances), row-scaled and random asymmetric matrices, signed entries with a zero diagonal, sweep caps that stop
    the search midway."""
    from deepaco_amd import engine
    rng = np.random.default_rng(2026)
    for case in range(60):
        n = int(rng.integers(4, 160))
        kind = case % 5
        if kind == 4:                                            # signed entries, zero diagonal (nothing the kernels assume)
            d = rng.uniform(-1.0, 1.0, size=(n, n)).astype(np.float32)
            if case % 2:
                d = ((d + d.T) / 2).astype(np.float32)
        elif kind == 0:                                          # integer grid: ties everywhere
            c = rng.integers(0, 6, size=(n, 2)).astype(np.float32)
            d = np.sqrt(((c[:, None] - c[None]) ** 2).sum(-1)).astype(np.float32)
        elif kind == 1:                                          # uniform points, a few duplicated
            c = rng.random((n, 2)).astype(np.float32)
            c[rng.integers(0, n, size=max(1, n // 10))] = c[0]
            d = np.sqrt(((c[:, None] - c[None]) ** 2).sum(-1)).astype(np.float32)
        elif kind == 2:                                          # row-scaled distances (the NLS perturbation matrix's shape)
            c = rng.random((n, 2)).astype(np.float32)
            d = np.sqrt(((c[:, None] - c[None]) ** 2).sum(-1)).astype(np.float32)
            d = (d * rng.uniform(1.0, 300.0, size=(n, 1))).astype(np.float32)
        else:                                                    # random asymmetric
            d = (rng.random((n, n)) * 10 ** rng.uniform(-2, 4)).astype(np.float32)
        np.fill_diagonal(d, 0.0 if kind == 4 else 1e9)
        Tn = int(rng.integers(1, 9))
        maxit = int(rng.choice([1, 3, 17, 10000]))
        tours = np.stack([rng.permutation(n) for _ in range(Tn)]).astype(np.int16)
        dd = T(d)
        tabs = engine.TwoOptTables(dd)
        os.environ["DACO_TWO_OPT_WIDE"] = str((case // 4) % 2)   # 256 / 1024 threads per tour
        a, sa = engine.two_opt_(dd, T(tours), maxit, want_sweeps=True)
        b, sb = engine.two_opt_(dd, T(tours), maxit, want_sweeps=True, tables=tabs, kernel="nbr")
        c2, sc = engine.two_opt_(dd, T(tours), maxit, want_sweeps=True, tables=tabs)
        assert torch.equal(a, b) and torch.equal(sa, sb), (case, n, kind, maxit)
        assert torch.equal(a, c2) and torch.equal(sa, sc), (case, n, kind, maxit)
        if case % 3 == 0:
            ref, rs = oracle.two_opt_batch(d, tours.astype(np.uint16), maxit)
            assert np.array_equal(a.cpu().numpy().astype(np.uint16), ref) and np.array_equal(sa[0].cpu().numpy(), rs), (case, n, kind)
    os.environ.pop("DACO_TWO_OPT_WIDE", None)


# ------------------------------------------------------------------ daco_tsp_nls: dirty-list sweeps, the whole NLS in one launch
# threads per tour, entries per thread and round ("64" / "128": one / two wavefronts per tour, n <= 127 / 255 -- the training
# step's tours; larger n falls through to the default choice)
NLS_SHAPES = (("192", "3"), ("192", "4"), ("256", "3"), ("256", "2"), ("256", "4"), ("512", "2"), ("1024", "2"), ("256", "1"), ("64", "2"), ("128", "2"))   # ("192", "4"): config 3's default since round 6; above n = 575 it is the 256 x 4 launch of n > 512


@pytest.mark.parametrize("n,Tn,B,maxit", [(4, 3, 1, 50), (5, 4, 2, 50), (33, 6, 1, 1000), (129, 5, 2, 1000), (257, 4, 1, 30),
                                           (500, 6, 1, 125), (511, 3, 1, 60), (512, 3, 1, 60), (1000, 2, 1, 40),
                                           (1024, 2, 1, 25)])
def test_cached_list_kernel_vs_oracle(n, Tn, B, maxit, monkeypatch):
    """daco_tsp_nls without rounds = one 2-opt search whose sweeps re-walk only the candidate lists the previous move
    can have changed: tours and sweep counts equal the oracle's full evaluation bit for bit, from random permutations
    (long segments, every list dirty) to convergence (short ones), for every thread / queue shape."""
    from deepaco_amd import engine
    d = tsp_instance(n, 7 + n, B)
    rng = np.random.default_rng(n)
    tours = np.stack([[rng.permutation(n) for _ in range(Tn)] for _ in range(B)]).astype(np.int16)
    dd = d.to(dev())
    tabs = engine.TwoOptTables(dd)
    refs = [oracle.two_opt_batch(d[b].numpy(), tours[b].astype(np.uint16), maxit) for b in range(B)]
    for nt, queue in NLS_SHAPES:
        monkeypatch.setenv("DACO_NLS_THREADS", nt)
        monkeypatch.setenv("DACO_NLS_GROUP", queue)
        out, sweeps = engine.two_opt_(dd, T(tours), maxit, want_sweeps=True, tables=tabs, kernel="cached")
        for b in range(B):
            ref, rs = refs[b]
            assert np.array_equal(out[b].cpu().numpy().astype(np.uint16), ref), (n, b, nt, queue)
            assert np.array_equal(sweeps[b].cpu().numpy(), rs), (nt, queue)


def _nls_case(B, n, A, seed, kind):
    from deepaco_amd import engine
    d = tsp_instance(n, seed, B).to(dev())
    if kind == "asym_dist":                      # a distance matrix that is not symmetric (nothing the search assumes)
        g = torch.Generator().manual_seed(seed)
        d = (d * (1 + 0.2 * torch.rand(B, n, n, generator=g).to(dev()))).contiguous()
    eta = 1 / d
    if kind == "sparse":                         # the k-sparse heuristic of config 3: a plateau of 1e5 in the perturbation matrix
        _, idx = torch.topk(d, k=max(4, n // 10), dim=2, largest=False)
        sp = torch.full_like(d, 1e10).scatter_(2, idx, torch.gather(d, 2, idx))
        eta = 1 / sp
    elif kind == "learned":                      # dense positive heuristic of very different magnitudes
        g = torch.Generator().manual_seed(seed + 1)
        eta = (torch.rand(B, n, n, generator=g).to(dev()) ** 4 + 1e-10).contiguous()
    paths, _, _, _ = engine.tsp_sample(torch.ones_like(d), eta if kind != "learned" else 1 / d, A, mode="scan", seed=seed,
                                       fixed_start=0)
    tours = paths.permute(0, 2, 1).to(torch.int16).contiguous()
    hd = (1 / (eta / eta.amax(dim=-1, keepdim=True) + 1e-5)).contiguous()
    if kind == "sym_hd":                         # a symmetric perturbation matrix takes the one-walk path in both kinds of pass
        hd = ((hd + hd.transpose(1, 2)) / 2).contiguous()
    return d, hd, tours


@pytest.mark.parametrize("kind,B,n,A,maxt", [("dense", 2, 300, 24, 75), ("sparse", 2, 500, 16, 125), ("sparse", 1, 200, 40, 10000),
                                              ("learned", 2, 150, 32, 37), ("asym_dist", 2, 120, 24, 30), ("sym_hd", 1, 257, 12, 64),
                                              ("dense", 3, 40, 20, 10), ("sparse", 1, 1000, 3, 250), ("learned", 4, 100, 30, 25),
                                              ("sparse", 2, 127, 16, 31)])
def test_fused_nls_equals_pass_by_pass_driver(kind, B, n, A, maxt, monkeypatch):
    """engine.nls_ fused (one launch of daco_tsp_nls per colony iteration) against the pass-by-pass driver over the
    round-2 kernels -- which test_nls_driver_matches_reference pins on the reference's output and the tests above on
    the oracle: the same tours, and lengths equal to daco_tour_costs bit for bit."""
    from deepaco_amd import engine
    d, hd, tours = _nls_case(B, n, A, 11 + n, kind)
    td, th = engine.TwoOptTables(d), engine.TwoOptTables(hd)
    ref = engine.nls_(d, hd, tours, maxt, tables=td, heuristic_tables=th, fused=False)
    ref_costs = engine.tour_costs(d, ref.permute(0, 2, 1).to(torch.int64).contiguous())
    for nt, queue in NLS_SHAPES:
        monkeypatch.setenv("DACO_NLS_THREADS", nt)
        monkeypatch.setenv("DACO_NLS_GROUP", queue)
        counters = torch.zeros(2, dtype=torch.int64, device=dev())
        out, costs = engine.nls_(d, hd, tours, maxt, tables=td, heuristic_tables=th, fused=True, want_costs=True,
                                 counters=counters)
        assert torch.equal(out, ref), (kind, nt, queue)
        assert torch.equal(costs.view(torch.int32), ref_costs.view(torch.int32)), (kind, nt, queue)
        assert int(counters[0]) >= 21 * B * A and int(counters[1]) > 0
    # other schedules: no rounds, one round, a single perturbation sweep
    for T_nls, T_p in ((0, 20), (1, 1), (3, 5)):
        a = engine.nls_(d, hd, tours, maxt, T_nls=T_nls, T_p=T_p, tables=td, heuristic_tables=th, fused=False)
        b = engine.nls_(d, hd, tours, maxt, T_nls=T_nls, T_p=T_p, tables=td, heuristic_tables=th, fused=True)
        assert torch.equal(a, b), (kind, T_nls, T_p)


@pytest.mark.parametrize("kind", ["sparse", "net"])
def test_fused_nls_vs_oracle_at_config3_size(kind):
    """tsp_nls/aco.py:241-258 at BASELINE config 3's size, the fused kernel (daco_tsp_nls, one launch) directly against
    the oracle's restatement of the schedule (oracle.nls_batch over orc_two_opt_batch = tsp_nls/two_opt.py:6-39): n = 500,
    maxt = n // 4 = 125, T_nls = 10, T_p = 20, eight sampled tours -- the tours and their f32 lengths bit for bit.
    kind = sparse: the k-sparse 1/d heuristic of the reference's inference (perturbation matrix with a plateau);
    kind = net: the heuristic of the pretrained network (the reference's tsp500 checkpoint, w_tsp_tsp500) on the instance."""
    from deepaco_amd import engine
    n, A, maxt = 500, 8, 125
    d = tsp_instance(n, 4242, 1).to(dev())
    if kind == "sparse":
        _, idx = torch.topk(d, k=50, dim=2, largest=False)
        eta = 1 / torch.full_like(d, 1e10).scatter_(2, idx, torch.gather(d, 2, idx))
    else:
        from deepaco_amd.tsp.net import Net
        from deepaco_amd.tsp.utils import gen_pyg_data
        wz = np.load(os.path.join(GOLDEN, "w_tsp_tsp500.npz"))
        net = Net()
        net.load_state_dict({k[3:]: torch.from_numpy(wz[k]) for k in wz.files if k.startswith("w__")}, strict=False)
        net = net.to(dev()).eval()
        g = torch.Generator().manual_seed(4242)
        coords = torch.rand(n, 2, generator=g).to(dev())
        with torch.no_grad():
            pyg, distances = gen_pyg_data(coords, k_sparse=50)
            eta = (net.reshape(pyg, net(pyg)) + 1e-10)[None].contiguous()
        d = distances[None].contiguous()
    paths, _, _, _ = engine.tsp_sample(torch.ones_like(d), eta, A, mode="scan", seed=77, fixed_start=0)
    tours = paths.permute(0, 2, 1).to(torch.int16).contiguous()
    hd = (1 / (eta / eta.amax(dim=-1, keepdim=True) + 1e-5)).contiguous()    # tsp_nls/aco.py:58
    out, costs = engine.nls_(d, hd, tours, maxt, fused=True, want_costs=True)
    ref, _ = oracle.nls_batch(d[0].cpu().numpy(), hd[0].cpu().numpy(), tours[0].cpu().numpy().astype(np.uint16), maxt)
    assert np.array_equal(out[0].cpu().numpy().astype(np.uint16), ref), kind
    ref_costs = oracle.tour_costs(d[0].cpu().numpy(), np.ascontiguousarray(ref.T.astype(np.int64)), closed=True)
    assert np.array_equal(costs[0].cpu().numpy().view(np.int32), np.asarray(ref_costs, dtype=np.float32).view(np.int32)), kind
    # the search did something: every tour got shorter than it was sampled
    raw = oracle.tour_costs(d[0].cpu().numpy(), np.ascontiguousarray(tours[0].cpu().numpy().T.astype(np.int64)), closed=True)
    assert (np.asarray(ref_costs) < 0.9 * np.asarray(raw)).all()


def test_cached_list_kernel_many_small_cases_with_ties(monkeypatch):
    """Ninety small searches, dirty-list kernel vs dense kernel (and the oracle for a third): integer grids (ties on
    (i, j), tolerance ranks with ties), duplicate points, row-scaled and random asymmetric matrices, signed entries,
    sweep caps that stop the search midway, every thread shape."""
    from deepaco_amd import engine
    rng = np.random.default_rng(3031)
    for case in range(90):
        n = int(rng.integers(4, 200))
        kind = case % 5
        if kind == 4:
            d = rng.uniform(-1.0, 1.0, size=(n, n)).astype(np.float32)
            if case % 2:
                d = ((d + d.T) / 2).astype(np.float32)
        elif kind == 0:
            c = rng.integers(0, 6, size=(n, 2)).astype(np.float32)
            d = np.sqrt(((c[:, None] - c[None]) ** 2).sum(-1)).astype(np.float32)
        elif kind == 1:
            c = rng.random((n, 2)).astype(np.float32)
            c[rng.integers(0, n, size=max(1, n // 10))] = c[0]
            d = np.sqrt(((c[:, None] - c[None]) ** 2).sum(-1)).astype(np.float32)
        elif kind == 2:
            c = rng.random((n, 2)).astype(np.float32)
            d = np.sqrt(((c[:, None] - c[None]) ** 2).sum(-1)).astype(np.float32)
            d = (d * rng.uniform(1.0, 300.0, size=(n, 1))).astype(np.float32)
        else:
            d = (rng.random((n, n)) * 10 ** rng.uniform(-2, 4)).astype(np.float32)
        np.fill_diagonal(d, 0.0 if kind == 4 else 1e9)
        Tn = int(rng.integers(1, 9))
        maxit = int(rng.choice([1, 3, 17, 10000]))
        tours = np.stack([rng.permutation(n) for _ in range(Tn)]).astype(np.int16)
        dd = T(d)
        tabs = engine.TwoOptTables(dd)
        nt, queue = NLS_SHAPES[case % len(NLS_SHAPES)]
        monkeypatch.setenv("DACO_NLS_THREADS", nt)
        monkeypatch.setenv("DACO_NLS_GROUP", queue)
        a, sa = engine.two_opt_(dd, T(tours), maxit, want_sweeps=True)
        b, sb = engine.two_opt_(dd, T(tours), maxit, want_sweeps=True, tables=tabs, kernel="cached")
        assert torch.equal(a, b) and torch.equal(sa, sb), (case, n, kind, maxit, nt)
        if case % 3 == 0:
            ref, rs = oracle.two_opt_batch(d, tours.astype(np.uint16), maxit)
            assert np.array_equal(b.cpu().numpy().astype(np.uint16), ref) and np.array_equal(sb[0].cpu().numpy(), rs), (case, n, kind)
