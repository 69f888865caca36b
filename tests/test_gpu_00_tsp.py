"""GPU parity tests for the TSP rollout path, through the C ABI (deepaco_amd.engine -> ctypes ->
libdeepaco_hip.so).  Three layers:
  1. golden vectors captured from the reference (recorded-noise race): tours bit-exact,
     pheromone update bitwise, costs/log-probs to tolerance;
  2. the CPU oracle on seeded inputs in the Philox modes: tours, costs and pheromone bit-exact;
  3. size-independent properties at BASELINE.json's full sizes (TSP-500 x 512 ants).
"""
import glob
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu

RTOL_COST = 1e-5   # north_star: costs within 1e-5 relative
ATOL_LOGP = 2e-6


def dev():
    return torch.device("cuda:0")


def names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev())


def make_instance(n, seed, B=1):
    g = torch.Generator().manual_seed(seed)
    coords = torch.rand(B, n, 2, generator=g)
    dist = torch.cdist(coords, coords)
    idx = torch.arange(n)
    dist[:, idx, idx] = 1e9
    tau = torch.rand(B, n, n, generator=g) + 0.1
    tau = (tau + tau.transpose(1, 2)) / 2
    eta = torch.rand(B, n, n, generator=g) ** 2 + 1e-10
    return dist, tau, eta


# ------------------------------------------------------------------ 1. golden vectors
@pytest.mark.parametrize("name", names("g1_tsp") + names("g1_nls"))
def test_sampler_recorded_noise_bit_exact(name):
    from deepaco_amd import engine
    g = load_golden(name)
    passes = 2 if "nls" in name else 1
    paths, logp, _, flags = engine.tsp_sample(T(g["pheromone"])[None], T(g["heuristic"])[None], g["paths"].shape[1],
                                              mode="race_noise", norm_passes=passes, start=T(g["start"])[None],
                                              noise=T(g["noise"])[None], require_prob=True)
    assert int(flags.sum()) == 0
    assert np.array_equal(paths[0].cpu().numpy(), g["paths"])
    np.testing.assert_allclose(logp[0].cpu().numpy(), g["log_probs"], atol=ATOL_LOGP, rtol=1e-5)
    # the un-normalised race picks the same tours
    p0, _, _, _ = engine.tsp_sample(T(g["pheromone"])[None], T(g["heuristic"])[None], g["paths"].shape[1],
                                    mode="race_noise", norm_passes=0, start=T(g["start"])[None],
                                    noise=T(g["noise"])[None])
    assert np.array_equal(p0[0].cpu().numpy(), g["paths"])
    costs = engine.tour_costs(T(g["distances"])[None], paths)[0].cpu().numpy()
    np.testing.assert_allclose(costs, g["costs"], rtol=RTOL_COST)
    assert np.array_equal(costs, oracle.tour_costs(g["distances"], g["paths"]))


@pytest.mark.parametrize("name", names("g2_tsp"))
def test_update_bitwise_vs_reference(name):
    from deepaco_amd import engine
    g = load_golden(name)
    tau = T(g["pheromone_in"])[None].clone().contiguous()
    cmin = cmax = None
    if "clamp_max" in g:
        cmin, cmax = T(g["clamp_min"]).reshape(1), T(g["clamp_max"]).reshape(1)
    engine.pheromone_update_(tau, T(g["paths"])[None], T(g["costs"])[None], float(g["decay"]),
                             bool(g["elitist"]), True, cmin, cmax)
    assert np.array_equal(tau[0].cpu().numpy().view(np.uint32), g["pheromone_out"].view(np.uint32))


@pytest.mark.parametrize("name", names("u2_tsp"))
def test_aco_class_run_trace(name):
    """ACO.run (tsp/aco.py:75-92) through the drop-in class, the reference's noise injected."""
    from deepaco_amd.tsp.aco import ACO
    g = load_golden(name)
    dist = T(g["distances"])
    kw = dict(elitist=bool(g["elitist"]), min_max=bool(g["min_max"]))
    aco = ACO(dist, n_ants=g["paths"].shape[2], device="cuda:0", **kw)
    for it in range(g["tau_in"].shape[0]):
        # start every iteration from the reference's pheromone: one differing ulp in a cost
        # would otherwise make later iterations incomparable (SURVEY.md 7 'chaos')
        aco.pheromone = T(g["tau_in"][it]).clone()
        orig = aco.gen_path
        start, noise = T(g["start"][it]), T(g["noise"][it])
        aco.gen_path = lambda require_prob=False, _o=orig, _s=start, _q=noise: _o(require_prob, _start=_s, _noise=_q)
        aco.run(1)
        aco.gen_path = orig
        np.testing.assert_allclose(aco.pheromone.cpu().numpy(), g["tau_out"][it], rtol=2e-6)
        np.testing.assert_allclose(float(aco.lowest_cost), float(g["lowest"][it]), rtol=RTOL_COST)
    assert sorted(aco.shortest_path.cpu().tolist()) == list(range(dist.shape[0]))
    ref_sp = g["shortest_path"]
    assert np.array_equal(aco.shortest_path.cpu().numpy(), ref_sp)


# ------------------------------------------------------------------ 2. oracle, Philox modes
SHAPES = [(2, 1, 1), (3, 2, 1), (5, 4, 1), (20, 7, 2), (63, 5, 1), (64, 9, 1), (65, 5, 2), (100, 33, 1),
          (128, 6, 1), (129, 6, 1), (200, 17, 2), (256, 4, 1), (257, 4, 1), (500, 12, 1), (777, 5, 1),
          (1000, 6, 1), (1025, 3, 1),
          # sixteen / eight / two ants per wavefront (n <= 128 / <= 256 / <= 1024): every chunk count, odd ant counts
          (16, 3, 1), (17, 18, 1), (40, 35, 2), (50, 65, 1), (70, 5, 1), (90, 19, 1), (112, 70, 1), (113, 16, 1), (170, 9, 1),
          (224, 11, 1), (225, 8, 1), (130, 3, 1), (384, 7, 1), (385, 5, 2), (512, 9, 1), (640, 3, 1), (700, 4, 1), (896, 3, 1), (1024, 5, 1)]


@pytest.mark.parametrize("mode", ["scan", "race"])
@pytest.mark.parametrize("n,A,B", SHAPES)
def test_sampler_philox_bit_exact_vs_oracle(mode, n, A, B):
    from deepaco_amd import engine
    dist, tau, eta = make_instance(n, 100 + n, B)
    seed, it, gid0 = 0xDEADBEEF1234, 3, 1000
    paths, logp, _, flags = engine.tsp_sample(tau.to(dev()), eta.to(dev()), A, mode=mode, seed=seed, it=it,
                                              ant_gid0=gid0, require_prob=True)
    fn = oracle.tsp_sample_scan if mode == "scan" else oracle.tsp_sample_race
    assert int(flags.sum()) == 0
    for b in range(B):
        P = oracle.prob_matrix(tau[b].numpy(), eta[b].numpy())
        rp, rl, rc = fn(P, A, seed, it, gid0 + b * A, require_prob=True)
        assert rc == 0
        assert np.array_equal(paths[b].cpu().numpy(), rp), (mode, n, b)
        if n > 2:
            np.testing.assert_allclose(logp[b].cpu().numpy(), rl, atol=ATOL_LOGP, rtol=1e-5)


@pytest.mark.parametrize("n,A", [(40, 33), (100, 64), (128, 7), (130, 64), (400, 16), (1030, 4)])
@pytest.mark.parametrize("wave", [False, True])
def test_scan_draw_extreme_rows_bit_exact(n, A, wave):
    """Rows spanning 45 orders of magnitude with a third of exact zeros: every lane layout of the scan draw
    still follows the specified arithmetic (rounding fallbacks, empty lanes, tiny row sums) bit for bit."""
    from deepaco_amd import engine
    rng = np.random.default_rng(1000 * n + A)
    P = np.exp(rng.uniform(-90.0, 12.0, size=(n, n))).astype(np.float32)
    P[rng.random((n, n)) < 0.33] = 0.0
    np.fill_diagonal(P, 0.0)
    idx = np.arange(n)
    for off in (1, 2, 3):                                  # keep most rows feasible until late in the tour
        P[idx, (idx + off) % n] = np.maximum(P[idx, (idx + off) % n], np.float32(1e-30))
    tau = torch.from_numpy(P)[None].contiguous()
    eta = torch.ones(1, n, n)
    mode = "scan_wave" if wave else "scan"
    paths, logp, _, flags = engine.tsp_sample(tau.to(dev()), eta.to(dev()), A, mode=mode, seed=77, it=1, require_prob=True)
    rp, rl, rc = oracle.tsp_sample_scan(P, A, 77, 1, require_prob=True, wave=wave)
    assert int(flags[0]) == (1 if rc else 0)
    # (a draw without feasible candidate is flagged -- the reference raises -- and moves to node 0 in every kernel)
    assert np.array_equal(paths[0].cpu().numpy(), rp)
    if rc == 0:
        np.testing.assert_allclose(logp[0].cpu().numpy(), rl, atol=ATOL_LOGP, rtol=1e-5)


@pytest.mark.parametrize("mode", ["scan", "race"])
def test_fixed_start_and_no_logp(mode):
    from deepaco_amd import engine
    n, A = 150, 10
    dist, tau, eta = make_instance(n, 5)
    paths, logp, _, _ = engine.tsp_sample(tau.to(dev()), eta.to(dev()), A, mode=mode, seed=9, it=0, fixed_start=0)
    assert logp is None and bool((paths[0, 0] == 0).all())
    fn = oracle.tsp_sample_scan if mode == "scan" else oracle.tsp_sample_race
    rp, _, _ = fn(oracle.prob_matrix(tau[0].numpy(), eta[0].numpy()), A, 9, 0, 0, fixed_start=0)
    assert np.array_equal(paths[0].cpu().numpy(), rp)


@pytest.mark.parametrize("B,A", [(1, 1), (2, 7), (3, 16), (2, 33)])
def test_two_ants_per_wave_fused_outputs(B, A):
    """n <= 1024 in scan mode: fused costs and neighbour table of the several-ants-per-wave kernels, any ant
    count; 'scan_wave' keeps the one-ant-per-wave draw (a different but equally valid stream)."""
    from deepaco_amd import engine
    n = 300 if A != 7 else 90
    dist, tau, eta = make_instance(n, 7 * B + A, B)
    out = {}
    for mode in ("scan", "scan_wave"):
        paths, logp, _, flags, costs, nbr = engine.tsp_sample(tau.to(dev()), eta.to(dev()), A, mode=mode, seed=21, it=2,
                                                              dist=dist.to(dev()), want_nbr=True, require_prob=True)
        assert int(flags.sum()) == 0
        for b in range(B):
            rp, rl, _ = oracle.tsp_sample_scan(oracle.prob_matrix(tau[b].numpy(), eta[b].numpy()), A, 21, 2, b * A,
                                               require_prob=True, wave=(mode == "scan_wave"))
            assert np.array_equal(paths[b].cpu().numpy(), rp), (mode, b)
            np.testing.assert_allclose(logp[b].cpu().numpy(), rl, atol=ATOL_LOGP, rtol=1e-5)
            assert np.array_equal(costs[b].cpu().numpy(), oracle.tour_costs(dist[b].numpy(), rp))
        t1 = tau.to(dev()).clone().contiguous()
        t2 = t1.clone()
        engine.pheromone_update_(t1, paths, costs, 0.9, nbr=nbr)
        engine.pheromone_update_(t2, paths, costs, 0.9)
        assert torch.equal(t1, t2)
        out[mode] = paths
    assert not torch.equal(out["scan"], out["scan_wave"])


@pytest.mark.parametrize("elitist,mmas", [(False, False), (True, False), (False, True)])
def test_batched_run_bit_exact_vs_oracle(elitist, mmas):
    """Multi-iteration colony (sample -> costs -> best -> update) == the same loop on the oracle."""
    from deepaco_amd import engine
    B, n, A, iters, seed = 2, 50, 24, 5, 77
    dist, _, _ = make_instance(n, 42, B)
    col = engine.BatchedTSP(dist.to(dev()), n_ants=A, elitist=elitist, min_max=mmas, seed=seed)
    col.run(iters)
    for b in range(B):
        d = dist[b].numpy()
        eta = (1.0 / dist[b]).numpy()
        tau = np.ones((n, n), np.float32) * (np.float32(0.1) if mmas else np.float32(1))
        lowest, cmax = np.float32(np.inf), None
        for it in range(iters):
            paths, _, rc = oracle.tsp_sample_scan(oracle.prob_matrix(tau, eta), A, seed, it, b * A)
            costs = oracle.tour_costs(d, paths)
            if costs.min() < lowest:
                lowest = costs.min()
            if mmas:
                new_max = (np.float32(1) / lowest) * np.float32(n)
                if cmax is None:
                    tau = tau * (new_max / tau.max())
                cmax = new_max
            tau = oracle.pheromone_update_tsp(tau, paths, costs, 0.9, elitist, 0.1 if mmas else 0.0,
                                              float(cmax) if mmas else 0.0)
        assert np.array_equal(col.pheromone[b].cpu().numpy().view(np.uint32), tau.view(np.uint32))
        assert np.float32(col.lowest_cost[b].item()) == lowest


def test_sampler_distribution_chi_square():
    """Both samplers draw the first move from p_k ~ P[start][k] (exact categorical)."""
    from deepaco_amd import engine
    n, A, B = 12, 4096, 8
    g = torch.Generator().manual_seed(1)
    tau = torch.rand(n, n, generator=g) + 0.05
    eta = torch.rand(n, n, generator=g) ** 2 + 1e-3
    p = (tau[0] * eta[0]).double()
    p[0] = 0
    p = p / p.sum()
    for mode in ("scan", "race"):
        paths, _, _, _ = engine.tsp_sample(tau.to(dev()), eta.to(dev()), A, mode=mode, seed=5, fixed_start=0, batch=B)
        first = paths[:, 1, :].reshape(-1).cpu()
        N = first.numel()
        obs = torch.bincount(first, minlength=n).double()
        chi2 = float((((obs - N * p) ** 2) / (N * p).clamp(min=1e-12))[1:].sum())
        assert chi2 < 40.0, (mode, chi2)     # dof = 10: P(chi2 > 40) ~ 2e-5
        # and the second move, conditioned on the first, is uniform-free of bias too: all tours valid
        assert bool((paths.sort(dim=1).values == torch.arange(n, device=dev()).view(1, n, 1)).all())


# ------------------------------------------------------------------ 3. properties at full size
@pytest.mark.parametrize("mode", ["scan", "race"])
def test_full_size_properties_tsp500(mode):
    from deepaco_amd import engine
    B, n, A = 4, 500, 512
    dist, tau, eta = make_instance(n, 2024, B)
    d, t, e = dist.to(dev()), tau.to(dev()), eta.to(dev())
    paths, _, _, flags = engine.tsp_sample(t, e, A, mode=mode, seed=11, it=0)
    assert int(flags.sum()) == 0
    # every tour is a permutation of the nodes
    assert bool((paths.sort(dim=1).values == torch.arange(n, device=dev()).view(1, n, 1)).all())
    # different ants build different tours; different iterations differ
    assert len({tuple(paths[0, :, a].tolist()) for a in range(32)}) > 16
    p2, _, _, _ = engine.tsp_sample(t, e, A, mode=mode, seed=11, it=1)
    assert not torch.equal(paths, p2)
    # same (seed, it) reproduces the tours exactly
    p3, _, _, _ = engine.tsp_sample(t, e, A, mode=mode, seed=11, it=0)
    assert torch.equal(paths, p3)
    # costs against an independent torch gather-sum
    costs = engine.tour_costs(d, paths)
    u = paths.transpose(1, 2)
    v = torch.roll(u, 1, dims=2)
    ref = torch.stack([d[b][u[b], v[b]].double().sum(1) for b in range(B)])
    torch.testing.assert_close(costs.double(), ref, rtol=RTOL_COST, atol=0)
    # update: symmetric stays symmetric; deposited mass = 2n * sum(1/c)
    t2 = t.clone().contiguous()
    engine.pheromone_update_(t2, paths, costs, 0.9)
    assert torch.equal(t2, t2.transpose(1, 2))
    mass = (t2.double() - 0.9 * t.double()).sum(dim=(1, 2))
    torch.testing.assert_close(mass, 2 * n * (1 / costs.double()).sum(1), rtol=1e-4, atol=0)
    # one instance cross-checked bit-exactly against the oracle at full size
    b = 1
    P = oracle.prob_matrix(tau[b].numpy(), eta[b].numpy())
    fn = oracle.tsp_sample_scan if mode == "scan" else oracle.tsp_sample_race
    rp, _, _ = fn(P, 64, 11, 0, b * A)
    assert np.array_equal(paths[b, :, :64].cpu().numpy(), rp)
    rt = oracle.pheromone_update_tsp(tau[b].numpy(), paths[b].cpu().numpy(), costs[b].cpu().numpy(), 0.9)
    assert np.array_equal(t2[b].cpu().numpy().view(np.uint32), rt.view(np.uint32))


@pytest.mark.parametrize("n,A,B", [(500, 512, 4), (400, 256, 2), (1000, 256, 2), (1000, 2048, 2), (100, 512, 8)])
def test_every_ant_of_a_full_size_launch_vs_oracle(n, A, B):
    """All B x A tours of a full-size launch against the oracle (heavy rows with random diagonals: the rare paths
    of the draw -- no running sum reaching the threshold, lane sums that round differently from the scan -- show
    up once in a few thousand tours; they did, in the first version of the in-lane search).  Covers the headline
    shape, config 3's ant count, config 5's shape (TSP-1000 x 2048 ants, two instances) and config 2's shape."""
    from deepaco_amd import engine
    dist, tau, eta = make_instance(n, 2024 + n, B)
    paths, _, _, flags, costs, nbr = engine.tsp_sample(tau.to(dev()), eta.to(dev()), A, mode="scan", seed=11, it=3,
                                                      dist=dist.to(dev()), want_nbr=True)
    assert int(flags.sum()) == 0
    p = paths.cpu().numpy()
    for b in range(B):
        rp, _, rc = oracle.tsp_sample_scan(oracle.prob_matrix(tau[b].numpy(), eta[b].numpy()), A, 11, 3, b * A)
        assert rc == 0
        bad = [a for a in range(A) if not np.array_equal(rp[:, a], p[b, :, a])]
        assert not bad, (b, bad[:8])
        assert np.array_equal(costs[b].cpu().numpy(), oracle.tour_costs(dist[b].numpy(), rp))
    t1 = tau.to(dev()).clone().contiguous()
    t2 = t1.clone()
    engine.pheromone_update_(t1, paths, costs, 0.9, nbr=nbr)       # the table written by the sampler's epilogue
    engine.pheromone_update_(t2, paths, costs, 0.9)                # rebuilt from the paths
    assert torch.equal(t1, t2)


def test_infeasible_row_sets_flag():
    from deepaco_amd import engine
    n, A = 10, 4
    tau = torch.ones(n, n)
    eta = torch.zeros(n, n)
    for mode in ("scan", "race"):
        _, _, _, flags = engine.tsp_sample(tau.to(dev()), eta.to(dev()), A, mode=mode, seed=1)
        assert int(flags[0]) == 1


# ------------------------------------------------------------------ I1: the scan draw IS the reference's roulette
def test_scan_kernels_reproduce_reference_roulette_routes(mode="scan_wave"):
    """g6: routes the reference's `_inference_sample` (tsp_nls/aco.py:260-275) built from an injected uniform
    stream.  The same uniforms fed to the HIP scan kernel with one ant per wavefront give the same routes (n = 30: one
    chunk, the lanes walk the candidates in index order like the reference; the packed layouts -- sixteen candidates per
    chunk at this size since round 3 -- are pinned on the relabelled instances of g6w below, n = 30 included)."""
    from deepaco_amd import engine
    g = load_golden("g6_roulette_n30")
    n, A = g["probmat"].shape[0], g["uniforms"].shape[0]
    u = torch.from_numpy(g["uniforms"].astype(np.float32).T.copy()).to(dev())[None]       # [1][n-1][A]
    paths, _, _, flags = engine.tsp_sample(T(g["probmat"])[None], torch.ones(1, n, n, device=dev()), A, mode=mode,
                                           fixed_start=0, noise=u)
    assert int(flags.sum()) == 0
    assert np.array_equal(paths[0].cpu().numpy().T.astype(np.uint16), g["routes"])
    rp, _, _ = oracle.tsp_sample_scan_injected(g["probmat"], u[0].cpu().numpy(), fixed_start=0, wave=(mode == "scan_wave"))
    assert np.array_equal(paths[0].cpu().numpy(), rp)


@pytest.mark.parametrize("n", [30, 100, 200, 300, 600])
def test_scan_kernels_reproduce_reference_roulette_routes_lane_order(n):
    """g6w: the reference's roulette on the instance relabelled by a layout's lane order (tests/golden/gen_g6_wide.py), rows
    with exact zeros and a k-sparse row: the HIP scan kernels (sixteen / eight / two ants per wavefront; one ant per wavefront) fed the
    recorded uniforms build exactly the reference's routes."""
    from deepaco_amd import engine
    g = load_golden(f"g6w_roulette_n{n}")
    P = T(g["probmat"])[None]
    for lanes, mode in (((4 if n <= 128 else 8 if n <= 256 else 32), "scan"), (64, "scan_wave")):
        u = torch.from_numpy(g[f"uniforms_l{lanes}"].T.copy()).to(dev())[None]          # [1][n-1][A]
        A = u.shape[2]
        paths, _, _, flags = engine.tsp_sample(P, torch.ones(1, n, n, device=dev()), A, mode=mode, fixed_start=0, noise=u)
        assert int(flags.sum()) == 0
        assert np.array_equal(paths[0].cpu().numpy().T.astype(np.uint16), g[f"routes_l{lanes}"]), (n, mode)


def _layout_order(n, lanes):
    """position of candidate k in the order the lanes of a layout walk a row: (lane, chunk, slot)."""
    vec = 4 if lanes < 64 else (4 if n > 128 else (2 if n > 64 else 1))
    k = np.arange(n)
    key = ((k // vec) % lanes) * 1_000_000 + (k // (lanes * vec)) * 100 + (k % vec)
    return np.argsort(key, kind="stable")


@pytest.mark.parametrize("n,lanes", [(100, 4), (128, 4), (200, 8), (256, 8), (500, 32), (640, 32), (1100, 64)])
def test_scan_draw_equals_reference_roulette_arithmetic(n, lanes):
    """The benchmarked sampler against the LITERAL arithmetic of the reference's roulette (tsp_nls/aco.py:266-274:
    r = U * sum(prob_row * mask) in f64, subtract prob[k] one after the other until r <= 0), step by step along
    the tours the kernel built from an injected uniform stream.  A categorical draw does not depend on the order
    in which the candidates are walked; the kernels walk them lane by lane (coalesced 16-byte loads), so the
    reference's loop is run over the same order.  The two can only differ where U*S falls within rounding of a
    boundary between two neighbouring candidates (f32 tree sums vs f64 subtraction): every mismatch must be such
    a boundary case, and they must be rare."""
    from deepaco_amd import engine
    A = 8
    g = torch.Generator().manual_seed(n)
    tau = torch.rand(1, n, n, generator=g) + 0.1
    eta = torch.rand(1, n, n, generator=g) ** 2 + 1e-10
    u = torch.rand(1, n - 1, A, generator=g).clamp_(1e-7, 1 - 1e-7)
    paths, _, _, flags = engine.tsp_sample(tau.to(dev()), eta.to(dev()), A, mode="scan", fixed_start=0, noise=u.to(dev()))
    assert int(flags.sum()) == 0
    assert {4: n <= 128, 8: 128 < n <= 256, 32: 256 < n <= 1024, 64: n > 1024}[lanes]
    P = oracle.prob_matrix(tau[0].numpy(), eta[0].numpy())
    order = _layout_order(n, lanes)
    p = paths[0].cpu().numpy()
    un = u[0].numpy()
    mismatches = 0
    for a in range(A):
        mask = np.ones(n, dtype=np.float32)
        mask[p[0, a]] = 0
        for t in range(1, n):
            prob = (P[p[t - 1, a]] * mask)[order]                      # f32, the reference's prob_row * mask
            rnd = np.float64(un[t - 1, a]) * np.float64(prob.sum(dtype=np.float32))
            cum = np.cumsum(prob.astype(np.float64))
            j = int(np.searchsorted(cum, rnd, side="left"))           # first j with rnd - cum[j] <= 0
            j = min(j, n - 1)
            k_ref, k_gpu = int(order[j]), int(p[t, a])
            if k_ref != k_gpu:
                mismatches += 1
                jg = int(np.where(order == k_gpu)[0][0])
                lo, hi = min(j, jg), max(j, jg)
                assert prob[jg] > 0 and not (prob[lo + 1:hi] > 0).any(), "not a neighbouring open candidate"
                assert abs(cum[lo] - rnd) <= 4e-6 * cum[-1], "not a rounding-boundary draw"
            mask[k_gpu] = 0
    assert mismatches <= 2, mismatches                                   # (expected: 0 of ~4000 draws)


# ------------------------------------------------------------------ U3: sparsify (tsp/aco.py:52-67)
def test_sparsify_matches_reference():
    """ACO.sparsify(k) and the batched BatchedTSP.sparsify against the heuristic the reference built on the captured
    instance (fixture g1_tsp_n50_a16_sparse: `heuristic` is the reference's aco.heuristic after sparsify(k))."""
    from deepaco_amd import engine
    from deepaco_amd.tsp.aco import ACO
    g = load_golden("g1_tsp_n50_a16_sparse")
    k = int(g["sparsify_k"])
    aco = ACO(T(g["distances"]), n_ants=4, device="cuda:0")
    aco.sparsify(k)
    assert np.array_equal(aco.heuristic.cpu().numpy().view(np.uint32), g["heuristic"].view(np.uint32))
    n = g["distances"].shape[0]
    assert int((aco.heuristic.cpu() > 1e-9).sum()) == n * k            # k live edges per node, 1e-10 elsewhere
    col = engine.BatchedTSP(T(g["distances"])[None].repeat(3, 1, 1), n_ants=4)
    col.sparsify(k)
    for b in range(3):
        assert np.array_equal(col.heuristic[b].cpu().numpy().view(np.uint32), g["heuristic"].view(np.uint32))


# ------------------------------------------------------------------ 4. H3: inference schedule, statistical
def test_inference_schedule_matches_cpu_port_statistically():
    """tsp/test.ipynb infer_instance/test: incremental aco.run(t_diff) over t_aco = [1, 10, 20]; mean best
    cost over instances must agree with the reference's CPU op sequence (oracle/torch_port.py) within
    sampling noise, for both samplers (identical categorical distribution, different RNG streams)."""
    from deepaco_amd.tsp.aco import ACO
    from oracle import torch_port
    n, A, inst = 60, 20, 10
    t_aco = [1, 10, 20]
    diffs = [t_aco[0]] + [t_aco[i + 1] - t_aco[i] for i in range(len(t_aco) - 1)]
    dist, _, _ = make_instance(n, 777, inst)
    torch.manual_seed(12345)
    cpu = np.zeros(len(t_aco))
    for b in range(inst):
        d = dist[b]
        tau, low = torch.ones_like(d), float("inf")
        for i, td in enumerate(diffs):
            for _ in range(td):
                paths = torch_port.rollout(tau, 1 / d, A)
                costs = torch_port.tour_lengths(d, paths)
                low = min(low, float(costs.min()))
                tau = torch_port.deposit(tau, paths, costs, 0.9)
            cpu[i] += low / inst
    for sampler in ("scan", "race"):
        gpu = np.zeros(len(t_aco))
        for b in range(inst):
            aco = ACO(dist[b].to(dev()), n_ants=A, device="cuda:0", sampler=sampler, seed=1000 + b)
            for i, td in enumerate(diffs):
                gpu[i] += float(aco.run(td)) / inst
        assert np.all(np.diff(gpu) <= 1e-6)                      # best-so-far never gets worse
        np.testing.assert_allclose(gpu, cpu, rtol=0.05), (sampler, gpu, cpu)


def test_ant_sharded_colony_world1_matches_batched():
    """The ant-sharded colony (deposit on zeros + tau*decay + delta) with one rank draws the same tours as
    BatchedTSP and keeps the same pheromone up to summation order."""
    from deepaco_amd import engine
    B, n, A, iters = 3, 40, 16, 4
    dist, _, _ = make_instance(n, 9, B)
    col = engine.ant_sharded_tsp(dist.to(dev()), A, 0, 1, seed=5)
    ref = engine.BatchedTSP(dist.to(dev()), n_ants=A, seed=5)
    for _ in range(iters):
        p1, c1 = col.step()
        p2, c2 = ref.step()
        assert torch.equal(p1, p2) and torch.equal(c1, c2)
    torch.testing.assert_close(col.tau, ref.pheromone, rtol=2e-6, atol=0)
    torch.testing.assert_close(col.lowest_cost, ref.lowest_cost, rtol=0, atol=0)


# ------------------------------------------------------------------ 5. limits and odd shapes
def test_maximum_size_and_limit():
    from deepaco_amd import engine
    n, A = 4096, 2
    dist, tau, eta = make_instance(n, 4096)
    paths, _, _, flags = engine.tsp_sample(tau.to(dev()), eta.to(dev()), A, mode="scan", seed=3, it=0)
    assert int(flags.sum()) == 0
    rp, _, rc = oracle.tsp_sample_scan(oracle.prob_matrix(tau[0].numpy(), eta[0].numpy()), A, 3, 0, 0)
    assert rc == 0 and np.array_equal(paths[0].cpu().numpy(), rp)
    big = torch.ones(1, 4097, 4097, device=dev())
    with pytest.raises(ValueError):
        engine.tsp_sample(big, big, 2)


@pytest.mark.parametrize("B,A", [(1, 1), (3, 5), (9, 3), (17, 6), (5, 130)])
def test_odd_batch_and_ant_counts(B, A):
    """Workgroup remap (XCD-aware, bijective for any grid) and partial last workgroups (A % 4 != 0)."""
    from deepaco_amd import engine
    n = 90
    dist, tau, eta = make_instance(n, B * 100 + A, B)
    paths, _, _, flags, costs, nbr = engine.tsp_sample(tau.to(dev()), eta.to(dev()), A, mode="scan", seed=8, it=1,
                                                       dist=dist.to(dev()), want_nbr=True)
    assert int(flags.sum()) == 0
    for b in range(B):
        rp, _, _ = oracle.tsp_sample_scan(oracle.prob_matrix(tau[b].numpy(), eta[b].numpy()), A, 8, 1, b * A)
        assert np.array_equal(paths[b].cpu().numpy(), rp), (B, A, b)
        assert np.array_equal(costs[b].cpu().numpy(), oracle.tour_costs(dist[b].numpy(), rp))
    # the fused neighbour table equals the one rebuilt from the paths (same deposit either way)
    t1 = tau.to(dev()).clone().contiguous()
    t2 = t1.clone()
    engine.pheromone_update_(t1, paths, costs, 0.9, nbr=nbr)
    engine.pheromone_update_(t2, paths, costs, 0.9)
    assert torch.equal(t1, t2)


@pytest.mark.parametrize("kw", [{}, {"elitist": True}, {"min_max": True}])
@pytest.mark.parametrize("n,A,B", [(60, 20, 2), (150, 9, 1)])
def test_graph_replay_equals_eager_loop(kw, n, A, B):
    """BatchedTSP.run(graph=True): the iteration captured into a HIP graph (device-side Philox iteration counter)
    gives the same colony, bit for bit, as the eager loop."""
    from deepaco_amd import engine
    dist, _, _ = make_instance(n, 17 + n, B)
    c1 = engine.BatchedTSP(dist.to(dev()), n_ants=A, seed=4, **kw)
    c2 = engine.BatchedTSP(dist.to(dev()), n_ants=A, seed=4, **kw)
    r1 = c1.run(7)
    r2 = c2.run(7, graph=True)
    torch.cuda.synchronize()
    assert c1.iteration == c2.iteration == 7
    assert torch.equal(c1.pheromone, c2.pheromone)
    assert torch.equal(r1, r2) and torch.equal(c1.shortest_path, c2.shortest_path)
    r1, r2 = c1.run(2), c2.run(4, graph=True)      # the colonies stay usable either way
    assert bool((r2 <= r1).all())


@pytest.mark.parametrize("kw", [{}, {"elitist": True}, {"min_max": True}])
def test_sync_free_run_equals_plain_sequence(kw):
    """ACO.run's device-side bookkeeping (no host sync, fused costs / neighbour table) gives the same colony as the
    reference's call sequence gen_path -> gen_path_costs -> update_pheronome."""
    from deepaco_amd.tsp.aco import ACO
    n, A = 80, 24
    dist, _, _ = make_instance(n, 31)
    a1 = ACO(dist[0].to(dev()), n_ants=A, device="cuda:0", seed=12, **kw)
    a2 = ACO(dist[0].to(dev()), n_ants=A, device="cuda:0", seed=12, **kw)
    r1 = a1.run(6)
    r2 = a2._run_plain(6)
    assert torch.equal(a1.pheromone, a2.pheromone)
    assert float(r1) == float(r2) and torch.equal(a1.shortest_path, a2.shortest_path)
    r1b = a1.run(3)                                     # continues from tensor state
    assert float(r1b) <= float(r1)


@pytest.mark.parametrize("kw", [{}, {"elitist": True}, {"min_max": True}, {"min_max": True, "min": 0.05}])
@pytest.mark.parametrize("n,A,k,learned", [(300, 24, 30, False), (500, 50, 50, True), (700, 20, 70, False)])
def test_run_on_head_rows_is_a_one_instance_colony_with_the_plain_loops_results(n, A, k, learned, kw):
    """ACO.run on head / tail rows keeps a one-instance engine.BatchedTSP (round 6: the update's head rows, the LDS-heads variant for
    the reference's own ant counts, compact tours).  Against the plain call sequence of tsp/aco.py:75-92 on the same object state
    (gen_path -> gen_path_costs -> best-so-far -> update_pheronome, `_run_plain`): the same record, best tour and pheromone after
    two run() calls (the second continues the first: iteration counter, MMAS bound), AS / elitist / MMAS; `learned`: a k-sparse
    heuristic nobody announced (sampler auto picks the head)."""
    from deepaco_amd.tsp.aco import ACO
    g = torch.Generator().manual_seed(n + A)
    c = torch.rand(n, 2, generator=g)
    d = (c[:, None] - c).norm(dim=-1)
    d[torch.arange(n), torch.arange(n)] = 1e9
    d = d.to(dev())
    heu = None
    if learned:
        _, idx = torch.topk(d, k=k, dim=1, largest=False)
        heu = torch.full_like(d, 1e-10).scatter_(1, idx, torch.rand(n, k, device=dev()) + 0.05)
    objs = []
    for _ in range(2):
        a = ACO(d, n_ants=A, heuristic=heu, device="cuda:0", seed=77, **kw)
        if not learned:
            a.sparsify(k)
        objs.append(a)
    fast, plain = objs
    assert fast.resolved_sampler()[0] == "scan_sparse"
    held = fast.pheromone
    before = held.clone()
    for iters in (4, 3):
        fast.run(iters)
        plain._run_plain(iters)
        assert float(fast.lowest_cost) == float(plain.lowest_cost)
        assert torch.equal(fast.shortest_path, plain.shortest_path)
        assert torch.equal(fast.pheromone, plain.pheromone.to(torch.float32))
    assert torch.equal(held, before)                        # a tensor the caller still holds is never modified
    assert fast._calls == plain._calls == 7
    fast.check_feasible()
