"""GPU tests of the route-exact CVRP local search (daco_hgs_prepare / daco_hgs_local_search, csrc/daco_hgs_ls.hip).

The reference: cvrp_nls/aco.py:114-126 -> swapstar.py:324-346 -> HGS-CVRP-main/Program/LocalSearch.cpp.  Parity is ROUTE FOR
ROUTE: (a) fixtures g11 = the reference's own outputs (its Python over HGS built from its sources) for every loop bound,
both matrices and the three-stage neural_swapstar; (b) the oracle (oracle/hgs_ls.c, itself pinned on the reference's
library) on fresh random solutions at sizes up to BASELINE's configuration 4 (CVRP-100, 512 ants)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

import oracle

pytestmark = pytest.mark.gpu
FILES = sorted(glob.glob(os.path.join(GOLDEN, "g11_hgs_ls_n*.npz")))


def dev():
    return torch.device("cuda:0")


@pytest.fixture(params=["0", "1"], ids=["throughput", "latency"], autouse=True)
def hgs_mode(request, monkeypatch):
    """Every test of this module runs on both forms of the kernel: the persistent wavefronts of the batched colonies
    (DACO_HGS_LATENCY=0) and the latency mode -- one wavefront per workgroup with the stage's matrix in LDS, what a call with a
    handful of solutions takes by itself -- forced wherever the matrix fits (=1; larger instances fall back)."""
    monkeypatch.setenv("DACO_HGS_LATENCY", request.param)
    return request.param


def run(paths_in, stages, demands, Lpad=2, want_stats=False):
    from deepaco_amd import engine
    pin = torch.as_tensor(np.asarray(paths_in, dtype=np.int64))
    if pin.dim() == 2:
        pin = pin[None]
    B, L, A = pin.shape
    p = torch.zeros((B, L + Lpad, A), dtype=torch.int64)
    p[:, :L] = pin
    p = p.to(dev()).contiguous()
    out = engine.hgs_local_search_(p, stages, torch.as_tensor(demands).to(dev()), want_stats=want_stats)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_fixtures_route_for_route(path):
    from deepaco_amd import engine
    z = np.load(path)
    td = engine.HgsTables(torch.as_tensor(z["distances"]).to(dev()))
    th = engine.HgsTables(torch.as_tensor(z["heuristic_dist"]).to(dev()))
    for c in (0, 1, 2, 100):
        out = run(z["paths_in"], [(td, c)], z["demands"])
        np.testing.assert_array_equal(out[0].cpu().numpy(), z[f"paths_as_run_c{c}"].astype(np.int64), err_msg=f"count {c}")
    out = run(z["paths_in"], [(th, 10)], z["demands"])
    np.testing.assert_array_equal(out[0].cpu().numpy(), z["paths_as_run_hd_c10"].astype(np.int64))
    lim = int(z["limit"])
    out, status, stats = run(z["paths_in"], [(td, lim), (th, 10), (td, lim)], z["demands"], want_stats=True)
    np.testing.assert_array_equal(out[0].cpu().numpy(), z["paths_as_run_nls"].astype(np.int64))
    assert int(status.abs().sum()) == 0 and int(stats[..., 0].min()) > 0


def test_tables_equal_the_oracles_correlated_vertices():
    from deepaco_amd import engine
    z = np.load(FILES[2])
    for key in ("distances", "heuristic_dist"):
        m = z[key]
        n = m.shape[0]
        t = engine.HgsTables(torch.as_tensor(m).to(dev()))
        torch.cuda.synchronize()
        raw = t.tables.cpu().numpy()
        lists, lens = oracle.hgs_correlated(m, 20)
        a16 = lambda x: (x + 15) & ~15
        o_order = 64
        o_len = o_order + a16(2 * (n - 1))
        o_off = o_len + a16(2 * n)
        o_ent = o_off + a16(4 * n)
        assert raw[:8].view(np.float64)[0] == m.max()
        glen = raw[o_len:o_len + 2 * n].view(np.uint16)
        goff = raw[o_off:o_off + 4 * n].view(np.uint32)
        np.testing.assert_array_equal(glen, lens.astype(np.uint16))
        for i in range(1, n):
            e = raw[o_ent + 2 * goff[i]: o_ent + 2 * (goff[i] + glen[i])].view(np.uint16)
            np.testing.assert_array_equal(e, lists[i, :lens[i]].astype(np.uint16))
        order = raw[o_order:o_order + 2 * (n - 1)].view(np.uint16)
        nc = n - 1
        _, d1 = oracle.hgs_shuffle(np.arange(nc), seed=1)                          # the draws of Individual's own shuffle
        want, _ = oracle.hgs_shuffle(np.arange(1, nc + 1), seed=1, skip_draws=d1)
        np.testing.assert_array_equal(order, want.astype(np.uint16))


def random_solutions(rng, n, cap, A):
    pos = rng.random((n + 1, 2))
    d = np.linalg.norm(pos[:, None] - pos[None], axis=-1)
    d[np.arange(n + 1), np.arange(n + 1)] = 1e-10
    dem = np.concatenate(([0.0], rng.integers(1, 10, n) / cap))
    cols = []
    for _ in range(A):
        seq, load = [0], 0.0
        for c in rng.permutation(np.arange(1, n + 1)):
            if load + dem[c] > 1.0:
                seq.append(0); load = 0.0
            seq.append(int(c)); load += dem[c]
        cols.append(seq + [0])
    L = max(map(len, cols))
    paths = np.zeros((L, A), dtype=np.int64)
    for a, s in enumerate(cols):
        paths[:len(s), a] = s
    return pos, d, dem, paths


@pytest.mark.parametrize("n,cap,A,count", [(12, 20, 64, 100), (33, 30, 64, 3), (100, 50, 96, 100), (150, 50, 24, 100), (500, 150, 6, 100),
                                           (1000, 200, 3, 4), (2000, 300, 2, 1)])
def test_random_solutions_against_the_oracle(n, cap, A, count):
    """Random (far from optimal) solutions: hundreds of moves each, empty routes appear and are used."""
    from deepaco_amd import engine
    rng = np.random.default_rng(n)
    pos, d, dem, paths = random_solutions(rng, n, cap, A)
    hd = 1 / ((1 / d) / (1 / d).max(-1, keepdims=True) * (0.3 + rng.random(d.shape)) + 1e-5)
    td = engine.HgsTables(torch.as_tensor(d).to(dev()))
    th = engine.HgsTables(torch.as_tensor(hd).to(dev()))
    L = paths.shape[0] + 2
    out, status, stats = run(paths, [(td, count)], dem, want_stats=True)
    got = out[0].cpu().numpy()
    moves = 0
    for a in range(A):
        want, rc, st = oracle.hgs_local_search(pos, d, dem, paths[:, a], count, out_len=L, want_stats=True)
        np.testing.assert_array_equal(got[:, a], want, err_msg=f"ant {a}")
        assert int(stats[0, a, 0]) == st[0] and int(stats[0, a, 1]) == st[1]
        moves += st[0]
    assert moves > 5 * A
    if n > 500:                      # (cvrp_nls/utils.py:5 lists sizes up to 2000: the LDS plan and the 16-bit node ids hold there)
        return
    out = run(paths, [(td, count), (th, 10), (td, count)], dem)
    got = out[0].cpu().numpy()
    for a in range(0, A, 3):
        want = oracle.hgs_neural_swapstar(pos, d, hd, dem, paths[:, a], count)
        np.testing.assert_array_equal(got[:, a], want, err_msg=f"nls ant {a}")


def test_batch_of_instances_and_config4_shape():
    """B instances side by side at BASELINE configuration 4's colony size (CVRP-100, 512 ants): instances 0 and B-1 ant for
    ant against the oracle (sampled ants), every column a complete solution."""
    from deepaco_amd import engine
    B, n, A = 6, 100, 512
    rng = np.random.default_rng(7)
    inst = [random_solutions(rng, n, 50, A) for _ in range(B)]
    L = max(i[3].shape[0] for i in inst) + 2
    paths = np.zeros((B, L, A), dtype=np.int64)
    for b, i in enumerate(inst):
        paths[b, :i[3].shape[0]] = i[3]
    d = torch.as_tensor(np.stack([i[1] for i in inst])).to(dev())
    dem = torch.as_tensor(np.stack([i[2] for i in inst])).to(dev())
    td = engine.HgsTables(d)
    p = torch.as_tensor(paths).to(dev()).contiguous()
    out, status, stats = engine.hgs_local_search_(p, [(td, 100)], dem, want_stats=True)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert int(status.abs().sum()) == 0
    for b in (0, B - 1):
        pos, dd, de, pin = inst[b]
        for a in list(range(0, A, 37)) + [A - 1]:
            want, _ = oracle.hgs_local_search(pos, dd, de, pin[:, a], 100, out_len=L)
            np.testing.assert_array_equal(got[b, :, a], want, err_msg=f"instance {b} ant {a}")
    srt = np.sort(got, axis=1)
    assert (srt[:, -n:, :] == np.arange(1, n + 1)[None, :, None]).all()


def test_stage_that_hgs_refuses_keeps_its_input():
    """A matrix beyond HGS's scale check (Params.cpp:106-108) throws there; the reference keeps the routes (swapstar.py:341-345)."""
    from deepaco_amd import engine
    rng = np.random.default_rng(3)
    pos, d, dem, paths = random_solutions(rng, 30, 30, 16)
    big = engine.HgsTables(torch.as_tensor(d * 1e6).to(dev()))
    td = engine.HgsTables(torch.as_tensor(d).to(dev()))
    out, status, _ = run(paths, [(big, 10)], dem, want_stats=True)
    assert int(status.min()) == 1
    L = paths.shape[0] + 2
    for a in range(16):
        want, rc = oracle.hgs_local_search(pos, d * 1e6, dem, paths[:, a], 10, out_len=L)
        assert rc == 1
        np.testing.assert_array_equal(out[0, :, a].cpu().numpy(), want)
    out, status, _ = run(paths, [(td, 5), (big, 10), (td, 5)], dem, want_stats=True)
    for a in range(16):
        s1, _ = oracle.hgs_local_search(pos, d, dem, paths[:, a], 5, out_len=L)
        s3, _ = oracle.hgs_local_search(pos, d, dem, s1, 5, out_len=L)
        np.testing.assert_array_equal(out[0, :, a].cpu().numpy(), s3)


def test_incomplete_column_is_left_untouched():
    from deepaco_amd import engine
    rng = np.random.default_rng(4)
    pos, d, dem, paths = random_solutions(rng, 20, 30, 8)
    bad = paths.copy()
    bad[np.flatnonzero(bad[:, 3])[0], 3] = 0                       # ant 3 loses a client
    td = engine.HgsTables(torch.as_tensor(d).to(dev()))
    out, status, _ = run(bad, [(td, 10)], dem, want_stats=True)
    assert status[0].tolist() == [0, 0, 0, 2, 0, 0, 0, 0]
    np.testing.assert_array_equal(out[0, :bad.shape[0], 3].cpu().numpy(), bad[:, 3])


def test_batched_cvrp_colony_with_the_local_search():
    """engine.BatchedCVRP(local_search="hgs"): the iteration of cvrp_nls/aco.py:134-171 for B instances -- the eight cheapest ants
    of every instance go through neural_swapstar before best tracking and the deposit.  First iteration against a twin colony
    without the local search (same seed: the same sampled routes) and the oracle on the selected ants; best costs improve."""
    from deepaco_amd import engine
    B, n, A = 3, 60, 24
    g = torch.Generator().manual_seed(9)
    loc = torch.cat((torch.full((B, 1, 2), 0.5, dtype=torch.double), torch.rand(B, n, 2, generator=g, dtype=torch.double)), 1)
    dem = torch.cat((torch.zeros(B, 1, dtype=torch.double), torch.randint(1, 10, (B, n), generator=g).double()), 1)
    d = (loc[:, :, None] - loc[:, None]).norm(dim=-1)
    d[:, torch.arange(n + 1), torch.arange(n + 1)] = 1e-10
    plain = engine.BatchedCVRP(d.to(dev()), dem.to(dev()), n_ants=A, capacity=30, seed=4)
    ls = engine.BatchedCVRP(d.to(dev()), dem.to(dev()), n_ants=A, capacity=30, seed=4, local_search="hgs")
    p0, c0 = plain.step()
    p1, c1 = ls.step()
    idx = c0.topk(8, dim=1, largest=False).indices
    hd = (1 / ((1 / d) / (1 / d).max(-1, keepdim=True).values + 1e-5)).numpy()
    L = p0.shape[1]
    touched = torch.zeros(B, A, dtype=torch.bool, device=dev())
    touched.scatter_(1, idx, True)
    assert torch.equal(p1.permute(0, 2, 1)[~touched], p0.permute(0, 2, 1)[~touched])          # the other ants keep their routes
    for b in range(B):
        for a in idx[b].tolist()[:3]:
            want = oracle.hgs_neural_swapstar(loc[b].numpy(), d[b].numpy(), hd[b], (dem[b] / 30).numpy(), p0[b, :, a].cpu().numpy(),
                                              max(n + 1, 50))
            np.testing.assert_array_equal(p1[b, :, a].cpu().numpy(), want[:L], err_msg=f"instance {b} ant {a}")
            assert not want[L:].any()
    assert bool((c1 <= c0 + 1e-5).all()) and bool((ls.lowest_cost < plain.lowest_cost).all())
    ls.run(3)
    assert bool((ls.lowest_cost <= c1.min(dim=1).values + 1e-6).all())
