"""One launch per BASELINE.json configuration at its REAL batch size, instances {0, B/2, B-1} ant for ant against the oracle
(the workgroup -> instance mapping and the XCD remap depend on the grid size; B <= 8 elsewhere).

  config 2   TSP-100,  512 ants, 256 instances           scan16_kernel
  headline   TSP-500,  512 ants,  64 instances           tsp_scan32_kernel
  config 3   TSP-500 + NLS, 256 ants, 64 instances       nls_kernel (a slice of the ants through the oracle's schedule)
  config 4   CVRP-100, 512 ants, 256 instances           scan16_kernel<CVRP> + the route-exact local search on top
  config 5   TSP-1000, 2048 ants, 64 instances per GPU   tsp_scan32_kernel
  headline / config 5 on the head rows (sampler "auto" after sparsify(k): what bench.py times)   scan_sparse_kernel, EVERY ant
The whole colony iteration (costs, deposit in ant order) is compared for the checked instances, bit for bit."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def tsp_instances(B, n, seed):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(B, n, 2, generator=g)
    d = (c[:, :, None] - c[:, None]).norm(dim=-1)        # (torch.cdist's matmul form returns exact zeros for close pairs: 1/d = inf)
    i = torch.arange(n)
    d[:, i, i] = 1e9
    return d


def sparse_heuristic(d, k):
    _, idx = torch.topk(d, k=k, dim=2, largest=False)
    return 1 / torch.full_like(d, 1e10).scatter_(2, idx, torch.gather(d, 2, idx))


@pytest.mark.parametrize("n,A,B,k", [(100, 512, 256, 20), (500, 512, 64, 50), (1000, 2048, 64, 100)])
def test_tsp_colony_iteration_at_the_configurations_batch(n, A, B, k):
    from deepaco_amd import engine
    d = tsp_instances(B, n, 31 * n)
    eta = sparse_heuristic(d, k)
    g = torch.Generator().manual_seed(n)
    tau = torch.rand(B, n, n, generator=g) * 0.5 + 0.75
    D, T, E = d.to(dev()), tau.to(dev()).contiguous(), eta.to(dev()).contiguous()
    paths, _, _, flags, costs, nbr = engine.tsp_sample(T, E, A, mode="scan", seed=5, it=2, batch=B, dist=D, want_nbr=True)
    assert int(flags.sum()) == 0
    t2 = T.clone()
    engine.pheromone_update_(t2, paths, costs, 0.9, nbr=nbr)
    for b in (0, B // 2, B - 1):
        rp, _, rc = oracle.tsp_sample_scan(oracle.prob_matrix(tau[b].numpy(), eta[b].numpy()), A, 5, 2, b * A)
        assert rc == 0
        got = paths[b].cpu().numpy()
        bad = [a for a in range(A) if not np.array_equal(rp[:, a], got[:, a])]
        assert not bad, (b, bad[:8])
        rc_ = oracle.tour_costs(d[b].numpy(), rp)
        assert np.array_equal(costs[b].cpu().numpy(), rc_)
        rt = oracle.pheromone_update_tsp(tau[b].numpy(), rp, rc_, 0.9)
        assert np.array_equal(t2[b].cpu().numpy().view(np.uint32), rt.view(np.uint32)), b
    # every tour of every instance is a permutation
    assert bool((paths.sort(dim=1).values == torch.arange(n, device=dev()).view(1, n, 1)).all())


@pytest.mark.parametrize("n,A,B,k", [(500, 512, 64, 50), (1000, 2048, 64, 100)])
def test_head_row_sampler_at_the_bench_shape_every_ant(n, A, B, k):
    """The kernel the headline times: BatchedTSP(sampler="auto") after sparsify(k) resolves to the head / tail
    rows (scan_sparse_kernel<2,false,4> at TSP-500 x 512 x 64, k = 50; <4,false,8> at config 5's per-GPU share) -- one colony
    iteration at the bench's grid, EVERY ant of EVERY instance against oracle.tsp_sample_scan_sparse (the restatement the CPU
    suite holds against the categorical of tsp/aco.py:165-177), the three step counters, the fused tour lengths and the deposit
    in ant order (tsp/aco.py:95-118) bit for bit."""
    from deepaco_amd import engine
    d = tsp_instances(B, n, 31 * n + 1)
    g = torch.Generator().manual_seed(n + 7)
    tau = torch.rand(B, n, n, generator=g) * 0.5 + 0.75
    D = d.to(dev())
    col = engine.BatchedTSP(D, n_ants=A, seed=5, sampler="auto", pheromone=tau.to(dev()))
    col.sparsify(k)
    col.heuristic = col.heuristic.contiguous()
    assert col.resolved_sampler() == ("scan_sparse", k)
    head = col._head_table(k)
    assert head.shape[2] == (64 if k <= 63 else 128)
    # the launch by itself (step counters), then the colony's step: the same tours, then costs / bookkeeping / deposit
    paths, flags, costs, nbr, stats = engine.tsp_sample_sparse(col.pheromone, col.heuristic, A, head, seed=5, it=0, batch=B, dist=D,
                                                               want_nbr=True, want_stats=True)
    assert int(flags.sum()) == 0
    p2, c2 = col.step()
    assert torch.equal(p2, paths) and torch.equal(c2.view(torch.int32), costs.view(torch.int32))
    eta = col.heuristic.cpu().numpy()
    hid = head.cpu().numpy().view(np.uint16)
    got_p, got_c, got_t = paths.cpu().numpy(), costs.cpu().numpy(), col.pheromone.cpu().numpy()
    low = col.lowest_cost.cpu().numpy()

    def check(b):
        ids = hid[b].copy()
        cnt = ids[:, -1].astype(np.uint8)
        ids[:, -1] = 0
        ref, rc, st = oracle.tsp_sample_scan_sparse(oracle.prob_matrix(tau[b].numpy(), eta[b]), ids, cnt, A, seed=5, it=0, ant_gid0=b * A)
        bad = int((got_p[b] != ref).any(axis=0).sum()) + (rc != 0)
        rc_ = oracle.tour_costs(d[b].numpy(), ref)
        bad_c = int((got_c[b].view(np.uint32) != rc_.view(np.uint32)).sum())
        rt = oracle.pheromone_update_tsp(tau[b].numpy(), ref, rc_, 0.9)
        bad_t = int((got_t[b].view(np.uint32) != rt.view(np.uint32)).sum())
        return bad, bad_c, bad_t, st, float(rc_.min()) == float(low[b])

    with ThreadPoolExecutor(max_workers=max(1, min(16, os.cpu_count() or 1))) as ex:     # (the C oracle releases the GIL)
        res = list(ex.map(check, range(B)))
    assert sum(r[0] for r in res) == 0, ("ants that differ", [(b, r[0]) for b, r in enumerate(res) if r[0]][:8])
    assert sum(r[1] for r in res) == 0 and sum(r[2] for r in res) == 0, ("costs / pheromone entries that differ", [(b, r[1], r[2]) for b, r in enumerate(res) if r[1] or r[2]][:8])
    assert all(r[4] for r in res)
    ref_stats = np.sum([r[3] for r in res], axis=0)
    assert np.array_equal(stats.cpu().numpy(), ref_stats), (stats.cpu().numpy(), ref_stats)
    assert ref_stats[1] == 0 and ref_stats[2] == 0 and 0 < ref_stats[0] < 0.05 * B * A * n        # k-sparse: never past the head, few dense steps


def test_config3_nls_at_its_batch():
    """TSP-500 + NLS, 256 ants x 64 instances in one launch of the fused search; ants {0, 100, 255} of instances
    {0, 32, 63} through the oracle's schedule (tsp_nls/aco.py:241-258 over two_opt.py:6-39): tours and f32 lengths."""
    from deepaco_amd import engine
    n, A, B, maxt = 500, 256, 64, 125
    d = tsp_instances(B, n, 77)
    eta = sparse_heuristic(d, 50)
    D, E = d.to(dev()), eta.to(dev()).contiguous()
    paths, _, _, _ = engine.tsp_sample(torch.ones_like(D), E, A, mode="scan", seed=9, fixed_start=0, batch=B)
    tours = paths.permute(0, 2, 1).to(torch.int16).contiguous()
    hd = (1 / (E / E.amax(dim=-1, keepdim=True) + 1e-5)).contiguous()
    out, costs = engine.nls_(D, hd, tours, maxt, fused=True, want_costs=True)
    ants = [0, 100, 255]
    for b in (0, B // 2, B - 1):
        t_in = tours[b, ants].cpu().numpy().astype(np.uint16)
        ref, _ = oracle.nls_batch(d[b].numpy(), hd[b].cpu().numpy(), t_in, maxt)
        assert np.array_equal(out[b, ants].cpu().numpy().astype(np.uint16), ref), b
        rc = oracle.tour_costs(d[b].numpy(), np.ascontiguousarray(ref.T.astype(np.int64)), closed=True)
        assert np.array_equal(costs[b, ants].cpu().numpy().view(np.int32), np.asarray(rc, dtype=np.float32).view(np.int32))
    assert bool((out.to(torch.int64).sort(dim=2).values == torch.arange(n, device=dev()).view(1, 1, n)).all())


def test_config4_cvrp_at_its_batch_with_the_local_search_on_top():
    """CVRP-100 (capacity 50, demands 1..9), 512 ants x 256 instances: construction + costs + directed deposit against the
    oracle for instances {0, 128, 255}; then the route-exact local search on ALL 131 072 solutions in one launch, ants
    {0, 255, 511} of the same instances against oracle/hgs_ls.c (float64 instance data as cvrp_nls/utils.py builds it)."""
    from deepaco_amd import engine
    n, A, B, cap = 100, 512, 256, 50.0
    g = torch.Generator().manual_seed(4)
    loc = torch.cat((torch.full((B, 1, 2), 0.5, dtype=torch.double), torch.rand(B, n, 2, generator=g, dtype=torch.double)), 1)
    dem_i = torch.cat((torch.zeros(B, 1), torch.randint(1, 10, (B, n), generator=g).float()), 1)
    d64 = (loc[:, :, None] - loc[:, None]).norm(dim=-1)
    ii = torch.arange(n + 1)
    d64[:, ii, ii] = 1e-10
    d = d64.float()
    tau = torch.rand(B, n + 1, n + 1, generator=g) * 0.5 + 0.75
    eta = 1 / d
    D, T, E, DM = d.to(dev()), tau.to(dev()).contiguous(), eta.to(dev()).contiguous(), dem_i.to(dev())
    paths, _, _, lens, flags, costs, table = engine.cvrp_sample(T, E, DM, cap, A, seed=3, it=1, batch=B, dist=D, want_table=True)
    assert int(flags.sum()) == 0
    for b in (0, B // 2, B - 1):
        P = oracle.prob_matrix(tau[b].numpy(), eta[b].numpy())
        rp, _, Lb = oracle.cvrp_sample_rng(P, dem_i[b].numpy(), cap, A, "scan", 3, 1, b * A)
        assert Lb == int(lens[b].max())
        assert np.array_equal(paths[b, :Lb].cpu().numpy(), rp), b
        assert not paths[b, Lb:].any()
        rc = oracle.tour_costs(d[b].numpy(), rp, closed=False)
        assert np.array_equal(costs[b].cpu().numpy(), rc)
    # the local search of cvrp_nls on top (normalised float64 demands, capacity 1.0 -> HGS's 1000 / 1000.001)
    L = int(lens.max())
    work = paths[:, :L].contiguous()
    before = work.clone()
    td = engine.HgsTables(d64.to(dev()))
    dem64 = (dem_i.double() / cap)
    _, status, stats = engine.hgs_local_search_(work, [(td, 101)], dem64.to(dev()), want_stats=True)
    torch.cuda.synchronize()
    assert int(status.abs().sum()) == 0 and int(stats[..., 3].abs().sum()) == 0
    for b in (0, B // 2, B - 1):
        for a in (0, 255, 511):
            want, rc = oracle.hgs_local_search(loc[b].numpy(), d64[b].numpy(), dem64[b].numpy(), before[b, :, a].cpu().numpy(), 101,
                                               out_len=L)
            assert rc == 0
            np.testing.assert_array_equal(work[b, :, a].cpu().numpy(), want, err_msg=f"instance {b} ant {a}")
    srt = work.sort(dim=1).values
    assert bool((srt[:, -n:, :] == torch.arange(1, n + 1, device=dev()).view(1, n, 1)).all())
