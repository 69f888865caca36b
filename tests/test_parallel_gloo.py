"""world_size-2 gloo tests (CPU) of the multi-GPU host logic in deepaco_amd/parallel.py.

The kernels are replaced by the CPU oracle here (tests may use it as the checker); what is
under test is sharding, the barrier/max timing helper, the final gather of the instance-sharded
mode and both exchanges of the ant-sharded mode (delta-tau all-reduce; all-gather of the tours, exact)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from deepaco_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _instances(B, n, seed):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(B, n, 2, generator=g)
    d = torch.cdist(c, c)
    i = torch.arange(n)
    d[:, i, i] = 1e9
    return d


def _oracle_colony(d, A, iters, seed, gid0):
    n = d.shape[0]
    tau = np.ones((n, n), np.float32)
    eta = (1.0 / d).numpy()
    low = np.inf
    for it in range(iters):
        paths, _, _ = oracle.tsp_sample_scan(oracle.prob_matrix(tau, eta), A, seed, it, gid0)
        costs = oracle.tour_costs(d.numpy(), paths)
        low = min(low, float(costs.min()))
        tau = oracle.pheromone_update_tsp(tau, paths, costs, 0.9)
    return low, tau


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    out = {}
    # --- instance-sharded: each rank runs its slice, results gathered; equals the 1-process run
    B, n, A, iters, seed = 5, 16, 8, 3, 21
    D = _instances(B, n, 3)
    lo, hi = parallel.shard_range(B, rank, world)
    mine = torch.tensor([_oracle_colony(D[b], A, iters, seed, b * A)[0] for b in range(lo, hi)])
    out["gathered"] = parallel.gather_best(mine, B, rank, world).tolist()
    # --- timing helper: max over ranks
    import time
    out["t"] = parallel.barrier_max_time(lambda: time.sleep(0.05 * (rank + 1)), dev, True)
    # --- ant-sharded: delta-tau all-reduce
    Bs, A2 = 2, 12
    Ds = _instances(Bs, n, 9)
    eta = (1.0 / Ds).numpy()

    def sample_fn(tau, gid0, n_local, it):
        ps = [oracle.tsp_sample_scan(oracle.prob_matrix(tau[b].numpy(), eta[b]), n_local, seed, it, b * A2 + gid0)[0]
              for b in range(Bs)]
        return torch.from_numpy(np.stack(ps))

    def cost_fn(paths):
        return torch.from_numpy(np.stack([oracle.tour_costs(Ds[b].numpy(), paths[b].numpy()) for b in range(Bs)]))

    def deposit_fn(zero, paths, costs):
        return torch.from_numpy(np.stack([oracle.pheromone_update_tsp(zero[b].numpy(), paths[b].numpy(),
                                                                       costs[b].numpy(), 1.0) for b in range(Bs)]))

    col = parallel.AntShardedColony(torch.ones(Bs, n, n), A2, 0.9, rank, world, sample_fn, cost_fn, deposit_fn)
    for _ in range(3):
        col.step()
    out["tau"] = col.tau.numpy()
    out["low"] = col.lowest_cost.tolist()
    # --- ant-sharded, exact: all-gather of the tours (int16), full deposit on every rank; 13 ants -> 7 + 6
    A3 = 13

    def sample3(tau, gid0, n_local, it):
        ps = [oracle.tsp_sample_scan(oracle.prob_matrix(tau[b].numpy(), eta[b]), n_local, seed, it, b * A3 + gid0)[0]
              for b in range(Bs)]
        return torch.from_numpy(np.stack(ps))

    def update_fn(tau, paths, costs, elitist, cmin, cmax):
        for b in range(Bs):
            tau[b] = torch.from_numpy(oracle.pheromone_update_tsp(
                tau[b].numpy(), paths[b].numpy(), costs[b].numpy(), 0.9, elitist=bool(elitist),
                clamp_min=0.0 if cmin is None else float(cmin[b]), clamp_max=0.0 if cmax is None else float(cmax[b])))

    col3 = parallel.AntShardedColony(torch.ones(Bs, n, n), A3, 0.9, rank, world, sample3, cost_fn, None,
                                     exchange="tours", update_fn=update_fn)
    for _ in range(3):
        col3.step()
    out["tau3"] = col3.tau.numpy()
    out["low3"] = col3.lowest_cost.tolist()
    out["sp3"] = col3.shortest_path.numpy()
    # --- elitist and MMAS colonies (tsp/aco.py:78-88, 103-107, 116-118), both exchanges
    for tag, exch, kw in (("el_t", "tours", dict(elitist=True)), ("mm_t", "tours", dict(min_max=True)),
                          ("el_d", "delta", dict(elitist=True)), ("mm_d", "delta", dict(min_max=True))):
        c = parallel.AntShardedColony(torch.ones(Bs, n, n), A3, 0.9, rank, world, sample3, cost_fn, deposit_fn,
                                      exchange=exch, update_fn=update_fn, problem_size=n, **kw)
        for _ in range(3):
            c.step()
        out[tag] = (c.tau.numpy(), c.lowest_cost.tolist(), c.shortest_path.numpy())
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # instance-sharded == single process
    B, n, A, iters, seed = 5, 16, 8, 3, 21
    D = _instances(B, n, 3)
    single = [_oracle_colony(D[b], A, iters, seed, b * A)[0] for b in range(B)]
    assert res[0]["gathered"] == pytest.approx(single, rel=0, abs=0)
    assert res[1]["gathered"] == res[0]["gathered"]
    # timing: both ranks report the slower rank's time
    assert res[0]["t"] == pytest.approx(res[1]["t"], abs=1e-9) and res[0]["t"] >= 0.1
    # ant-sharded: replicas stay identical, and equal a single-process run with all ants up to
    # summation order (1e-6 relative)
    assert np.array_equal(res[0]["tau"], res[1]["tau"])
    assert res[0]["low"] == res[1]["low"]
    Bs, A2 = 2, 12
    Ds = _instances(Bs, n, 9)
    for b in range(Bs):
        tau = np.ones((n, n), np.float32)
        eta = (1.0 / Ds[b]).numpy()
        low = np.inf
        for it in range(3):
            paths, _, _ = oracle.tsp_sample_scan(oracle.prob_matrix(tau, eta), A2, seed, it, b * A2)
            costs = oracle.tour_costs(Ds[b].numpy(), paths)
            low = min(low, float(costs.min()))
            tau = oracle.pheromone_update_tsp(tau, paths, costs, 0.9)
        np.testing.assert_allclose(res[0]["tau"][b], tau, rtol=2e-6)
        assert res[0]["low"][b] == pytest.approx(low, rel=1e-6)
    # tour exchange: bit-identical to the single-process colony with all 13 ants, on both ranks
    assert np.array_equal(res[0]["tau3"], res[1]["tau3"]) and res[0]["low3"] == res[1]["low3"]
    for b in range(Bs):
        tau = np.ones((n, n), np.float32)
        eta = (1.0 / Ds[b]).numpy()
        low = np.inf
        for it in range(3):
            paths, _, _ = oracle.tsp_sample_scan(oracle.prob_matrix(tau, eta), 13, seed, it, b * 13)
            costs = oracle.tour_costs(Ds[b].numpy(), paths)
            low = min(low, float(costs.min()))
            tau = oracle.pheromone_update_tsp(tau, paths, costs, 0.9)
        assert np.array_equal(res[0]["tau3"][b].view(np.uint32), tau.view(np.uint32))
        assert res[0]["low3"][b] == low


def _single_colony(d, A, iters, seed, gid0, elitist=False, min_max=False):
    """tsp/aco.py:75-118 in one process on the oracle's kernels: (tau, lowest, shortest path)."""
    n = d.shape[0]
    tau = np.ones((n, n), np.float32) * (np.float32(0.1) if min_max else np.float32(1.0))
    eta = (1.0 / d).numpy()
    low, sp, mx = np.float32(np.inf), None, None
    for it in range(iters):
        paths, _, _ = oracle.tsp_sample_scan(oracle.prob_matrix(tau, eta), A, seed, it, gid0)
        costs = oracle.tour_costs(d.numpy(), paths)
        b = int(np.argmin(costs))
        if costs[b] < low:
            low, sp = costs[b], paths[:, b].copy()
        cmin = cmax = 0.0
        if min_max:
            new = np.float32(np.float32(1.0) / low) * np.float32(n)
            if mx is None:
                tau = tau * np.float32(new / tau.max())
            mx = new
            cmin, cmax = 0.1, float(mx)
        tau = oracle.pheromone_update_tsp(tau, paths, costs, 0.9, elitist=elitist, clamp_min=cmin, clamp_max=cmax)
    return tau, float(low), sp


def test_world2_gloo_elitist_and_mmas():
    """Elitist / MMAS / best-tour tracking of the ant-sharded colony on two gloo ranks against the single-process colony:
    tour exchange and the elitist delta mode bit for bit; (the workers ran inside test_world2_gloo's processes: rerun here)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, seed, Bs, A3 = 16, 21, 2, 13
    Ds = _instances(Bs, n, 9)
    for tag in ("el_t", "mm_t", "el_d", "mm_d"):
        for k in range(3):
            assert np.array_equal(np.asarray(res[0][tag][k]), np.asarray(res[1][tag][k])), tag      # replicas agree
    for b in range(Bs):
        tau, low, sp = _single_colony(Ds[b], A3, 3, seed, b * A3)
        assert np.array_equal(res[0]["sp3"][b], sp) and res[0]["low3"][b] == low
        for tag, kw in (("el_t", dict(elitist=True)), ("mm_t", dict(min_max=True)), ("el_d", dict(elitist=True))):
            tau, low, sp = _single_colony(Ds[b], A3, 3, seed, b * A3, **kw)
            got = res[0][tag]
            assert np.array_equal(got[0][b].view(np.uint32), tau.view(np.uint32)), tag
            assert got[1][b] == low and np.array_equal(got[2][b], sp), tag
        got = res[0]["mm_d"]
        assert got[0][b].min() >= np.float32(0.1) and sorted(got[2][b].tolist()) == list(range(n))


def test_shard_range_covers_everything():
    for total in (1, 5, 8, 64, 511, 512):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [h - l for l, h in spans]
            assert max(sizes) - min(sizes) <= 1
