"""Two ranks (gloo rendezvous, both on cuda:0) of the ant-sharded colony with the tour exchange against the
single-process colony: same seed -> the same pheromone bit for bit (colony-wide ant ids, full deposit in ant
order on every rank).  The delta exchange agrees to summation order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


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


def _cvrp_instances(B, n, seed):
    g = torch.Generator().manual_seed(seed)
    loc = torch.cat((torch.full((B, 1, 2), 0.5), torch.rand(B, n, 2, generator=g)), 1)
    dem = torch.cat((torch.zeros(B, 1), torch.randint(1, 10, (B, n), generator=g).float()), 1)
    d = torch.cdist(loc, loc)
    i = torch.arange(n + 1)
    d[:, i, i] = 1e-10
    return d, dem


def _worker(rank, world, port, q, B, n, A, iters, exchange, kw, problem="tsp", backend="gloo"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda:0")
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepaco_amd import engine
    if problem == "tsp":
        kw = dict(kw)
        k_sparse = kw.pop("k_sparse", None)
        d = _instances(B, n, 5).to(dev)
        if k_sparse:                                             # what BatchedTSP.sparsify(k) makes of the heuristic
            _, idx = torch.topk(d, k=k_sparse, dim=2, largest=False)
            kw["heuristic"] = 1 / torch.full_like(d, 1e10).scatter_(2, idx, torch.gather(d, 2, idx))
            kw["head_k"] = k_sparse
        col = engine.ant_sharded_tsp(d, A, rank, world, seed=77, exchange=exchange, **kw)
    else:
        d, dem = _cvrp_instances(B, n, 5)
        col = engine.ant_sharded_cvrp(d.to(dev), dem.to(dev), A, rank, world, capacity=30, seed=77, exchange=exchange, **kw)
    for _ in range(iters):
        col.step()
    torch.cuda.synchronize()
    q.put((rank, col.tau.cpu().numpy(), col.lowest_cost.cpu().numpy(), col.shortest_path.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _run_ranks(world, *args, **kwargs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q) + args, kwargs=kwargs) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, tau, low, sp = q.get(timeout=300)
        res[r] = (tau, low, sp)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(1, world):                                                        # replicas agree
        assert (res[0][0] == res[r][0]).all() and (res[0][1] == res[r][1]).all() and (res[0][2] == res[r][2]).all()
    return res[0]


@pytest.mark.parametrize("exchange,n,A,kw", [("tours", 40, 13, {}), ("tours", 300, 16, {}), ("delta", 40, 12, {}),
                                             ("tours", 60, 13, dict(elitist=True)), ("tours", 60, 14, dict(min_max=True)),
                                             ("delta", 60, 12, dict(elitist=True)), ("delta", 60, 12, dict(min_max=True))])
def test_two_ranks_equal_single_process(exchange, n, A, kw):
    """AS, elitist and MMAS colonies (tsp/aco.py:75-118): with the tour exchange the two-rank colony IS the single-GPU one --
    pheromone, best costs and best tours bit for bit; with the delta exchange the tours are the same (colony-wide ant
    ids), the elitist deposit is exact (one ant, applied by every rank), the AS / MMAS sums agree to summation order."""
    import numpy as np
    from deepaco_amd import engine
    B, iters, world = 2, 4, 2
    tau, low, sp = _run_ranks(world, B, n, A, iters, exchange, kw)
    single = engine.BatchedTSP(_instances(B, n, 5).to("cuda:0"), n_ants=A, seed=77, **kw)
    single.run(iters)
    ref_tau, ref_low, ref_sp = single.pheromone.cpu().numpy(), single.lowest_cost.cpu().numpy(), single.shortest_path.cpu().numpy()
    if exchange == "tours" or kw.get("elitist"):
        assert (tau.view("uint32") == ref_tau.view("uint32")).all()
        assert (low == ref_low).all() and (sp == ref_sp).all()
    elif kw.get("min_max"):
        # the pheromone feeds back into the draws: after the first iteration the tours may part ways; the bounds hold
        assert tau.shape == ref_tau.shape and (tau >= 0.1 * (1 - 1e-6)).all()
        assert (np.sort(sp, axis=1) == np.arange(n)).all()
    else:
        # colony-wide ant ids: the same tours as the single-GPU colony, deposits summed in a different order
        np.testing.assert_allclose(tau, ref_tau, rtol=2e-5)
        np.testing.assert_allclose(low, ref_low, rtol=0, atol=0)
        assert (sp == ref_sp).all()


@pytest.mark.parametrize("kw,n", [(dict(sampler="auto", k_sparse=20), 200), (dict(local_search="2opt", fixed_start=0), 120),
                                  (dict(local_search="nls", fixed_start=0, sampler="auto", k_sparse=15), 150)])
def test_two_ranks_with_head_rows_and_local_search(kw, n):
    """The ant-sharded colony on head / tail rows (sampler 'auto' after sparsify) and with the local search of tsp_nls
    (tsp_nls/aco.py:105-129: every rank improves its own ants' tours before the exchange): two ranks, tour exchange ==
    BatchedTSP bit for bit (pheromone, best costs, best tours)."""
    from deepaco_amd import engine
    B, A, iters, world = 2, 13, 3, 2
    tau, low, sp = _run_ranks(world, B, n, A, iters, "tours", kw)
    skw = {k: v for k, v in kw.items() if k != "k_sparse"}
    single = engine.BatchedTSP(_instances(B, n, 5).to("cuda:0"), n_ants=A, seed=77, **skw)
    if kw.get("k_sparse"):
        single.sparsify(kw["k_sparse"])
    single.run(iters)
    assert (tau.view("uint32") == single.pheromone.cpu().numpy().view("uint32")).all()
    assert (low == single.lowest_cost.cpu().numpy()).all() and (sp == single.shortest_path.cpu().numpy()).all()


@pytest.mark.parametrize("exchange,kw", [("tours", {}), ("tours", dict(elitist=True)), ("tours", dict(min_max=True)), ("delta", {})])
def test_two_ranks_cvrp_equal_single_process(exchange, kw):
    """cvrp/aco.py:67-130 ant-sharded (directed deposits, floor 1e-10): the two-rank colony against BatchedCVRP."""
    import numpy as np
    from deepaco_amd import engine
    B, n, A, iters, world = 2, 30, 13 if exchange == "tours" else 12, 3, 2
    tau, low, sp = _run_ranks(world, B, n, A, iters, exchange, kw, problem="cvrp")
    d, dem = _cvrp_instances(B, n, 5)
    single = engine.BatchedCVRP(d.to("cuda:0"), dem.to("cuda:0"), n_ants=A, capacity=30, seed=77, **kw)
    single.run(iters)
    ref_tau, ref_low, ref_sp = single.pheromone.cpu().numpy(), single.lowest_cost.cpu().numpy(), single.shortest_path.cpu().numpy()
    # tau[0][0] only records padding: cvrp/aco.py:107-130 deposits on the (0, 0) pairs behind every ant that finished before the
    # longest one of ITS instance's trimmed paths; the exchanged sequences are the untrimmed [2n+1] buffers, where every ant has
    # such a pair.  The sampler never reads the entry (the depot is closed for an ant standing on it, cvrp/aco.py:176-180).
    tau[:, 0, 0] = ref_tau[:, 0, 0]
    if exchange == "tours":
        assert (tau.view("uint32") == ref_tau.view("uint32")).all()
        assert (low == ref_low).all() and (sp == ref_sp).all()
    else:
        np.testing.assert_allclose(tau, ref_tau, rtol=2e-5)
        np.testing.assert_allclose(low, ref_low, rtol=0, atol=0)


@pytest.mark.parametrize("exchange,kw", [("tours", dict(min_max=True)), ("delta", {}), ("delta", dict(elitist=True))])
def test_world_size_one_on_rccl(exchange, kw):
    """The RCCL code path on ONE GPU: init_process_group("nccl", device_id=...), the collectives of
    parallel.py on device tensors (no host staging), barrier_max_time, gather_best -- world size 1, the same calls the driver's
    8-GPU run makes.  (A world-size-1 colony skips its exchanges, so they are called directly on the colony's buffers too.)"""
    from deepaco_amd import engine
    B, n, A, iters = 2, 80, 16, 3
    tau, low, sp = _run_ranks(1, B, n, A, iters, exchange, kw, backend="nccl")
    single = engine.BatchedTSP(_instances(B, n, 5).to("cuda:0"), n_ants=A, seed=77, **kw)
    single.run(iters)
    import numpy as np
    if exchange == "delta" and not kw:          # AS deposits are summed before they meet tau: equal to summation order
        np.testing.assert_allclose(tau, single.pheromone.cpu().numpy(), rtol=2e-5)
    else:
        assert (tau.view("uint32") == single.pheromone.cpu().numpy().view("uint32")).all()
    assert (sp == single.shortest_path.cpu().numpy()).all()


def _rccl_collectives(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from deepaco_amd import parallel
    out = {}
    x = torch.arange(12, dtype=torch.float32, device=dev).view(3, 4)
    parallel.all_reduce_(x)
    out["sum"] = x.cpu().tolist()
    m = torch.tensor([3.0, 1.0], device=dev)
    parallel.all_reduce_(m, dist.ReduceOp.MIN)
    out["min"] = m.cpu().tolist()
    i32 = torch.arange(6, dtype=torch.int32, device=dev)
    parallel.all_reduce_(i32)
    out["i32"] = i32.cpu().tolist()
    src = (torch.arange(10, dtype=torch.int16, device=dev) - 3).view(2, 5)
    dst = torch.zeros((world, 2, 5), dtype=torch.int16, device=dev)
    parallel.all_gather_into_(dst.view(torch.uint8), src.view(torch.uint8))      # the tours' exchange: int16 bytes as uint8
    out["i16"] = dst.cpu().tolist()
    f = torch.rand(4, device=dev)
    g = torch.zeros((world, 4), device=dev)
    parallel.all_gather_into_(g, f)
    out["f_ok"] = bool(torch.equal(g[0], f))
    out["best"] = parallel.gather_best(torch.tensor([2.0, 5.0], device=dev), 2, rank, world).cpu().tolist()
    out["t"] = parallel.barrier_max_time(lambda: torch.cuda.synchronize(), dev, True)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_collectives_world_size_one():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_collectives, args=(0, 1, _free_port(), q))
    p.start()
    _, out = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert out["sum"] == [[0.0, 1.0, 2.0, 3.0], [4.0, 5.0, 6.0, 7.0], [8.0, 9.0, 10.0, 11.0]] and out["min"] == [3.0, 1.0]
    assert out["i32"] == list(range(6)) and out["i16"] == [[[-3, -2, -1, 0, 1], [2, 3, 4, 5, 6]]]
    assert out["f_ok"] and out["best"] == [2.0, 5.0] and out["t"] >= 0


def _bench_line(*args, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], capture_output=True, text=True,
                         timeout=600, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]               # ONE JSON line, from rank 0
    assert out.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) <= 4096   # the LAST line, compact
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` (no torchrun) starts two ranks by itself; both on cuda:0 here, gloo rendezvous.
    Instance-sharded: n_gpus = 2, twice the tours of the one-rank run per step, no data-path collective."""
    common = ("--no-cpu", "--no-extras", "--min-seconds", "0", "--steps", "3", "--warmup", "1", "--nodes", "120",
              "--ants", "64", "--batch", "4")
    one = _bench_line(*common)
    two = _bench_line("--gpus", "2", "--dist-backend", "gloo", "--force-device", "0", *common)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["rccl"]["ranks"] == 2 and two["config"]["parallelism"] == "instance-sharded x2"
    assert two["scaling"] == "weak" and two["value"] > 0 and one["value"] > 0
    assert one["roofline"]["bound"] == "l2" and 0 < one["roofline"]["frac"] < 1.0
    # ant-sharded, exact exchange: strong scaling, same colony on both ranks
    ants = _bench_line("--gpus", "2", "--dist-backend", "gloo", "--force-device", "0", "--shard", "ants", *common)
    assert ants["n_gpus"] == 2 and ants["scaling"] == "strong"


def test_bench_four_ranks_on_one_gpu_both_shardings():
    """Four ranks (all on cuda:0, gloo rendezvous): instance-sharded (weak: four times the tours per step), and the
    ant-sharded colony with the delta-tau all-reduce as its one data-path collective (strong), pre-allocated exchange
    buffers.  What the driver's --gpus 4 run does, short of four devices."""
    common = ("--no-cpu", "--no-extras", "--min-seconds", "0", "--steps", "3", "--warmup", "1", "--nodes", "120",
              "--ants", "64", "--batch", "4", "--gpus", "4", "--dist-backend", "gloo", "--force-device", "0")
    inst = _bench_line(*common)
    assert inst["n_gpus"] == 4 and inst["rccl"]["ranks"] == 4 and inst["scaling"] == "weak"
    assert inst["config"]["parallelism"] == "instance-sharded x4" and inst["value"] > 0
    delta = _bench_line("--shard", "ants", "--exchange", "delta", *common)
    assert delta["n_gpus"] == 4 and delta["scaling"] == "strong" and "all-reduce of delta-tau" in delta["rccl"]["data_path_collective"]
    assert delta["gpu_mean_best_cost"] > 0


def test_bench_eight_ranks_the_drivers_launch_path():
    """Eight ranks on one GPU (gloo rendezvous): the world size of the driver's `--gpus 8` scaling run, instance-sharded
    (BASELINE config 5's partitioning: no data-path collective, weak scaling) and `--shard ants --exchange delta` (what
    `--config c5 --shard ants --exchange delta` runs at TSP-1000 x 2048 x 64, here at a test size: the delta-tau all-reduce on
    the data path).  And through torchrun, as the driver starts it."""
    import subprocess
    import sys
    common = ("--no-cpu", "--no-extras", "--min-seconds", "0", "--steps", "2", "--warmup", "1", "--nodes", "130",
              "--ants", "32", "--batch", "2", "--gpus", "8", "--dist-backend", "gloo", "--force-device", "0")
    inst = _bench_line(*common)
    assert inst["n_gpus"] == 8 and inst["rccl"]["ranks"] == 8 and inst["scaling"] == "weak"
    assert inst["config"]["parallelism"] == "instance-sharded x8" and inst["value"] > 0
    delta = _bench_line("--shard", "ants", "--exchange", "delta", *common)
    assert delta["n_gpus"] == 8 and delta["scaling"] == "strong" and "all-reduce of delta-tau" in delta["rccl"]["data_path_collective"]
    # the driver's own command line: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
                          "127.0.0.1", "--master-port", "29631", os.path.join(root, "bench.py"), "--gpus", "4", "--no-cpu",
                          "--no-extras", "--min-seconds", "0", "--steps", "2", "--warmup", "1", "--nodes", "130", "--ants", "32",
                          "--batch", "2", "--dist-backend", "gloo", "--force-device", "0"], capture_output=True, text=True,
                         timeout=600, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    import json
    assert json.loads(lines[0])["n_gpus"] == 4


def test_bench_world_size_one_on_rccl():
    """bench.py --force-dist: the distributed code path (init_process_group("nccl", device_id), barrier / max reduce on device
    tensors, the all-reduce bus-bandwidth probe, the `rccl` object on the line) with ONE rank on one GPU."""
    common = ("--no-cpu", "--no-extras", "--min-seconds", "0", "--steps", "3", "--warmup", "1", "--nodes", "120",
              "--ants", "64", "--batch", "4", "--force-dist")
    inst = _bench_line(*common)
    assert inst["n_gpus"] == 1 and inst["rccl"]["ranks"] == 1 and inst["rccl"]["backend"] == "nccl"
    assert inst["rccl"]["allreduce_ms"] > 0 and inst["value"] > 0
    ants = _bench_line("--shard", "ants", "--exchange", "delta", *common)
    assert ants["rccl"]["backend"] == "nccl" and ants["scaling"] == "strong" and ants["gpu_mean_best_cost"] > 0


def test_c_abi_allreduce_of_delta_tau_on_a_one_rank_rccl_communicator():
    """include/deepaco_hip.h daco_allreduce_delta_tau (the collective entry a binder without torch.distributed
    calls): a communicator of one rank created through the RCCL this process holds (torch's copy), the deposits of a colony's ants
    summed in place -- one rank: the sum is the buffer itself -- on torch's current stream, then tau <- decay tau + delta equal to
    the fused update up to the one rounding SURVEY.md 8(e) allows."""
    import ctypes as C
    from deepaco_amd import _lib, engine
    rccl = None
    for line in open("/proc/self/maps"):
        if "librccl" in line:
            rccl = C.CDLL(line.split()[-1])
            break
    if rccl is None:                          # (not mapped yet: torch loads it lazily on some builds)
        import os
        rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    torch.cuda.set_device(0)
    dev = lambda: torch.device("cuda:0")
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        n, A, B = 60, 16, 2
        g = torch.Generator().manual_seed(1)
        c = torch.rand(B, n, 2, generator=g)
        d = (c[:, :, None] - c[:, None]).norm(dim=-1)
        d[:, torch.arange(n), torch.arange(n)] = 1e9
        D = d.to(dev())
        tau = (torch.rand(B, n, n, generator=g) + 0.5).to(dev())
        paths, _, _, _, costs, nbr = engine.tsp_sample(tau, 1 / D, A, mode="scan", seed=2, batch=B, dist=D, want_nbr=True)
        delta = engine.pheromone_update_(torch.zeros_like(tau), paths, costs, 1.0, nbr=nbr)
        before = delta.clone()
        rc = _lib.lib().daco_allreduce_delta_tau(comm, engine._stream(dev()), delta.data_ptr(), delta.numel())
        _lib.check(rc, "daco_allreduce_delta_tau")
        torch.cuda.synchronize()
        assert torch.equal(delta, before)
        fused = engine.pheromone_update_(tau.clone(), paths, costs, 0.9, nbr=nbr)
        torch.testing.assert_close(tau * 0.9 + delta, fused, rtol=1e-5, atol=0)
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
