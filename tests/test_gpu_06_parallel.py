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


def _worker(rank, world, port, q, B, n, A, iters, exchange):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepaco_amd import engine
    dev = torch.device("cuda:0")
    col = engine.ant_sharded_tsp(_instances(B, n, 5).to(dev), A, rank, world, seed=77, exchange=exchange)
    for _ in range(iters):
        col.step()
    torch.cuda.synchronize()
    q.put((rank, col.tau.cpu().numpy(), col.lowest_cost.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange,n,A", [("tours", 40, 13), ("tours", 300, 16), ("delta", 40, 12)])
def test_two_ranks_equal_single_process(exchange, n, A):
    from deepaco_amd import engine
    B, iters, world = 2, 4, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, B, n, A, iters, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, tau, low = q.get(timeout=300)
        res[r] = (tau, low)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = engine.BatchedTSP(_instances(B, n, 5).to("cuda:0"), n_ants=A, seed=77)
    single.run(iters)
    ref_tau, ref_low = single.pheromone.cpu().numpy(), single.lowest_cost.cpu().numpy()
    assert (res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all()          # replicas agree
    if exchange == "tours":
        assert (res[0][0].view("uint32") == ref_tau.view("uint32")).all()
        assert (res[0][1] == ref_low).all()
    else:
        # colony-wide ant ids: the same tours as the single-GPU colony, deposits summed in a different order
        assert res[0][0].shape == ref_tau.shape and (res[0][0] > 0).all()
        import numpy as np
        np.testing.assert_allclose(res[0][0], ref_tau, rtol=2e-5)
        np.testing.assert_allclose(res[0][1], ref_low, rtol=0, atol=0)


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
