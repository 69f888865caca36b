#!/usr/bin/env python3
"""bench.py -- ant-tours/sec of the DeepACO rollout hot path on MI355X.

Workload (BASELINE.json `metric`): TSP-500, n_ants = 512, B = 64 random-Euclidean instances
per GPU (instance-sharded across GPUs: weak scaling, no data-path collective).  One "step" =
one colony iteration over the whole batch: tour construction (the dominant kernel) -> tour
costs -> best-so-far tracking -> fused evaporate+deposit pheromone update.  Inputs are resident
in HBM before the timed region.  value = N_gpus * B * A * steps / wall.

Extra objects on the JSON line:
  roofline     dominant kernel (tsp_scan32_kernel for the default workload) timed live with HIP events on the launch
               stream; achieved = algorithmic bytes per launch / average launch duration.
               Algorithmic bytes per ant-tour follow SURVEY.md 8(d):
               (n-1)*8n + 8n + 8n^2/A + 8n  (two f32 rows per step, i64 path writes, amortised
               update) -- the kernel itself streams ONE fused row per step, see DESIGN.md.
  cpu_baseline torch-CPU port of the reference's op sequence (oracle/torch_port.py), timed on
               this host's cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
L2_ROW_STREAM_GBS = 18800.0    # measured L2->L1 ceiling for 2 KB row reads (tools/l2_row_stream_bench.hip)


def make_instances(B, n, seed):
    """coords ~ U[0,1)^2 (tsp/train.ipynb:84), distances with diag 1e9 (tsp/utils.py:4-14)."""
    g = torch.Generator().manual_seed(seed)
    coords = torch.rand(B, n, 2, generator=g)
    dist = torch.cdist(coords, coords)
    idx = torch.arange(n)
    dist[:, idx, idx] = 1e9
    return dist


def bytes_per_tour(n, A):
    return (n - 1) * 8 * n + 8 * n + 8 * n * n / A + 8 * n


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(d, k_sparse, n_ants, iters, budget_s=25.0):
    """Reference CPU path (torch port) on ONE instance of the same workload, `iters` colony
    iterations (bounded: stops early once budget_s is exceeded).  The intra-op thread count is
    calibrated on a few rollout steps (torch's default of one thread per logical CPU is far from
    optimal for [512 x 500] tensors on a many-core host) and reported as `cores`."""
    from oracle import torch_port
    _, idx = torch.topk(d, k=k_sparse, dim=1, largest=False)
    sparse = torch.full_like(d, 1e10)
    sparse.scatter_(1, idx, torch.gather(d, 1, idx))
    heu = 1 / sparse
    n = d.shape[0]
    ncpu = os.cpu_count() or 1
    best_t, best_threads = None, 1
    for th in sorted({t for t in (4, 8, 16, 32, 64, ncpu) if t <= ncpu}):
        torch.set_num_threads(th)
        tau = torch.ones_like(d)
        cur = torch.randint(0, n, (n_ants,))
        mask = torch.ones(n_ants, n)
        t0 = time.perf_counter()
        for _ in range(12):                      # 12 steps of tsp/aco.py pick_move's op stream
            w = (tau[cur] ** 1) * (heu[cur] ** 1) * mask
            cur = torch.distributions.Categorical(w + 1e-30).sample()
        dt = time.perf_counter() - t0
        log(f"cpu calibration: {th} threads -> {dt/12*1e3:.2f} ms/step")
        if best_t is None or dt < best_t:
            best_t, best_threads = dt, th
    torch.set_num_threads(best_threads)
    torch.manual_seed(1234)
    tau = torch.ones_like(d)
    lowest, done = float("inf"), 0
    t0 = time.perf_counter()
    for _ in range(iters):
        paths = torch_port.rollout(tau, heu, n_ants)
        costs = torch_port.tour_lengths(d, paths)
        lowest = min(lowest, float(costs.min()))
        tau = torch_port.deposit(tau, paths, costs, 0.9)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    log(f"cpu port: {done} iterations in {dt:.1f}s")
    out = {"value": n_ants * done / dt, "unit": "ant-tours/s", "cores": best_threads, "kind": "port",
           "sample": f"1 instance x {n_ants} ants x {done} colony iterations of the same TSP-{n} workload "
                     f"(oracle/torch_port.py: the reference's aten op sequence, torch {torch.__version__} CPU, "
                     f"{best_threads} intra-op threads calibrated on a {ncpu}-CPU host)"}
    return out, lowest, done


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nodes", type=int, default=500)
    ap.add_argument("--ants", type=int, default=512)
    ap.add_argument("--batch", type=int, default=64, help="instances per GPU")
    ap.add_argument("--sampler", default="scan", choices=["scan", "scan_wave", "race"])
    ap.add_argument("--k-sparse", type=int, default=None)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for the barrier / max-time reduce (nccl = RCCL)")
    ap.add_argument("--shard", default="instances", choices=["instances", "ants"],
                    help="instances: B colonies per GPU, no collective (weak scaling, default); "
                         "ants: the same B colonies on every GPU, A/N ants each, one all-reduce of the "
                         "pheromone deposits per iteration (strong scaling)")
    ap.add_argument("--exchange", default="tours", choices=["tours", "delta"],
                    help="--shard ants: all-gather of the tours (int16; exact, default) or all-reduce of delta-tau")
    ap.add_argument("--force-device", type=int, default=None,
                    help="testing only: put every rank on this GPU (needs --dist-backend gloo)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if distributed:
        import torch.distributed as dist_pkg
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    dev_index = local_rank if args.force_device is None else args.force_device
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if distributed:
        if args.dist_backend == "nccl":
            dist_pkg.init_process_group("nccl", device_id=dev)
        else:
            dist_pkg.init_process_group("gloo")

    from deepaco_amd import engine
    from deepaco_amd.parallel import barrier_max_time

    n, A, B = args.nodes, args.ants, args.batch
    k_sparse = args.k_sparse or max(5, n // 10)
    ant_sharded = args.shard == "ants"
    dist_cpu = make_instances(B, n, 1234 + (0 if ant_sharded else rank))
    if ant_sharded:
        d_dev = dist_cpu.to(dev)
        _, idx = torch.topk(d_dev, k=k_sparse, dim=2, largest=False)
        sparse = torch.full_like(d_dev, 1e10)
        sparse.scatter_(2, idx, torch.gather(d_dev, 2, idx))
        colony = engine.ant_sharded_tsp(d_dev, A, rank, world, heuristic=(1 / sparse).contiguous(),
                                        sampler=args.sampler, seed=1234, exchange=args.exchange)
        _step = colony.step
        colony.step = lambda events=None: _step()
    else:
        colony = engine.BatchedTSP(dist_cpu.to(dev), n_ants=A, sampler=args.sampler, seed=1234,
                                   ant_gid0=rank * B * A)
        colony.sparsify(k_sparse)
        colony.heuristic = colony.heuristic.contiguous()

    log(f"rank {rank}/{world}: TSP-{n} x {A} ants x {B} instances, sampler={args.sampler}")
    for _ in range(args.warmup):
        colony.step()
    torch.cuda.synchronize()
    log("warm-up done")
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:      # create the handles; the library re-records them around the kernel
        a.record(); b.record()
    torch.cuda.synchronize()

    def timed():
        for s in range(args.steps):
            colony.step(events=ev[s])

    elapsed = barrier_max_time(timed, dev, distributed)
    log(f"timed region: {elapsed*1e3:.1f} ms for {args.steps} steps")
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps if not ant_sharded else None
    tours = (B * A if ant_sharded else world * B * A) * args.steps
    value = tours / elapsed
    bpt = bytes_per_tour(n, A)
    per_launch = (B * (colony.hi - colony.lo) if ant_sharded else B * A) * bpt
    achieved = per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms else None      # GB/s, dominant kernel, this rank
    gpu_best = colony.lowest_cost.detach().cpu()
    # daco_tsp_sample's layout rule: 4 / 2 / 1 ants per wavefront
    lanes = 64 if args.sampler != "scan" or n > 512 else (16 if n <= 256 else 32)
    kernel_name = {16: "scan16_kernel", 32: "tsp_scan32_kernel", 64: "tsp_sample_kernel"}[lanes]
    row_floats = (n + 4 * lanes - 1) // (4 * lanes) * (4 * lanes) if lanes < 64 else ((n + 255) // 256 * 256 if n > 128 else n)

    if rank == 0:
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(f"tsp{n}_a{A}_b{B}_{args.sampler}")
            except Exception:
                traffic = None
        line = {
            "metric": "ant-tours/sec, TSP-500 n_ants=512", "value": value, "unit": "ant-tours/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if ant_sharded else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"TSP-{n} random-Euclidean, n_ants={A}, {B} instances per GPU, "
                                   f"AS update, heuristic 1/d sparsified k={k_sparse}, sampler={args.sampler}",
                       "nodes": n, "n_ants": A, "instances_per_gpu": B, "sampler": args.sampler,
                       "parallelism": f"{'ant' if ant_sharded else 'instance'}-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": achieved / PEAK_HBM_GBS if achieved else None, "traffic": traffic,
                         "kernel": kernel_name, "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": per_launch,
                         "note": "the transition rows (tau^a*eta^b fused, one row per ant-step) are served by L2, "
                                 "not HBM, so the algorithmic GB/s of SURVEY 8(d) exceeds the HBM peak and `traffic` "
                                 "(measured HBM bytes) is far below the algorithmic bytes; the binding resource is "
                                 "L2->L1 row streaming, see roofline_l2 (DESIGN.md 3.1, profiles/)"},
            # the bytes the kernel really moves per launch: one padded fused row per ant-step out of L2
            "roofline_l2": None if not kern_ms else {
                "bound": "l2", "unit": "GB/s", "peak": L2_ROW_STREAM_GBS,
                "achieved": B * A * (n - 1) * 4.0 * row_floats / (kern_ms * 1e-3) / 1e9,
                "frac": B * A * (n - 1) * 4.0 * row_floats / (kern_ms * 1e-3) / 1e9 / L2_ROW_STREAM_GBS,
                "peak_spec": 34500.0,
                "note": "row bytes streamed per launch / kernel time; peak = 18.8 TB/s, what a bare kernel that only "
                        "streams the same 2 KB rows out of L2 reaches on MI355X at any occupancy "
                        "(tools/l2_row_stream_bench.hip, profiles/r01_g_l2_row_stream.txt); peak_spec = the 34.5 TB/s "
                        "aggregate L2 figure of MI355X_MICROARCH.md"},
            "gpu_mean_best_cost": float(gpu_best.mean()),
        }
        if world == 1 and not args.no_cpu:
            cb, cpu_best, done = cpu_baseline(dist_cpu[0], k_sparse, A, args.cpu_iters)
            line["cpu_baseline"] = cb
            # best-cost gap at equal iterations on the same instance (fresh GPU colonies, 16 seeds)
            reps = 16
            gcol = engine.BatchedTSP(dist_cpu[:1].repeat(reps, 1, 1).to(dev), n_ants=A, sampler=args.sampler, seed=99)
            gcol.sparsify(k_sparse)
            gcol.run(done)
            gb = float(gcol.lowest_cost.mean())
            line["best_cost_gap"] = {"gpu_mean_best": gb, "cpu_best": cpu_best, "gap": (gb - cpu_best) / cpu_best,
                                     "instances": 1, "iterations": done,
                                     "note": "same instance, equal iterations; GPU value is the mean over 16 seeds"}
            line["speedup_vs_cpu"] = value / cb["value"]
        print(json.dumps(line), flush=True)
    if distributed:
        dist_pkg.barrier()
        dist_pkg.destroy_process_group()


if __name__ == "__main__":
    main()
