#!/usr/bin/env python3
"""bench.py -- ant-tours/sec of the DeepACO rollout hot path on MI355X.

Workload (BASELINE.json `metric`): TSP-500, n_ants = 512, B = 64 random-Euclidean instances
per GPU (instance-sharded across GPUs: weak scaling, no data-path collective).  One "step" =
one colony iteration over the whole batch: tour construction (the dominant kernel) -> tour
costs -> best-so-far tracking -> fused evaporate+deposit pheromone update.  Inputs are resident
in HBM before the timed region.  value = N_gpus * B * A * steps / wall.

Extra objects on the JSON line:
  roofline     dominant kernel (tsp_sample_kernel) timed live with HIP events on the launch
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


def cpu_baseline(dist_cpu, k_sparse, n_ants, iters, gap_instances, gap_iters):
    from oracle import torch_port
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    d = dist_cpu[0]
    _, idx = torch.topk(d, k=k_sparse, dim=1, largest=False)
    sparse = torch.full_like(d, 1e10)
    sparse.scatter_(1, idx, torch.gather(d, 1, idx))
    heu = 1 / sparse
    torch.manual_seed(1234)
    torch_port.colony_iterations(d[:64, :64].contiguous(), heu[:64, :64].contiguous(), 8, 1)   # warm-up
    t0 = time.perf_counter()
    torch_port.colony_iterations(d, heu, n_ants, iters)
    dt = time.perf_counter() - t0
    out = {"value": n_ants * iters / dt, "unit": "ant-tours/s", "cores": threads, "kind": "port",
           "sample": f"1 instance x {n_ants} ants x {iters} iterations of the same TSP-{d.shape[0]} workload "
                     f"(torch {torch.__version__} CPU, {threads} threads; oracle/torch_port.py)"}
    best = []
    for b in range(gap_instances):
        db = dist_cpu[b]
        _, idx = torch.topk(db, k=k_sparse, dim=1, largest=False)
        sp = torch.full_like(db, 1e10)
        sp.scatter_(1, idx, torch.gather(db, 1, idx))
        low, _ = torch_port.colony_iterations(db, 1 / sp, n_ants, gap_iters)
        best.append(low)
    return out, best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nodes", type=int, default=500)
    ap.add_argument("--ants", type=int, default=512)
    ap.add_argument("--batch", type=int, default=64, help="instances per GPU")
    ap.add_argument("--sampler", default="scan", choices=["scan", "race"])
    ap.add_argument("--k-sparse", type=int, default=None)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-iters", type=int, default=2)
    ap.add_argument("--gap-instances", type=int, default=1)
    ap.add_argument("--gap-iters", type=int, default=3)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if distributed:
        import torch.distributed as dist_pkg
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_pkg.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from deepaco_amd import engine
    from deepaco_amd.parallel import barrier_max_time

    n, A, B = args.nodes, args.ants, args.batch
    k_sparse = args.k_sparse or max(5, n // 10)
    dist_cpu = make_instances(B, n, 1234 + rank)
    colony = engine.BatchedTSP(dist_cpu.to(dev), n_ants=A, sampler=args.sampler, seed=1234,
                               ant_gid0=rank * B * A)
    colony.sparsify(k_sparse)
    colony.heuristic = colony.heuristic.contiguous()

    for _ in range(args.warmup):
        colony.step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in ev:      # create the handles; the library re-records them around the kernel
        a.record(); b.record()
    torch.cuda.synchronize()

    def timed():
        for s in range(args.steps):
            colony.step(events=ev[s])

    elapsed = barrier_max_time(timed, dev, distributed)
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    tours = world * B * A * args.steps
    value = tours / elapsed
    bpt = bytes_per_tour(n, A)
    achieved = (B * A * bpt) / (kern_ms * 1e-3) / 1e9          # GB/s, dominant kernel, this rank
    gpu_best = colony.lowest_cost.detach().cpu()

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
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"TSP-{n} random-Euclidean, n_ants={A}, {B} instances per GPU, "
                                   f"AS update, heuristic 1/d sparsified k={k_sparse}, sampler={args.sampler}",
                       "nodes": n, "n_ants": A, "instances_per_gpu": B, "sampler": args.sampler,
                       "parallelism": f"instance-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": achieved / PEAK_HBM_GBS, "traffic": traffic,
                         "kernel": "tsp_sample_kernel", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": B * A * bpt},
            "gpu_mean_best_cost": float(gpu_best.mean()),
        }
        if world == 1 and not args.no_cpu:
            cb, cpu_best = cpu_baseline(dist_cpu, k_sparse, A, args.cpu_iters, args.gap_instances, args.gap_iters)
            line["cpu_baseline"] = cb
            # best-cost gap at equal iterations on the same instances (fresh GPU colonies)
            gcol = engine.BatchedTSP(dist_cpu[:args.gap_instances].to(dev), n_ants=A, sampler=args.sampler, seed=99)
            gcol.sparsify(k_sparse)
            gcol.run(args.gap_iters)
            gb = gcol.lowest_cost.cpu()
            cbm = sum(cpu_best) / len(cpu_best)
            line["best_cost_gap"] = {"gpu_mean_best": float(gb.mean()), "cpu_mean_best": cbm,
                                     "gap": (float(gb.mean()) - cbm) / cbm,
                                     "instances": args.gap_instances, "iterations": args.gap_iters}
            line["speedup_vs_cpu"] = value / cb["value"]
        print(json.dumps(line), flush=True)
    if distributed:
        dist_pkg.barrier()
        dist_pkg.destroy_process_group()


if __name__ == "__main__":
    main()
