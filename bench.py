#!/usr/bin/env python3
"""bench.py -- ant-tours/sec of the DeepACO rollout hot path on MI355X.

Workload (BASELINE.json `metric`): TSP-500, n_ants = 512, B = 64 random-Euclidean instances
per GPU (instance-sharded across GPUs: weak scaling, no data-path collective).  One "step" =
one colony iteration over the whole batch: tour construction (the dominant kernel) -> tour
costs -> best-so-far tracking -> fused evaporate+deposit pheromone update.  Inputs are resident
in HBM before the timed region.  value = N_gpus * B * A * steps / wall.

Launching: `python bench.py --gpus N` starts N ranks itself (one process per GPU, RCCL process group);
under torchrun (RANK / WORLD_SIZE in the environment) it is one of the ranks.

Objects on the JSON line besides the contract's keys:
  roofline      dominant kernel (tsp_scan32_kernel for the default workload) timed live with HIP events on the
                launch stream.  The fused transition rows live in L2/MALL, so the bound is the L2 -> CU row
                stream: achieved = row bytes per launch / kernel time against the guide's aggregate L2 figure
                (MI355X_MICROARCH.md: 34.5 TB/s).  `algorithmic` holds SURVEY.md 8(d)'s figure
                ((n-1)*8n + 8n + 8n^2/A + 8n bytes per ant-tour) over the same time and its ratio to the HBM peak;
                `hbm` the counter-measured HBM-side bytes per launch (from profiles/, source named) next to the
                compulsory floor.
  sustained     the same step loop run for >= --min-seconds after the timed region (so that samplers outside
                this process can see the GPU busy); not the metric.
  cpu_baseline  torch-CPU port of the reference's op sequence (oracle/torch_port.py) on this host's cores:
                min(instances, host CPUs / 4) colonies of the same workload side by side (one process each, two
                intra-op threads: 128 of the 256 hardware threads on the GPU box = its physical cores; the host's
                aggregate rate is flat from ~16 colonies on, the op sequence is memory-bound), --cpu-iters iterations
                each (rank 0, N = 1 only), and the best-cost gap against the GPU path on the same instances at the same
                number of iterations.
  extras        BASELINE.json's other single-GPU configurations (2, 3, 4, the per-GPU share of 5), the parity modes of
                the sampler, the learned heuristic and the GNN forward, each with its own roofline object and its own
                cpu_baseline on a bounded sample (N = 1 only; --no-extras skips).
  rccl          N > 1: ranks, backend and the measured all-reduce bus bandwidth of a [B, n, n] f32 buffer.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md, HBM section)
PEAK_L2_GBS = 34500.0          # aggregate L2 bandwidth (MI355X_MICROARCH.md, L2 section)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nodes", type=int, default=500)
    ap.add_argument("--ants", type=int, default=512)
    ap.add_argument("--batch", type=int, default=64, help="instances per GPU")
    ap.add_argument("--sampler", default="auto", choices=["auto", "scan", "scan_wave", "race", "scan_sparse"],
                    help="auto (the colonies' default): the scan draw, on head / tail rows where they apply -- here they do: the "
                         "heuristic is 1/d sparsified to k = n // 10 entries per row; scan: the dense scan")
    ap.add_argument("--k-sparse", type=int, default=None)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the other configurations")
    ap.add_argument("--cpu-instances", type=int, default=0,
                    help="colonies of the cpu_baseline leg, side by side (0 = min(instances, host CPUs // 4))")
    ap.add_argument("--cpu-iters", type=int, default=20,
                    help="colony iterations of every cpu_baseline colony (the best-cost gap is taken at this many iterations)")
    ap.add_argument("--cpu-seconds", type=float, default=150.0, help="safety stop of a cpu_baseline colony")
    ap.add_argument("--config", default="headline", choices=["headline", "c5"],
                    help="c5: the per-GPU share of BASELINE config 5 (TSP-1000, 2048 ants, 64 instances per GPU)")
    ap.add_argument("--min-seconds", type=float, default=12.0, help="length of the sustained (untimed-by-metric) loop")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams per GPU: the instances run as this many independent colonies (engine.StreamedTSP), one per "
                         "stream -- the same tours, costs and pheromone; 1 = one colony over all instances in lock-step (the default: "
                         "two streams gain 4 %% in a sustained loop and lose it in a 10-step timed region, profiles/r04_streams.txt)")
    ap.add_argument("--precondition-seconds", type=float, default=0.5,
                    help="untimed steps of a throw-away colony of the same shape before the W warm-up steps: the first ~20 ms "
                         "of a launch sequence run 5-8 %% slower than the steady state (profiles/r04_headline_clock_ramp.txt)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for the barrier / max-time reduce (nccl = RCCL)")
    ap.add_argument("--shard", default="instances", choices=["instances", "ants"],
                    help="instances: B colonies per GPU, no collective (weak scaling, default); "
                         "ants: the same B colonies on every GPU, A/N ants each, one collective per iteration "
                         "(strong scaling)")
    ap.add_argument("--exchange", default="tours", choices=["tours", "delta"],
                    help="--shard ants: all-gather of the tours (int16; exact, default) or all-reduce of delta-tau")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and take the distributed code path even with one rank (world size 1 on RCCL: "
                         "what an N-GPU run executes, on one GPU)")
    ap.add_argument("--force-device", type=int, default=None,
                    help="testing only: put every rank on this GPU (needs --dist-backend gloo)")
    return apply_config(ap.parse_args())


def apply_config(args):
    if args.config == "c5":
        args.nodes, args.ants, args.batch = 1000, 2048, 64
        args.steps = min(args.steps, 5)
    return args


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)



# ---------------------------------------------------------------------------------------------- the line
HEADLINE_MAX_BYTES = 4096


def _r(v, sig=6):
    """Numbers rounded to `sig` significant digits (the record is read by people and by a size-limited tail parser)."""
    if isinstance(v, float):
        return float(f"{v:.{sig}g}")
    if isinstance(v, dict):
        return {k: _r(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, sig) for x in v]
    return v


def headline_record(full):
    """The LAST stdout line: the contract's keys + roofline + cpu_baseline, numbers only, <= HEADLINE_MAX_BYTES.
    Everything else (per-configuration extras, pipe occupancies, notes) lives in bench_extras.json."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
    rec = {k: full[k] for k in keep if k in full}
    cfg = full.get("config") or {}
    rec["config"] = {k: cfg[k] for k in ("workload", "nodes", "n_ants", "instances_per_gpu", "sampler", "parallelism") if k in cfg}
    rf = full.get("roofline")
    if rf:
        rec["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms")}
        rec["roofline"]["valu_busy"] = rf.get("valu_busy")
        pp = rf.get("pipes") or {}
        # what the counter pass of this kernel says about the CU's units (the head-row kernel is bound by instruction issue and the
        # vector-memory return path, not by the byte rate `frac` is taken against)
        rec["roofline"]["units"] = {k: pp.get(k) for k in ("valu_active", "td_busy", "ta_busy", "lds_busy", "l2_hit_rate") if k in pp} or None
        alg, hbm = rf.get("algorithmic") or {}, rf.get("hbm") or {}
        rec["roofline"]["algorithmic"] = {"GBps": alg.get("GBps"), "over_hbm_peak": alg.get("over_hbm_peak")}
        rec["roofline"]["hbm"] = {"ratio": hbm.get("ratio"), "GBps": hbm.get("GBps"), "peak": hbm.get("peak")}
    else:
        rec["roofline"] = None
    cb = full.get("cpu_baseline")
    if cb:
        rec["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "host_cpus", "kind")}
        rec["cpu_baseline"]["sample"] = str(cb.get("sample", ""))[:400]
    g = full.get("best_cost_gap")
    if g:
        rec["best_cost_gap"] = {k: g.get(k) for k in ("gap", "ci95", "equal_or_better", "gpu_better_or_equal_on", "gpu_mean_best",
                                                      "cpu_mean_best", "instances", "iterations", "gpu_seeds", "sampler") if k in g}
        if isinstance(g.get("samplers"), dict):          # scan_sparse / scan / race: [gap, ci low, ci high, equal_or_better, wins]
            rec["best_cost_gap"]["samplers"] = {k: [v.get("gap")] + list(v.get("ci95") or [None, None]) + [v.get("equal_or_better"),
                                                                                                         v.get("gpu_better_or_equal_on")]
                                                for k, v in g["samplers"].items()}
        if isinstance(g.get("study"), dict) and isinstance(g["study"].get("samplers"), dict):
            # the committed full-power study (64 instances x 3 seeds x 20 iterations), same five numbers per sampler
            st = g["study"]
            rec["best_cost_gap"]["study"] = {"n": [st.get("instances"), st.get("cpu_seeds"), st.get("iterations")], "stored": True,
                                             **{k: [v.get("gap")] + list(v.get("ci95") or [None, None]) + [v.get("equal_or_better"),
                                                                                                       v.get("gpu_better_or_equal_on")]
                                                for k, v in st["samplers"].items()}}
    for k in ("speedup_vs_cpu", "gpu_mean_best_cost"):
        if k in full:
            rec[k] = full[k]
    if full.get("sustained"):
        rec["sustained"] = {k: full["sustained"].get(k) for k in ("seconds", "steps", "value")}
    if full.get("rccl"):
        rec["rccl"] = {k: v for k, v in full["rccl"].items() if not isinstance(v, (dict, list))}
    ex = full.get("extras")
    if isinstance(ex, dict) and isinstance(ex.get("headline_scan_sparse"), dict) and "scan" in ex["headline_scan_sparse"]:
        # the same workload on the dense scan (sampler="scan": whole rows), next to the default's head / tail rows
        d_ = ex["headline_scan_sparse"]["scan"]
        rec["dense_scan"] = {"value": d_.get("value"), "kernel_ms": d_.get("kernel_ms"),
                             "roofline_frac": (d_.get("roofline") or {}).get("frac"), "kernel": "tsp_scan32_kernel"}
    if isinstance(ex, dict):
        # one number per extra configuration; the objects are in the file
        rec["extras"] = {k: (_r(v.get("value"), 4) if isinstance(v, dict) and "value" in v else
                             ("error" if isinstance(v, dict) and "error" in v else None)) for k, v in ex.items()
                         if k != "error"}
        rec["extras_file"] = "bench_extras.json"
    rec = _r(rec)
    txt = json.dumps(rec, allow_nan=False)
    if len(txt) > HEADLINE_MAX_BYTES:                   # drop the optional parts first; the contract's keys never
        for k in ("extras", "sustained", "best_cost_gap"):
            rec.pop(k, None)
            txt = json.dumps(rec, allow_nan=False)
            if len(txt) <= HEADLINE_MAX_BYTES:
                break
    assert len(txt) <= HEADLINE_MAX_BYTES, f"headline record is {len(txt)} bytes"
    return txt


def _finite(v):
    """json.dumps would print NaN / Infinity, which strict parsers reject: they become null."""
    if isinstance(v, float):
        return v if v == v and abs(v) != float("inf") else None
    if isinstance(v, dict):
        return {k: _finite(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_finite(x) for x in v]
    return v


_REAL_STDOUT = None


def quiet_stdout():
    """Everything but the record goes to stderr: file descriptor 1 is pointed at stderr for the rest of the run and the record
    is written to the saved descriptor at the very end.  (RCCL prints a version banner on C stdout, flushed when the process
    exits -- i.e. AFTER a Python print of the record: the driver reads the LAST stdout line.)"""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(full):
    """Full record -> bench_extras.json (repo root, and gpurun_out/ when it exists so that it travels back from a GPU box);
    compact record -> the last stdout line."""
    full = _finite(full)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, "bench_extras.json"), "w") as f:
                    json.dump(full, f, indent=1)
        except OSError as e:
            log(f"could not write bench_extras.json under {d}: {e}")
    sys.stderr.flush()
    sys.stdout.flush()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (headline_record(full) + "\n").encode())


# ---------------------------------------------------------------------------------------------- launcher
def launch_ranks(args):
    """`python bench.py --gpus N` without torchrun: start N copies of this script, one per GPU, wired up through
    the torch.distributed environment variables.  Rank 0's stdout (the JSON line) passes through."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # dmabuf IPC (this image's driver has no legacy IPC: RCCL / tensor sharing across processes fail with
        # hipIpcGetMemHandle otherwise); whatever the caller exported wins
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


# ---------------------------------------------------------------------------------------------- inputs
def make_instances(B, n, seed):
    """coords ~ U[0,1)^2 (tsp/train.ipynb:84), distances with diag 1e9 (tsp/utils.py:4-14)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    coords = torch.rand(B, n, 2, generator=g)
    # the reference's expression (torch.norm of the coordinate differences); torch.cdist's matmul form returns exact zeros
    # for close pairs, i.e. an infinite 1/d
    dist = torch.cat([torch.norm(c[:, None] - c, dim=2, p=2).unsqueeze(0) for c in coords])
    idx = torch.arange(n)
    dist[:, idx, idx] = 1e9
    return dist


def bytes_per_tour(n, A, steps=None):
    """two f32 rows per step, i64 path writes, the update's tau round trip shared by A ants and its
    re-read of the paths."""
    steps = n - 1 if steps is None else steps
    return steps * 8 * n + 8 * n + 8 * n * n / A + 8 * n


def sampler_layout(n, sampler):
    """daco_tsp_sample's layout rule -> (kernel name, floats per fused row as the kernel streams it)."""
    lanes = 64 if sampler != "scan" or n > 1024 else (4 if n <= 128 else 8 if n <= 256 else 32)
    name = {4: "scan16_kernel", 8: "scan16_kernel", 32: "tsp_scan32_kernel", 64: "tsp_sample_kernel"}[lanes]
    if lanes < 64:
        row = (n + 4 * lanes - 1) // (4 * lanes) * (4 * lanes)
    else:
        row = (n + 255) // 256 * 256 if n > 128 else n
    return name, row


def roofline_rows(n, A, B, sampler, kern_ms, steps_per_tour=None, traffic=None, traffic_source=None, pipes=None, head_k=None):
    """Roofline object of a tour-construction launch: L2 row stream as the bound, SURVEY 8(d)'s algorithmic bytes and
    the HBM-side picture next to it; `pipes`: the counter-measured occupancy of the CU's units for this kernel."""
    name, row = sampler_layout(n, sampler)
    steps = n - 1 if steps_per_tour is None else steps_per_tour
    # the bytes a step NEEDS: the n live floats of its fused row (the kernel streams the padded row, `padded_row_floats`;
    # the pad is the kernel's own overhead and does not count as achieved bandwidth)
    row_bytes = B * A * steps * 4.0 * n
    if sampler == "scan_sparse":            # a head step reads 64 / 128 values + ids; the dense steps (counted in the run) a row
        head_row = 384.0 if (head_k or max(1, min(127, n // 10))) <= 63 else 768.0
        name, row_bytes = "scan_sparse_kernel", B * A * steps * head_row
    alg_bytes = B * A * bytes_per_tour(n, A, steps)
    ach = row_bytes / (kern_ms * 1e-3) / 1e9
    compulsory = B * (8.0 * n * n + 8.0 * A * n)
    return {"bound": "l2", "achieved": ach, "peak": PEAK_L2_GBS, "unit": "GB/s", "frac": ach / PEAK_L2_GBS,
            "traffic": traffic, "traffic_source": traffic_source,
            "kernel": name, "kernel_ms": kern_ms, "row_bytes_per_launch": row_bytes, "padded_row_floats": row,
            # valu_busy = SQ_INSTS_VALU x 2 cycles (a wave64 VALU instruction issues over two cycles on gfx950's SIMD-32)
            # / (SIMDs x kernel cycles); ta / td_busy: the CU's vector-memory address and data-return units
            "pipes": pipes, "valu_busy": (pipes or {}).get("valu_busy"),
            "algorithmic": {"bytes_per_launch": alg_bytes, "GBps": alg_bytes / (kern_ms * 1e-3) / 1e9,
                            "over_hbm_peak": alg_bytes / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                            "note": "SURVEY 8(d) bytes / kernel time; above the HBM peak because the rows are "
                                    "L2/MALL-resident and tau^a*eta^b is fused into one row -- not a roofline fraction"},
            "hbm": {"counter_bytes": traffic, "compulsory_bytes": compulsory,
                    "ratio": traffic / compulsory if traffic else None,
                    "GBps": traffic / (kern_ms * 1e-3) / 1e9 if traffic else None, "peak": PEAK_HBM_GBS}}


_WARM = {}


def warm_colony(col, seconds=0.15):
    """Two untimed iterations of the colony, after `seconds` of unrelated device work (a configuration is measured in the steady
    state, not in the first milliseconds after its set-up's idle gap: profiles/r04_headline_clock_ramp.txt).  The colony's
    iteration count stays what the entry says."""
    import torch
    dev = col.pheromone.device
    if dev not in _WARM:
        _WARM[dev] = torch.rand(2048, 2048, device=dev)
    x = _WARM[dev]
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            x @ x
        torch.cuda.synchronize()
    col.step()
    col.step()


def time_launches(fn, steps, warm=2):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


# ---------------------------------------------------------------------------------------------- CPU leg
def _cpu_colony(job):
    """One instance of the workload on the reference's CPU op sequence (a worker process of cpu_baseline)."""
    import torch
    from oracle import torch_port
    d, k_sparse, n_ants, iters, threads, seed, budget_s = job
    torch.set_num_threads(threads)
    torch.manual_seed(seed)
    _, idx = torch.topk(d, k=k_sparse, dim=1, largest=False)
    sparse = torch.full_like(d, 1e10)
    sparse.scatter_(1, idx, torch.gather(d, 1, idx))
    heu = 1 / sparse
    tau = torch.ones_like(d)
    lowest, done = float("inf"), 0
    trace = []                                   # best cost after 1, 2, ... iterations (the gap is taken where every colony got to)
    t0 = time.perf_counter()
    for _ in range(iters):
        paths = torch_port.rollout(tau, heu, n_ants)
        costs = torch_port.tour_lengths(d, paths)
        lowest = min(lowest, float(costs.min()))
        trace.append(lowest)
        tau = torch_port.deposit(tau, paths, costs, 0.9)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    return trace, done, time.perf_counter() - t0


def cpu_baseline(dist_cpu, k_sparse, n_ants, instances, iters, budget_s=110.0, seed0=4321, threads=None):
    """Reference CPU path (torch port): `instances` colonies of the same workload, `iters` iterations each, as
    parallel processes with a few intra-op threads each (torch's default of one thread per logical CPU is far from
    optimal for [512 x 500] tensors on a many-core host).
    Returns (cpu_baseline object, per-instance best costs, iterations done)."""
    import multiprocessing as mp
    import torch
    n = dist_cpu.shape[1]
    ncpu = os.cpu_count() or 1
    # intra-op threads per colony process: the [512 x 500] elementwise ops of a rollout step stop scaling beyond a few
    # threads (one process alone on the 256-CPU box: 2 -> 3.75, 4 -> 3.53, 8 -> 3.48, 16 -> 4.03 ms per step), and with 16
    # colonies side by side fewer threads each is faster still (20 iterations: 88 s at 2 threads, 150 s at 4, > 300 s at 8)
    threads = threads or max(1, min(2, ncpu // max(1, min(instances, dist_cpu.shape[0]))))
    instances = min(instances, dist_cpu.shape[0])
    procs = max(1, min(instances, ncpu // threads))
    jobs = [(dist_cpu[b].clone(), k_sparse, n_ants, iters, threads, seed0 + b, budget_s) for b in range(instances)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(_cpu_colony, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    done = min(r[1] for r in res)
    busy = max(r[2] for r in res)
    tours = sum(r[1] for r in res) * n_ants
    log(f"cpu port: {instances} instances x {done} iterations on {procs} processes x {threads} threads: "
        f"{busy:.1f}s of colony time ({wall:.1f}s with process start-up)")
    out = {"value": tours / busy, "unit": "ant-tours/s", "cores": procs * threads, "host_cpus": ncpu, "kind": "port",
           "sample": f"{instances} instances x {n_ants} ants x {done} colony iterations of the same TSP-{n} workload "
                     f"(oracle/torch_port.py: the reference's aten op sequence, torch {torch.__version__} CPU), "
                     f"{procs} processes x {threads} intra-op threads = {procs * threads} of the host's {ncpu} CPUs, {busy:.1f} s",
           "one_process_value": n_ants * res[0][1] / res[0][2]}
    return out, [r[0][done - 1] for r in res], done


# ---------------------------------------------------------------------------------------------- extras
def _cpu_small(job):
    """CPU legs of the extras (worker process): the reference's op sequence on this host, a bounded sample each."""
    import numpy as np
    import torch
    kind, threads, budget_s = job[0], job[1], job[2]
    torch.set_num_threads(threads)
    from oracle import torch_port
    t0 = time.perf_counter()
    if kind == "tsp":
        d, k_sparse, n_ants, seed = job[3:]
        torch.manual_seed(seed)
        _, idx = torch.topk(d, k=k_sparse, dim=1, largest=False)
        sparse = torch.full_like(d, 1e10)
        sparse.scatter_(1, idx, torch.gather(d, 1, idx))
        heu, tau, its = 1 / sparse, torch.ones_like(d), 0
        while True:
            paths = torch_port.rollout(tau, heu, n_ants)
            costs = torch_port.tour_lengths(d, paths)
            tau = torch_port.deposit(tau, paths, costs, 0.9)
            its += 1
            if time.perf_counter() - t0 > budget_s:
                break
        return its * n_ants, time.perf_counter() - t0
    if kind == "cvrp":
        d, dem, cap, n_ants, seed = job[3:]
        torch.manual_seed(seed)
        heu, tau, its = 1 / d, torch.ones_like(d), 0
        while True:
            paths = torch_port.cvrp_rollout(tau, heu, dem, cap, n_ants)
            costs = torch_port.route_lengths(d, paths)
            tau = torch_port.deposit_directed(tau, paths, costs, 0.9)
            its += 1
            if time.perf_counter() - t0 > budget_s:
                break
        return its * n_ants, time.perf_counter() - t0
    if kind == "nls":
        import oracle
        d, hd, tours, maxt = job[3:]
        done = sweeps = 0
        for i in range(0, tours.shape[0], 2):                       # two tours at a time until the budget is spent
            _, sw = oracle.nls_batch(d, hd, tours[i:i + 2], maxt)
            done += min(2, tours.shape[0] - i)
            sweeps += sw
            if time.perf_counter() - t0 > budget_s:
                break
        return done, time.perf_counter() - t0, sweeps
    if kind == "hgs":
        # the reference's own compiled local search when oracle/_ref/libhgscvrp.so travelled with the repository (HGS built from
        # the reference's sources; called as cvrp_nls/swapstar.py:240-271 does, through /tmp route files, with the parameters HGS
        # reads from the reference's structure: seed 1, no SWAP*), else the C restatement (oracle/hgs_ls.c)
        import ctypes as C
        import oracle
        pos, d, hd, dem, cols, limit, wid = job[3:]
        ref = os.path.join(ROOT, "oracle", "_ref", "libhgscvrp.so")
        lib = None
        if os.path.isfile(ref):
            try:
                lib = C.CDLL(ref)
            except OSError:
                lib = None
        done = 0
        if lib is not None:
            class AP(C.Structure):      # AlgorithmParameters.h:10-28
                _fields_ = [("nbGranular", C.c_int), ("mu", C.c_int), ("lambda_", C.c_int), ("nbElite", C.c_int), ("nbClose", C.c_int),
                            ("nbIterPenaltyManagement", C.c_int), ("targetFeasible", C.c_double), ("penaltyDecrease", C.c_double),
                            ("penaltyIncrease", C.c_double), ("seed", C.c_int), ("nbIter", C.c_int), ("nbIterTraces", C.c_int),
                            ("timeLimit", C.c_double), ("useSwapStar", C.c_int)]
            dp = C.POINTER(C.c_double)
            lib.local_search.argtypes = [C.c_int, dp, dp, dp, dp, dp, C.c_double, C.c_double, C.c_char, C.c_int, C.POINTER(AP), C.c_char,
                                         C.c_int, C.c_int]
            n = len(dem)
            x, y = np.ascontiguousarray(pos[:, 0]), np.ascontiguousarray(pos[:, 1])
            sv, dm = np.zeros(n), np.ascontiguousarray(dem * 1000)
            ap = AP(20, 25, 40, 4, 5, 100, 0.2, 0.85, 1.2, 1, 20000, 500, 0.0, 0)

            def call(mat, routes, count, cid):
                with open(f"/tmp/route-{cid}", "w") as f:
                    for i, r in enumerate(routes):
                        f.write(f"Route #{i + 1}: " + " ".join(map(str, r)) + "\n")
                m = np.ascontiguousarray(mat).reshape(-1)
                lib.local_search(n, x.ctypes.data_as(dp), y.ctypes.data_as(dp), m.ctypes.data_as(dp), sv.ctypes.data_as(dp),
                                 dm.ctypes.data_as(dp), 1000.001, sys.float_info.max, b'\0', len(routes), C.byref(ap), b'\0', cid, count)
                out = []
                with open(f"/tmp/swapstar-result-{cid}") as f:
                    for line in f:
                        if line.startswith("Route"):
                            out.append(list(map(int, line.split(":")[1].split())))
                os.remove(f"/tmp/swapstar-result-{cid}")
                os.remove(f"/tmp/route-{cid}")
                return out
            for a in range(cols.shape[1]):
                seq = cols[:, a].tolist()
                routes, cur = [], []
                for v in seq:
                    if v == 0:
                        if cur:
                            routes.append(cur)
                        cur = []
                    else:
                        cur.append(v)
                cid = 1000003 * (wid + 1) + a
                r = call(d, routes, limit, cid)
                r = call(hd, r, 10, cid)
                call(d, r, limit, cid)
                done += 1
                if time.perf_counter() - t0 > budget_s:
                    break
            return done, time.perf_counter() - t0, "reference"
        for a in range(cols.shape[1]):
            oracle.hgs_neural_swapstar(pos, d, hd, dem, cols[:, a], limit)
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
        return done, time.perf_counter() - t0, "port"
    raise ValueError(kind)


def _cpu_leg(jobs):
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(len(jobs)) as pool:
        return pool.map(_cpu_small, jobs, chunksize=1)


def extra_configs(dev, headline_colony, cpu=True):
    """BASELINE.json's other single-GPU configurations, each one timed end to end (all kernels of an iteration) with a
    roofline object for its dominant kernel (timed by HIP events around that kernel), the counter-measured HBM-side bytes
    where a PMC pass of this library version is committed under profiles/, and a cpu_baseline: the reference's CPU op
    sequence (oracle/torch_port.py, the C restatement of two_opt.py, torch ops for the network) on a bounded sample of
    the same workload on this host."""
    import numpy as np
    import torch
    from deepaco_amd import engine
    out = {}
    ncpu = os.cpu_count() or 1
    traffic = load_traffic()
    counters = load_counters()

    def events(steps):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for a, b in ev:
            a.record(); b.record()
        torch.cuda.synchronize()
        return ev

    def cpu_tsp(d_cpu, k, A, budget):
        if not cpu:
            return None
        procs = min(d_cpu.shape[0], 4, max(1, ncpu // 2))
        res = _cpu_leg([("tsp", 2, budget, d_cpu[b].clone(), k, A, 99 + b) for b in range(procs)])
        busy = max(r[1] for r in res)
        return {"value": sum(r[0] for r in res) / busy, "unit": "ant-tours/s", "cores": 2 * procs, "kind": "port",
                "sample": f"{procs} instances of the same workload side by side, {[r[0] // A for r in res]} colony iterations "
                          f"in {busy:.1f} s (oracle/torch_port.py: the aten op sequence of tsp/aco.py), 2 intra-op threads each"}

    def tsp(tag, n, A, B, k, steps, cpu_budget, head_rows=False):
        d_cpu = make_instances(B, n, 77)

        def colony(sampler):
            col = engine.BatchedTSP(d_cpu.to(dev), n_ants=A, seed=5, sampler=sampler)
            col.sparsify(k)
            col.heuristic = col.heuristic.contiguous()
            warm_colony(col)
            ev = events(steps)
            t0 = time.perf_counter()
            for s in range(steps):
                col.step(events=ev[s])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            kms = sum(a.elapsed_time(b) for a, b in ev) / steps
            col.run(10 - col.iteration)
            return dt, kms, float(col.lowest_cost.mean())

        dt, kms, best = colony("scan")
        sparse = None
        if head_rows:
            # the same colony on HEAD / TAIL rows (sampler "scan_sparse": the same categorical, its own uniform stream; k = 100 -> the
            # 128-slot head), reported next to the dense scan -- not the figure of this entry
            try:
                sdt, skms, sbest = colony("scan_sparse")
                sparse = {"value": B * A / sdt, "unit": "ant-tours/s", "ms_per_step": sdt * 1e3, "kernel_ms": skms,
                          "kernel": "scan_sparse_kernel<4, false, 8> (HIP events around the launch)",
                          "pipes": counters.get(f"tsp{n}_a{A}_b{B}_scan_sparse"),
                          "mean_best_cost_after_10_iterations": sbest, "dense_mean_best_cost_after_10_iterations": best,
                          "speedup_whole_iteration": dt / sdt}
            except Exception as e:
                sparse = {"error": repr(e)}
        tr, src = traffic.get(f"tsp{n}_a{A}_b{B}_scan", (None, None))
        out[tag] = {"workload": f"TSP-{n}, n_ants={A}, {B} instances, AS iteration", "value": B * A / dt,
                    "unit": "ant-tours/s", "ms_per_step": dt * 1e3, "steps": steps,
                    "roofline": roofline_rows(n, A, B, "scan", kms, traffic=tr, traffic_source=src,
                                              pipes=counters.get(f"tsp{n}_a{A}_b{B}_scan")),
                    "cpu_baseline": cpu_tsp(d_cpu, k, A, cpu_budget)}
        if sparse is not None:
            out[tag]["scan_sparse"] = sparse

    tsp("c2_tsp100_a512_b256", 100, 512, 256, 20, 10, 6.0)
    tsp("c5_share_tsp1000_a2048_b64", 1000, 2048, 64, 100, 5, 20.0, head_rows=True)

    def headline_small(tag, B, steps=200):
        """the headline workload (TSP-500, 512 ants, k = 50) at B = 1 -- the reference's own call pattern, one colony
        per instance, tsp/test.ipynb:66-68 -- and B = 8: the colony's default sampler (auto -> head rows; B = 1 keeps them in LDS)."""
        n, A, k = 500, 512, 50
        try:
            col = engine.BatchedTSP(make_instances(B, n, 1234).to(dev), n_ants=A, seed=5, sampler="auto")
            col.sparsify(k)
            col.heuristic = col.heuristic.contiguous()
            warm_colony(col)
            ev = events(steps)
            t0 = time.perf_counter()
            for s_ in range(steps):
                col.step(events=ev[s_])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            kms = sum(a.elapsed_time(b) for a, b in ev) / steps
            out[tag] = {"workload": f"TSP-{n}, n_ants={A}, {B} instance{'s' if B > 1 else ''}, 1/d sparsified k={k}, sampler auto -> "
                                    f"{col.resolved_sampler()[0]}", "value": B * A / dt, "unit": "ant-tours/s", "ms_per_step": dt * 1e3,
                        "steps": steps, "kernel_ms": kms,
                        "roofline": roofline_rows(n, A, B, "scan_sparse", kms, head_k=k)}
        except Exception as e:
            out[tag] = {"error": repr(e)}

    headline_small("headline_b1", 1)
    headline_small("headline_b8", 8)

    # the headline's 64 instances as TWO colonies of 32 on two HIP streams (engine.StreamedTSP: the same tours, costs and pheromone
    # as one colony over all instances, tests/test_gpu_12_streams.py): one colony's update and the tail of its construction launch
    # (2 048 workgroups on 1 536 resident: a third of a round at two per CU) run under the other's construction.  Not the default:
    # `roofline` is defined per launch, and two launches sharing the chip each look slower than they are together.
    try:
        n, A, B, k = 500, 512, 64, 50
        col = engine.StreamedTSP(make_instances(B, n, 1234).to(dev), parts=2, n_ants=A, sampler="auto", seed=1234)
        col.sparsify(k)
        dts = {}
        for wp in (True, False):                  # with the parts' int64 paths tensors (as every round measured it) / compact tours
            for _ in range(8):
                col.step(want_paths=wp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(40):
                col.step(want_paths=wp)
            torch.cuda.synchronize()
            dts[wp] = (time.perf_counter() - t0) / 40
        dt = dts[True]
        out["headline_two_streams"] = {"workload": f"TSP-{n}, n_ants={A}, {B} instances as two colonies on two HIP streams, 1/d sparsified k={k}, sampler auto",
                                       "value": B * A / dt, "unit": "ant-tours/s", "ms_per_step": dt * 1e3, "steps": 40,
                                       "compact_tours": {"value": B * A / dts[False], "ms_per_step": dts[False] * 1e3}}
        del col
    except Exception as e:
        out["headline_two_streams"] = {"error": repr(e)}

    # the headline iteration as BatchedTSP.run() executes it: the tours stay compact (u16 rows in the sampler's workspace, the best
    # one copied from there: step(want_paths=False)) instead of leaving as the int64 [B, n, A] tensor step() returns -- the
    # reference's loop keeps `paths` to itself too (tsp/aco.py:75-92).  Not the metric's default: the line above is the iteration
    # that also hands out the reference's paths tensor.  Same costs, records and pheromone (tests/test_gpu_11_scan_sparse.py).
    try:
        n, A, B, k = 500, 512, 64, 50
        col = engine.BatchedTSP(make_instances(B, n, 1234).to(dev), n_ants=A, seed=1234, sampler="auto")
        col.sparsify(k)
        col.heuristic = col.heuristic.contiguous()
        for _ in range(8):
            col.step(want_paths=False)
        ev = events(40)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s_ in range(40):
            col.step(events=ev[s_], want_paths=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 40
        kms = sum(a.elapsed_time(b) for a, b in ev) / 40
        out["headline_compact_tours"] = {"workload": f"TSP-{n}, n_ants={A}, {B} instances, 1/d sparsified k={k}, sampler auto, the iteration of "
                                                     "run(): tours as u16 rows, no int64 paths tensor",
                                         "value": B * A / dt, "unit": "ant-tours/s", "ms_per_step": dt * 1e3, "steps": 40, "kernel_ms": kms,
                                         "roofline": roofline_rows(n, A, B, "scan_sparse", kms, head_k=k)}
        del col
    except Exception as e:
        out["headline_compact_tours"] = {"error": repr(e)}

    # config 4: CVRP-100, capacity mask in the sampling kernel
    n, A, B = 100, 512, 256
    g = torch.Generator().manual_seed(3)
    loc = torch.cat((torch.full((B, 1, 2), 0.5), torch.rand(B, n, 2, generator=g)), 1)
    dem = torch.cat((torch.zeros(B, 1), torch.randint(1, 10, (B, n), generator=g).float()), 1)
    d = torch.cat([torch.norm(c[:, None] - c, dim=2, p=2).unsqueeze(0) for c in loc])
    i = torch.arange(n + 1)
    d[:, i, i] = 1e-10
    col = engine.BatchedCVRP(d.to(dev), dem.to(dev), n_ants=A, capacity=50, seed=1)
    warm_colony(col)
    steps = 10
    ev = events(steps)
    t0 = time.perf_counter()
    for s_ in range(steps):
        col.step(events=ev[s_])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kms = sum(a.elapsed_time(b) for a, b in ev) / steps
    L = float(col.last_lens.float().mean())
    tr, src = traffic.get("cvrp100_a512_b256_scan", (None, None))
    rf = roofline_rows(n + 1, A, B, "scan", kms, steps_per_tour=L - 1, traffic=tr, traffic_source=src,
                       pipes=counters.get("cvrp100_a512_b256_scan"))
    rf["kernel"] = "scan16_kernel<CVRP> (HIP events around the construction kernel)"
    cb = None
    if cpu:
        procs = min(4, max(1, ncpu // 2))
        res = _cpu_leg([("cvrp", 2, 6.0, d[b].clone(), dem[b].clone(), 50.0, A, 7 + b) for b in range(procs)])
        busy = max(r[1] for r in res)
        cb = {"value": sum(r[0] for r in res) / busy, "unit": "ant-tours/s", "cores": 2 * procs, "kind": "port",
              "sample": f"{procs} instances side by side, {[r[0] // A for r in res]} colony iterations in {busy:.1f} s "
                        f"(oracle/torch_port.py cvrp_rollout / route_lengths / deposit_directed: the aten op sequence of "
                        f"cvrp/aco.py, its per-step all-done sync included), 2 intra-op threads each"}
    out["c4_cvrp100_a512_b256"] = {"workload": f"CVRP-{n} (capacity mask), n_ants={A}, {B} instances, AS iteration",
                                   "value": B * A / dt, "unit": "ant-tours/s", "ms_per_step": dt * 1e3, "steps": steps,
                                   "mean_route_len": L, "roofline": rf, "cpu_baseline": cb}
    del col

    # config 3: TSP-500 + NLS; the local search is 99 % of it: ONE launch of daco_tsp_nls per iteration
    n, A, B = 500, 256, 64
    d_cpu = make_instances(B, n, 2)
    col = engine.BatchedTSP(d_cpu.to(dev), n_ants=A, seed=1, local_search="nls", fixed_start=0)
    col.sparsify(50)
    col.nls_counters = torch.zeros(2, dtype=torch.int64, device=dev)
    col.step()
    torch.cuda.synchronize()
    col.nls_counters.zero_()
    steps = 5
    ev = events(steps)
    t0 = time.perf_counter()
    for s_ in range(steps):
        col.step(ls_events=ev[s_])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kms = sum(a.elapsed_time(b) for a, b in ev) / steps
    sweeps, walked = [float(v) / steps for v in col.nls_counters.tolist()]
    # what a walked list entry costs by definition: its 8-byte table entry and the 4-byte matrix gather of its pair; a sweep
    # also re-reads the (<= n) changed edges' matrix entry and two ranks (8 bytes).  L2 -> L1 moves a 128-byte line for each.
    alg = walked * 12.0 + sweeps * 8.0 * 16
    tr, src = traffic.get("nls500_a256_b64", (None, None))
    lines_req = traffic.get("nls500_a256_b64_l2_line_requests", (None, None))[0]
    cb = None
    if cpu:
        paths, _, _, _ = engine.tsp_sample(col.pheromone[:1], col.heuristic[:1], 16, seed=3, batch=1, fixed_start=0)
        tours = paths[0].T.contiguous().cpu().numpy().astype(np.uint16)
        hd = col._heuristic_dist()[0].cpu().numpy()
        procs = min(8, max(1, ncpu // 2))
        res = _cpu_leg([("nls", 1, 8.0, d_cpu[0].numpy(), hd, tours[2 * r:2 * r + 2], n // 4) for r in range(procs)])
        busy = max(r[1] for r in res)
        cb = {"value": sum(r[0] for r in res) / busy, "unit": "ant-tours/s (local search only)", "cores": procs, "kind": "port",
              "sample": f"{sum(r[0] for r in res)} sampled tours of one instance through the NLS schedule (oracle.nls_batch: the C "
                        f"restatement of tsp_nls/two_opt.py -- the reference runs it numba-compiled in a thread pool -- "
                        f"{sum(r[2] for r in res)} sweeps), {procs} processes, {busy:.1f} s; construction and update not included",
              "sweeps_per_s": sum(r[2] for r in res) / busy}
    out["c3_tsp500_nls_a256_b64"] = {
        "workload": f"TSP-{n} + NLS (T_nls=10, T_p=20, maxt={n // 4}; 21 2-opt searches per tour and iteration, one launch), "
                    f"n_ants={A}, {B} instances",
        "value": B * A / dt, "unit": "ant-tours/s", "ms_per_step": dt * 1e3, "steps": steps,
        "local_search": {"kernel_ms": kms, "sweeps_per_iteration": sweeps, "sweeps_per_s": sweeps / (kms * 1e-3),
                         "list_entries_walked_per_sweep": walked / sweeps if sweeps else None,
                         "reference_pair_evaluations_per_s": sweeps * (n - 1) * (n - 2) / 2 / (kms * 1e-3)},
        # the bound this kernel sits on is the L2's LINE rate: every 8-byte table entry / 4-byte matrix gather moves a 128-byte
        # line (16-32 x inflation, structural: the second access of an entry is a function of the tour, not of the list).  With
        # the counter pass of this workload at hand, achieved = line requests x 128 B / this run's kernel time; without it, the
        # bytes the search needs by definition (12 B per walked entry), which no roofline bounds.
        "roofline": (lambda lines_GBps, alg_GBps: {
            "bound": "l2", "unit": "GB/s", "peak": PEAK_L2_GBS,
            "achieved": lines_GBps if lines_GBps is not None else alg_GBps,
            "frac": (lines_GBps if lines_GBps is not None else alg_GBps) / PEAK_L2_GBS,
            "basis": "L2 line requests x 128 B (TCP_TCC_READ_REQ_sum of the counter pass of this workload and library version, "
                     "profiles/r05_pmc_nls.txt) / this run's kernel time" if lines_GBps is not None else
                     "algorithmic bytes only (no counter pass of this library version in profiles/)",
            "traffic": tr, "traffic_source": src,
            "kernel": "nls_kernel (daco_tsp_nls; HIP events around the launch)", "kernel_ms": kms,
            "pipes": counters.get("nls500_a256_b64"), "valu_busy": (counters.get("nls500_a256_b64") or {}).get("valu_busy"),
            "algorithmic": {"bytes_per_launch": alg, "GBps": alg_GBps, "frac_of_l2": alg_GBps / PEAK_L2_GBS,
                            "note": "12 B per walked list entry (table entry + matrix gather) + 128 B per sweep for the changed "
                                    "edges, from the in-run counters"},
            "l2_line_requests_per_launch": lines_req,
        })(None if lines_req is None else lines_req * 128 / (kms * 1e-3) / 1e9, alg / (kms * 1e-3) / 1e9),
        "cpu_baseline": cb}
    del col

    # headline workload in the reference's exponential-race arithmetic (what the bit-exact fixtures pin) -- the price of it
    try:
        n, A, B = 500, 512, 64
        col = engine.BatchedTSP(headline_colony.distances, n_ants=A, sampler="race", seed=11)
        col.sparsify(max(5, n // 10))                      # (the head rows of daco_tsp_sample_race_head: the dense race's tours)
        warm_colony(col)
        dtr = time_launches(col.step, 5, warm=0)
        col.head_k = None                                  # the same colony kept on the dense race kernel
        dtr_dense = time_launches(col.step, 3, warm=1)
        del col
        Bn = 4                                                    # recorded-noise mode: q [B, n-1, A, n] f32 = 0.5 GB per instance
        noise = torch.empty((Bn, n - 1, A, n), device=dev).exponential_(1)
        tau, eta = headline_colony.pheromone[:Bn], headline_colony.heuristic[:Bn]
        dtn = time_launches(lambda: engine.tsp_sample(tau, eta, A, mode="race_noise", noise=noise, batch=Bn), 3, warm=1)
        del noise
        out["headline_parity_modes"] = {
            "workload": f"TSP-{n}, n_ants={A}: the draw as torch.multinomial makes it (argmax of p / q, q ~ Exp(1))",
            "race_philox": {"value": B * A / dtr, "unit": "ant-tours/s", "ms_per_step": dtr * 1e3, "instances": B,
                            "note": "in-kernel Philox noise, whole iteration; drawn from the head rows (64 variates per step, the "
                                    "dense race for an ant whenever a tail candidate could still win): bit-identical to the dense race",
                            "dense_kernel": {"value": B * A / dtr_dense, "ms_per_step": dtr_dense * 1e3},
                            "roofline": (lambda pc: {
                                "bound": "valu", "unit": "wave-instructions/s", "pipes": pc, "valu_busy": (pc or {}).get("valu_busy"),
                                "achieved": None if not pc else pc.get("valu_insts_per_launch", 0) / (dtr_dense),
                                "peak": 1024 * 2.4e9 / 2, "frac": (pc or {}).get("valu_busy"), "traffic": None,
                                "note": "the DENSE race kernel: one Philox4x32-10 block per four candidates and a degree-8 log polynomial per candidate, both "
                                        "fixed by the specification (bit-identical to the oracle): ~50 VALU instructions per candidate-"
                                        "lane; peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction; "
                                        "profiles/r04_pmc_race.txt"})(counters.get("tsp500_a512_b64_race"))},
            "race_noise": {"value": Bn * A / dtn, "unit": "ant-tours/s (construction only)", "ms_per_launch": dtn * 1e3,
                           "instances": Bn,
                           "note": "noise read from memory (the mode the reference-recorded fixtures are replayed in): "
                                   f"{4.0 * (n - 1) * n * A * Bn / 1e9:.1f} GB of q per launch"}}
    except Exception as e:
        out["headline_parity_modes"] = {"error": repr(e)}

    # headline workload on HEAD / TAIL rows (sampler "scan_sparse", include/deepaco_hip.h daco_tsp_sample_sparse): the same
    # distribution, 384 bytes per step while a row's k live entries last.  Its own uniform stream, so it is reported here
    # and the headline stays on the dense scan; best costs of the two samplers side by side on the same instances.
    try:
        n, A, B, k = 500, 512, 64, 50
        res = {}
        for tag in ("scan_sparse", "scan"):
            col = engine.BatchedTSP(headline_colony.distances, n_ants=A, sampler=tag, seed=21)
            col.sparsify(k)
            col.heuristic = col.heuristic.contiguous()
            warm_colony(col)
            ev = events(10)
            t0 = time.perf_counter()
            for s_ in range(10):
                col.step(events=ev[s_])
            torch.cuda.synchronize()
            dts = (time.perf_counter() - t0) / 10
            kms = sum(a.elapsed_time(b) for a, b in ev) / 10
            col.run(20 - col.iteration)
            res[tag] = {"value": B * A / dts, "unit": "ant-tours/s", "ms_per_step": dts * 1e3, "kernel_ms": kms,
                        "mean_best_cost_after_20_iterations": float(col.lowest_cost.mean())}
            if tag == "scan_sparse":
                _, _, _, _, st = engine.tsp_sample_sparse(col.pheromone, col.heuristic, A, col._head_table(), seed=1, batch=B,
                                                          want_stats=True, want_paths=True)
                st = st.tolist()
                res[tag]["steps_dense_tailwalk_rejected"] = st
                pc = counters.get("tsp500_a512_b64_scan_sparse")
                res[tag]["roofline"] = {
                    "bound": "l2", "achieved": B * A * (n - 1) * 384.0 / (kms * 1e-3) / 1e9, "peak": PEAK_L2_GBS, "unit": "GB/s",
                    "frac": B * A * (n - 1) * 384.0 / (kms * 1e-3) / 1e9 / PEAK_L2_GBS, "traffic": None,
                    "kernel": "scan_sparse_kernel<2, false, 4> (HIP events around the launch)", "kernel_ms": kms, "pipes": pc,
                    "valu_busy": (pc or {}).get("valu_busy"),
                    "note": "384 B per head step (64 values + 64 ids) over the kernel time against the L2 rate: the kernel is not "
                            "bound by it but by instruction issue (profiles/r04_scan_sparse_ablation.txt: without any memory "
                            "access the first version still took 0.72 of its time; 93 -> 35 VALU per wave-step bought 0.77 -> "
                            "0.58 ms); profiles/r04_pmc_scan_sparse.txt"}
            else:
                tr_, src_ = traffic.get(f"tsp{n}_a{A}_b{B}_scan", (None, None))
                res[tag]["roofline"] = roofline_rows(n, A, B, "scan", kms, traffic=tr_, traffic_source=src_,
                                                     pipes=counters.get(f"tsp{n}_a{A}_b{B}_scan"))
            del col
        res["speedup_whole_iteration"] = res["scan_sparse"]["value"] / res["scan"]["value"]
        out["headline_scan_sparse"] = {"workload": f"TSP-{n}, n_ants={A}, {B} instances, 1/d sparsified k={k}: sampler "
                                                   f"scan_sparse (head of {k} per row) next to the dense scan", **res}
    except Exception as e:
        out["headline_scan_sparse"] = {"error": repr(e)}

    # CVRP local search (cvrp_nls/aco.py:114-126, 443-448 through csrc/daco_hgs_ls.hip: the reference's routes, entry for entry):
    # CVRP-100, 512 ants, 64 instances, neural_swapstar's three stages (limit = max(n, 50) loops, 10 on the heuristic-derived
    # matrix, limit) in one launch
    try:
        n, A, B = 100, 512, 64
        g = torch.Generator().manual_seed(3)
        loc = torch.cat((torch.full((B, 1, 2), 0.5, dtype=torch.double), torch.rand(B, n, 2, generator=g, dtype=torch.double)), 1)
        dem = torch.cat((torch.zeros(B, 1, dtype=torch.double), torch.randint(1, 10, (B, n), generator=g).double() / 50.0), 1)
        dls = (loc[:, :, None] - loc[:, None]).norm(dim=-1)
        ii = torch.arange(n + 1)
        dls[:, ii, ii] = 1e-10
        heu = 1 / dls
        hdl = 1 / (heu / heu.amax(dim=-1, keepdim=True) + 1e-5)
        dls_d, hdl_d, dem_d = dls.to(dev), hdl.to(dev), dem.to(dev)
        col = engine.BatchedCVRP(dls_d.float(), dem_d, n_ants=A, capacity=1.0, seed=1)
        paths, costs0 = col.step(trim=True)
        limit = max(n + 1, 50)
        td, th = engine.HgsTables(dls_d), engine.HgsTables(hdl_d)
        stages = [(td, limit), (th, 10), (td, limit)]
        engine.hgs_local_search_(paths.clone(), stages, dem_d)
        torch.cuda.synchronize()
        wk = paths.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        _, st, stats = engine.hgs_local_search_(wk, stages, dem_d, want_stats=True)
        e1.record()
        torch.cuda.synchronize()
        dtl = time.perf_counter() - t0
        kms = e0.elapsed_time(e1)
        c1 = engine.tour_costs(dls_d.float(), wk, closed=False)
        moves, loops, rounds = [float(stats[..., k].float().mean()) for k in range(3)]
        pc = counters.get("hgs_ls_100_a512_b64")
        cb = None
        if cpu:
            procs = min(16, max(1, ncpu // 2))
            pn = paths.cpu().numpy()
            res = _cpu_leg([("hgs", 1, 8.0, loc[r % B].numpy(), dls[r % B].numpy(), hdl[r % B].numpy(), dem[r % B].numpy(),
                             pn[r % B], limit, r) for r in range(procs)])
            busy = max(r[1] for r in res)
            kind = res[0][2]
            cb = {"value": sum(r[0] for r in res) / busy, "unit": "solutions/s", "cores": procs, "host_cpus": ncpu, "kind": kind,
                  "sample": f"{sum(r[0] for r in res)} of the same sampled solutions through neural_swapstar's three local_search calls, "
                            f"{procs} processes (one thread each, the reference runs one task per ant in a thread pool), {busy:.1f} s: "
                            + ("HGS-CVRP built from the reference's sources (oracle/_ref/libhgscvrp.so), called through the /tmp route "
                               "files as cvrp_nls/swapstar.py:240-271 does" if kind == "reference" else
                               "oracle/hgs_ls.c, the C restatement of HGS's LocalSearch (oracle/_ref did not travel)")}
        # every evaluation round is ~450 wave instructions, most of them float64 (4 cycles on a SIMD): the kernel is bound by
        # VALU issue and the latency of the dependent L2 gathers between rounds, not by bytes
        out["cvrp_local_search_100_a512_b64"] = {
            "workload": f"CVRP-{n} local search, route for route with the reference (HGS moves 1-9, granular 20, first improvement), "
                        f"{B} x {A} sampled solutions, neural_swapstar's three stages in one launch",
            "value": B * A / dtl, "unit": "solutions/s", "seconds": dtl, "kernel_ms": kms,
            "moves_per_solution": moves, "loops_per_solution": loops, "evaluation_rounds_per_solution": rounds,
            "mean_cost_before": float(costs0.mean()), "mean_cost_after": float(c1.mean()),
            "status_nonzero": int((st != 0).sum()),
            "roofline": {"bound": "valu", "unit": "SIMD-cycles/s", "achieved": None if not pc else pc.get("valu_active_cycles", 0) / (kms * 1e-3),
                         "peak": 1024 * 2.4e9, "frac": (pc or {}).get("valu_active"), "traffic": (pc or {}).get("hbm_bytes"),
                         "kernel": "hgs_ls_kernel<4, 3> (HIP events around the launch)", "kernel_ms": kms, "pipes": pc,
                         "note": "frac = SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x kernel cycles) from the counter pass of this "
                                 "workload (profiles/r05_pmc_hgs_ls.txt); the wavefronts spend the other half parked on the L2 gathers "
                                 "of the next round (one wavefront per solution, a chain of ~1 900 dependent rounds)"},
            "cpu_baseline": cb}
        del col
    except Exception as e:
        out["cvrp_local_search_100_a512_b64"] = {"error": repr(e)}

    # configuration 4's colonies with cvrp_nls's local search in the loop (cvrp_nls/aco.py:134-171: the 8 cheapest ants of every
    # instance through neural_swapstar each iteration): engine.BatchedCVRP(local_search="hgs")
    try:
        n, A, B = 100, 512, 256
        g = torch.Generator().manual_seed(3)
        loc = torch.cat((torch.full((B, 1, 2), 0.5, dtype=torch.double), torch.rand(B, n, 2, generator=g, dtype=torch.double)), 1)
        dem = torch.cat((torch.zeros(B, 1, dtype=torch.double), torch.randint(1, 10, (B, n), generator=g).double()), 1)
        dl = (loc[:, :, None] - loc[:, None]).norm(dim=-1)
        dl[:, torch.arange(n + 1), torch.arange(n + 1)] = 1e-10
        col = engine.BatchedCVRP(dl.to(dev), dem.to(dev), n_ants=A, capacity=50, seed=1, local_search="hgs")
        plain = engine.BatchedCVRP(dl.to(dev), dem.to(dev), n_ants=A, capacity=50, seed=1)
        col.step()
        plain.step()
        dtl = time_launches(col.step, 5, warm=1)
        plain.run(6)
        out["c4_cvrp100_nls_a512_b256"] = {
            "workload": f"CVRP-{n}, n_ants={A}, {B} instances, the 8 cheapest ants of every instance through neural_swapstar each iteration "
                        f"(cvrp_nls/aco.py:134-171), AS iteration", "value": B * A / dtl, "unit": "ant-tours/s", "ms_per_step": dtl * 1e3,
            "local_search_solutions_per_iteration": 8 * B,
            "mean_best_cost_after_7_iterations": float(col.lowest_cost.mean()),
            "without_local_search_mean_best_cost_after_7_iterations": float(plain.lowest_cost.mean()),
            "note": "the local search of an iteration is 2 048 solutions = one round of wavefronts: its time is a single solution's chain "
                    "(DESIGN 3.8b: 3.7 us per round), not the kernel's throughput"}
        del col, plain
    except Exception as e:
        out["c4_cvrp100_nls_a512_b256"] = {"error": repr(e)}

    # headline workload with the LEARNED heuristic: Net + the reference's pretrained tsp500 weights
    # (tests/golden/w_tsp_tsp500.npz: the checkpoint as plain arrays), heu + 1e-10, next to the vanilla 1/d on the
    # same instances and seeds
    try:
        from deepaco_amd.tsp.net import Net as TspNet
        wz = np.load(os.path.join(ROOT, "tests", "golden", "w_tsp_tsp500.npz"))
        lnet = TspNet()
        lnet.load_state_dict({k[3:]: torch.from_numpy(wz[k]) for k in wz.files}, strict=False)
        lnet = lnet.to(dev).eval()
        n, A, B, k = 500, 512, 64, 50
        g = torch.Generator().manual_seed(4242)
        coords = torch.rand(B, n, 2, generator=g).to(dev)
        dist, ei, ea = engine.tsp_knn_graph(coords, k)
        t_net = []
        for _ in range(3):                                   # first call: workspace and CSR set-up, then steady state
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad():
                heu = lnet.reshape_batch(n, ei, lnet.forward_batch(coords, ei, ea, k_sparse=k), eps=1e-10)
            torch.cuda.synchronize()
            t_net.append((time.perf_counter() - t0) * 1e3)
        res = {}
        # learned_on_head_rows: the colony's DEFAULT (sampler="auto": the network's heuristic is k-sparse, so the draws run on head /
        # tail rows); learned_dense_scan: the same colony forced onto whole rows
        for tag, kw in (("learned_on_head_rows", dict(heuristic=heu)), ("learned_dense_scan", dict(heuristic=heu, sampler="scan")),
                        ("vanilla_1_over_d_sparsified_dense_scan", dict(sampler="scan"))):
            col = engine.BatchedTSP(dist, n_ants=A, seed=7, **kw)
            if "heuristic" not in kw:
                col.sparsify(k)
            col.heuristic = col.heuristic.contiguous()
            warm_colony(col)
            ev = events(10)
            t0 = time.perf_counter()
            for s_ in range(10):
                col.step(events=ev[s_])
            torch.cuda.synchronize()
            dtl = (time.perf_counter() - t0) / 10
            kms = sum(a.elapsed_time(b) for a, b in ev) / 10
            col2 = engine.BatchedTSP(dist, n_ants=A, seed=7, **kw)
            if "heuristic" not in kw:
                col2.sparsify(k)
            best = []
            for T in (1, 10, 20):
                col2.run(T - col2.iteration)
                best.append(float(col2.lowest_cost.mean()))
            res[tag] = {"value": B * A / dtl, "unit": "ant-tours/s", "ms_per_step": dtl * 1e3, "kernel_ms": kms,
                        "sampler": list(col.resolved_sampler()), "mean_best_cost_after_1_10_20_iterations": best}
        out["headline_learned_heuristic"] = {
            "workload": f"TSP-{n}, n_ants={A}, {B} instances, heuristic = Net(pretrained tsp500) + 1e-10 vs 1/d sparsified k={k}",
            "gnn_forward_plus_reshape_ms_for_the_batch": {"first_call": t_net[0], "steady": min(t_net[1:])}, **res}
        del lnet
    except Exception as e:
        out["headline_learned_heuristic"] = {"error": repr(e)}

    # config 5's LEARNED variant: TSP-1000, 2048 ants, heuristic = Net(pretrained tsp_nls/tsp1000.pt, start node 0)
    # + 1e-10 on the k = 100 nearest-neighbour graph; 8 instances (a share of the per-GPU 64: the n x n heuristic is dense)
    try:
        from deepaco_amd.tsp_nls.net import Net as NlsNet
        wz = np.load(os.path.join(ROOT, "tests", "golden", "w_tsp_nls_tsp1000.npz"))
        lnet = NlsNet()
        lnet.load_state_dict({k[3:]: torch.from_numpy(wz[k]) for k in wz.files}, strict=False)
        lnet = lnet.to(dev).eval()
        n, A, B, k = 1000, 2048, 8, 100
        g = torch.Generator().manual_seed(1000)
        coords = torch.rand(B, n, 2, generator=g).to(dev)
        dist, ei, ea = engine.tsp_knn_graph(coords, k)
        x = torch.zeros((B, n, 1), device=dev)
        x[:, 0, 0] = 1.0                                       # tsp_nls/utils.py:30-33: the start node's one-hot
        with torch.no_grad():
            heu = lnet.reshape_batch(n, ei, lnet.forward_batch(x, ei, ea, k_sparse=k), eps=1e-10)
        res = {}
        # (learned_on_head_rows: the same heuristic through sampler "scan_sparse" -- the k live entries of a row are its head)
        for tag, kw in (("learned_dense_scan", dict(heuristic=heu, sampler="scan")), ("learned_on_head_rows", dict(heuristic=heu)),
                        ("vanilla_1_over_d_sparsified_dense_scan", dict(sampler="scan"))):
            col = engine.BatchedTSP(dist, n_ants=A, seed=7, fixed_start=0, **kw)
            if "heuristic" not in kw:
                col.sparsify(k)
            col.heuristic = col.heuristic.contiguous()
            col.step()
            dtl = time_launches(col.step, 5, warm=1)
            col.run(10 - col.iteration)
            res[tag] = {"value": B * A / dtl, "unit": "ant-tours/s", "ms_per_step": dtl * 1e3,
                        "mean_best_cost_after_10_iterations": float(col.lowest_cost.mean())}
            res[tag]["sampler"] = list(col.resolved_sampler())
            if col.resolved_sampler()[0] == "scan_sparse":
                st = engine.tsp_sample_sparse(col.pheromone, col.heuristic, A, col._head_table(col.resolved_sampler()[1]), seed=1,
                                              batch=B, fixed_start=0, want_stats=True)[4].tolist()
                res[tag]["steps_dense_tailwalk_rejected_of"] = st + [B * A * (n - 1)]
            del col
        out["c5_learned_tsp1000_a2048_b8"] = {
            "workload": f"TSP-{n}, n_ants={A}, {B} instances, start node 0, heuristic = Net(pretrained tsp_nls/tsp1000) + 1e-10 vs 1/d "
                        f"sparsified k={k} (construction + update, no local search)", **res}
        del lnet, heu
    except Exception as e:
        out["c5_learned_tsp1000_a2048_b8"] = {"error": repr(e)}

    # one optimisation step of tsp_nls/train.py:15-44 for the reference's training batch (20 instances of TSP-100, 30 ants,
    # k = 10: tsp_nls/train.py:95-100), all on the device: pipeline.train_tsp_nls_batch; stages timed with events
    try:
        from deepaco_amd.pipeline import train_tsp_nls_batch, W_2OPT, EPS
        from deepaco_amd.tsp_nls.net import Net as TrainNet
        from deepaco_amd.autograd import TspBatchSampleFn
        Bt, nt, At, kt = 20, 100, 30, 10
        torch.manual_seed(0)
        tnet = TrainNet().to(dev)
        opt = torch.optim.AdamW(tnet.parameters(), lr=3e-4)
        for s_ in range(3):
            train_tsp_nls_batch(tnet, opt, torch.rand(Bt, nt, 2, device=dev), At, kt, seed=1, it=s_)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s_ in range(10):
            train_tsp_nls_batch(tnet, opt, torch.rand(Bt, nt, 2, device=dev), At, kt, seed=1, it=3 + s_)
        torch.cuda.synchronize()
        dts = (time.perf_counter() - t0) / 10
        # the same step with an event between its stages (pipeline.train_tsp_nls_batch, unrolled)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        stage_ms = [0.0] * 5
        for s_ in range(5):
            coords_t = torch.rand(Bt, nt, 2, device=dev)
            tnet.train()
            marks[0].record()
            dist_t, ei_t, ea_t = engine.tsp_knn_graph(coords_t, kt)
            xt = torch.zeros((Bt, nt, 1), device=dev)
            xt[:, 0] = 1.0
            heu_t = tnet.forward_batch_train(xt, ei_t, ea_t, k_sparse=kt)
            hm = tnet.reshape_batch(nt, ei_t, heu_t) + EPS
            marks[1].record()
            paths_t, logp_t, _ = TspBatchSampleFn.apply(hm, torch.ones((Bt, nt, nt), device=dev), At, 1.0, 1.0, "scan", 2, 0, 1, 20 + s_)
            marks[2].record()
            with torch.no_grad():
                c_t = engine.tour_costs(dist_t, paths_t)
                tours_t = paths_t.permute(0, 2, 1).to(torch.int16).contiguous()
                hd_t = (1 / (hm.detach() / hm.detach().amax(dim=-1, keepdim=True) + 1e-5)).contiguous()
                tours_t = engine.nls_(dist_t, hd_t, tours_t, nt // 4, dist_t="symmetric")
                cl_t = engine.tour_costs(dist_t, tours_t.permute(0, 2, 1).to(torch.int64).contiguous())
                adv = (cl_t - cl_t.mean(dim=1, keepdim=True)) * W_2OPT + (c_t - c_t.mean(dim=1, keepdim=True)) * (1 - W_2OPT)
            marks[3].record()
            loss_t = torch.sum(adv.unsqueeze(1) * logp_t) / At / Bt
            opt.zero_grad()
            loss_t.backward()
            marks[4].record()
            torch.nn.utils.clip_grad_norm_(parameters=tnet.parameters(), max_norm=3.0, norm_type=2)
            opt.step()
            marks[5].record()
            torch.cuda.synchronize()
            for j in range(5):
                stage_ms[j] += marks[j].elapsed_time(marks[j + 1]) / 5
        cb = None
        if cpu:
            # the reference's train_instance on the host: its module tree in training mode + the aten op sequence of the rollout
            # with log-probabilities + 2-opt / NLS through the C restatement, backward and optimizer step; one instance after the other
            from oracle import torch_port
            import oracle
            cnet = TrainNet()
            cnet.load_state_dict(tnet.state_dict())
            cnet.train()
            copt = torch.optim.AdamW(cnet.parameters(), lr=3e-4)
            torch.set_num_threads(min(8, ncpu))
            t0 = time.perf_counter()
            done_i = 0
            while time.perf_counter() - t0 < 8.0 and done_i < Bt:
                cc = torch.rand(nt, 2)
                dd = (cc[:, None] - cc).norm(dim=-1)
                dd[torch.arange(nt), torch.arange(nt)] = 1e9
                tk = torch.topk(dd, k=kt, dim=1, largest=False)
                eidx = torch.stack((torch.repeat_interleave(torch.arange(nt), kt), tk.indices.flatten()))
                xx = torch.zeros(nt, 1)
                xx[0] = 1.0
                hv = cnet.par_net_heu(cnet.emb_net(xx, eidx, tk.values.reshape(-1, 1)))
                hmat = torch.zeros(nt, nt)
                hmat[eidx[0], eidx[1]] = hv
                hmat = hmat + 1e-10
                pth, lps = torch_port.rollout(torch.ones(nt, nt), hmat, At, require_prob=True)
                cst = torch_port.tour_lengths(dd, pth)
                hdn = (1 / (hmat.detach() / hmat.detach().amax(dim=-1, keepdim=True) + 1e-5)).numpy()
                tl, _ = oracle.nls_batch(dd.numpy(), hdn, pth.T.numpy().astype(np.uint16), nt // 4)
                cls_ = torch_port.tour_lengths(dd, torch.from_numpy(tl.T.astype(np.int64)))
                advc = (cls_ - cls_.mean()) * W_2OPT + (cst - cst.mean()) * (1 - W_2OPT)
                lossc = torch.sum(advc * lps.sum(dim=0)) / At
                copt.zero_grad()
                lossc.backward()
                torch.nn.utils.clip_grad_norm_(parameters=cnet.parameters(), max_norm=3.0, norm_type=2)
                copt.step()
                done_i += 1
            busy = time.perf_counter() - t0
            cb = {"value": done_i / busy, "unit": "instances/s", "cores": min(8, ncpu), "host_cpus": ncpu, "kind": "port",
                  "sample": f"{done_i} instances, one after the other as tsp_nls/train.py:52-60 runs them, {busy:.1f} s: the module tree of "
                            f"tsp_nls/net.py as torch CPU ops with autograd, oracle/torch_port.rollout with log-probabilities, the NLS through "
                            f"the C restatement of two_opt.py (the reference: numba), AdamW; {min(8, ncpu)} intra-op threads"}
        # round 6: the same step on the flat parameter block (Net.flatten_parameters: no torch.cat of ~100 tensors and no split of
        # the flat gradient per step) eagerly, and as ONE captured HIP graph (pipeline.TspNlsTrainer) -- the value of this entry
        from deepaco_amd.pipeline import TspNlsTrainer

        def time_trainer(Bq, nq, Aq, kq, graph, steps_q):
            torch.manual_seed(0)
            tr = TspNlsTrainer(TrainNet().to(dev), Bq, nq, Aq, kq, lr=3e-4, seed=1, graph=graph)
            cs = [torch.rand(Bq, nq, 2, device=dev) for _ in range(4)]
            for s_ in range(4):
                tr.step(cs[s_])
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for s_ in range(steps_q):
                last = tr.step(cs[s_ % 4])
            torch.cuda.synchronize()
            dt_ = (time.perf_counter() - t0_) / steps_q
            ok = bool(all(torch.isfinite(p_).all() for p_ in tr.net.parameters()))
            return {"ms_per_step": dt_ * 1e3, "instances_per_s": Bq / dt_, "loss": float(last[0]), "mean_cost": float(last[1]),
                    "mean_cost_after_nls": float(last[2]), "parameters_finite": ok}
        flat_e = time_trainer(Bt, nt, At, kt, False, 20)
        flat_g = time_trainer(Bt, nt, At, kt, True, 40)
        out["train_step_tsp100_b20_a30"] = {
            "workload": f"one optimisation step of tsp_nls/train.py: {Bt} instances of TSP-{nt}, {At} ants, k = {kt}, NLS, AdamW",
            "value": flat_g["instances_per_s"], "unit": "instances/s", "ms_per_step": flat_g["ms_per_step"],
            "captured_hip_graph": flat_g, "eager_flat_block": flat_e,
            "eager_parameter_list": {"ms_per_step": dts * 1e3, "instances_per_s": Bt / dts},
            "stage_ms_eager_parameter_list": {"graph_and_network_forward": stage_ms[0], "construction_with_log_probs": stage_ms[1],
                                              "costs_and_local_search": stage_ms[2], "loss_and_backward": stage_ms[3],
                                              "clip_and_optimizer": stage_ms[4]},
            "note": "value = the step as one captured HIP graph on the flat parameter block (pipeline.TspNlsTrainer); the eager step on "
                    "the parameter list is what round 5 measured.  profiles/r06_kernel_stats_train_graph_tsp100.csv lists a step's kernels "
                    "(~330 launches: the NLS kernel a third of the time, the twelve layers' backward kernels most of the rest)",
            "cpu_baseline": cb}
        del tnet, opt
        # tsp/train.ipynb's size (TSP-500, 50 ants, k = 50) with the NLS as local search, 8 instances per step
        try:
            out["train_step_tsp500_b8_a50"] = {
                "workload": "one optimisation step: 8 instances of TSP-500, 50 ants, k = 50 (tsp/train.ipynb:245-251's size), NLS, AdamW",
                "unit": "instances/s", "captured_hip_graph": time_trainer(8, 500, 50, 50, True, 10),
                "eager_flat_block": time_trainer(8, 500, 50, 50, False, 10)}
            out["train_step_tsp500_b8_a50"]["value"] = out["train_step_tsp500_b8_a50"]["captured_hip_graph"]["instances_per_s"]
            out["train_step_tsp500_b8_a50"]["ms_per_step"] = out["train_step_tsp500_b8_a50"]["captured_hip_graph"]["ms_per_step"]
        except Exception as e:
            out["train_step_tsp500_b8_a50"] = {"error": repr(e)}
    except Exception as e:
        out["train_step_tsp100_b20_a30"] = {"error": repr(e)}

    # the six sibling problems (op, pctsp, sop, smtwtp, bpp, mkp: */aco.py gen_sol / gen_path), 512 ants, n = 100: one fused launch
    # per construction against the draw-by-draw service (torch mask bookkeeping + daco_pick_move, the reference's own structure)
    try:
        import subprocess
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "measure_siblings.py")], capture_output=True, text=True, timeout=300)
        rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
        out["siblings_n100_a512"] = {
            "workload": "solution construction of the sibling problems, n = 100, 512 ants, one instance (tools/measure_siblings.py)",
            "unit": "solutions/s", "problems": {x["problem"]: {"fused_solutions_per_s": x["fused_solutions_per_s"], "fused_ms": x["fused_ms"],
                                                              "stepwise_ms": x["stepwise_ms"]} for x in rows},
            "value": sum(x["fused_solutions_per_s"] for x in rows) / max(1, len(rows)),
            "note": "no CPU leg: the oracle restates the siblings' draws (fixtures s_*), not their Python drivers; the step-wise column is "
                    "the reference's call structure (one pick_move per step, masks as torch ops) on this GPU"}
    except Exception as e:
        out["siblings_n100_a512"] = {"error": repr(e)}

    # the reference's own call patterns through the drop-in classes (one instance per colony, its harness' ant counts: tsp/test.ipynb,
    # tsp_nls/test.py, cvrp/test.py), ms per ACO iteration -- tools/time_reference_calls.py
    try:
        import subprocess
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_reference_calls.py"), "50"], capture_output=True, text=True, timeout=300)
        rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
        out["reference_call_patterns"] = {
            "workload": "ACO.run through the drop-in classes, ONE instance per colony with the ant counts of the reference's harnesses "
                        "(tools/time_reference_calls.py: heuristic = sparsified 1/d; tsp_nls with its inference NLS)",
            "unit": "ms per ACO iteration", "rows": rows}
    except Exception as e:
        out["reference_call_patterns"] = {"error": repr(e)}

    # GNN forward (eval), 64 graphs of TSP-500 k=50 side by side
    from deepaco_amd.tsp.net import Net
    torch.manual_seed(0)
    net = Net().to(dev).eval()
    n, k, B = 500, 50, 64
    coords = torch.rand(B, n, 2, device=dev)
    _, ei, ea = engine.tsp_knn_graph(coords, k, want_dist=False)
    with torch.no_grad():
        dt = time_launches(lambda: net.forward_batch(coords, ei, ea, k_sparse=k), 10)
    E = n * k
    per_layer = 2.0 * E * 32 * 4 + 6.0 * n * 32 * 4 + 20e3
    alg = B * 12 * per_layer
    flops = B * 12 * 2.0 * 32 * 32 * (4 * n + E)
    tr, src = traffic.get("gnn_tsp500_k50_b64", (None, None))
    cb = None
    if cpu:
        # the network's own module tree evaluated with torch ops on the host (the aten op sequence of tsp/net.py:27-45),
        # one graph after the other as the reference does
        cnet = Net().eval()
        cnet.load_state_dict(net.state_dict())
        cx, cei, cea = coords.cpu(), ei.cpu(), ea.cpu()
        torch.set_num_threads(min(32, ncpu))
        t0 = time.perf_counter()
        graphs = 0
        with torch.no_grad():
            while time.perf_counter() - t0 < 6.0 and graphs < B:
                cnet.par_net_heu(cnet.emb_net(cx[graphs], cei[graphs], cea[graphs].view(-1, 1)))
                graphs += 1
        busy = time.perf_counter() - t0
        cb = {"value": graphs / busy, "unit": "graphs/s", "cores": min(32, ncpu), "kind": "port",
              "sample": f"{graphs} of the same graphs, one after the other, {busy:.1f} s: the module tree of tsp/net.py as torch "
                        f"CPU ops (torch {torch.__version__}, {min(32, ncpu)} intra-op threads; torch_geometric's mean "
                        f"pooling as index_add / bincount)"}
    out["gnn_tsp500_k50_b64"] = {"workload": f"Net.forward eval, {B} graphs of TSP-{n} (k={k}) in one pass",
                                 "value": B / dt, "unit": "graphs/s", "ms_per_step": dt * 1e3,
                                 "roofline": {"bound": "hbm", "achieved": alg / dt / 1e9, "peak": PEAK_HBM_GBS,
                                              "unit": "GB/s", "frac": alg / dt / 1e9 / PEAK_HBM_GBS, "traffic": tr,
                                              "traffic_source": src,
                                              "kernel": "gnn_fused2_layer_kernel x 12 layers (layer 0 makes the edge state) + node init + head (whole forward)",
                                              "mfma_tflops": flops / dt / 1e12,
                                              "pipes": counters.get("gnn_fused2_layer_tsp500_k50_b64"),
                                              "valu_busy": (counters.get("gnn_fused2_layer_tsp500_k50_b64") or {}).get("valu_busy")},
                                 "cpu_baseline": cb}
    return out


def load_traffic():
    """profiles/hbm_traffic.json: counter-measured HBM-side bytes per launch of the dominant kernels (rocprofv3 --pmc passes,
    corrected as MI355X_MICROARCH.md prescribes), stamped with the library version they were collected with.  Figures of
    another version are dropped: a kernel change that was not re-profiled must not keep an old number."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        tj = json.load(open(path))
        from deepaco_amd import _lib
        if int(tj.get("daco_version", -1)) != _lib.lib().daco_version():
            log(f"profiles/hbm_traffic.json is for library version {tj.get('daco_version')}, this is "
                f"{_lib.lib().daco_version()}: counter-measured traffic not reported")
            return {}
        src = tj.get("source", "profiles/hbm_traffic.json")
        return {k: (float(v), f"{src} (rocprofv3 --pmc passes of this workload and library version, not collected in this run)")
                for k, v in tj.items() if isinstance(v, (int, float)) and k != "daco_version"}
    except Exception:
        return {}


def live_traffic(needle, cmd, timeout=150):
    """HBM-side bytes per launch of the kernel whose name contains `needle`, collected NOW: two rocprofv3 counter passes (FETCH_SIZE,
    then WRITE_SIZE: they do not fit one pass, and counters are collected with the kernel trace only -- MI355X_MICROARCH.md) of the
    command `cmd`, which launches that kernel at this workload a few times.  bytes = FETCH_SIZE KiB x 1024 x 2 (the guide's gfx950
    correction for wide coalesced reads) + WRITE_SIZE KiB x 1024, mean per launch.  None if rocprofv3 is not there or a pass fails
    (the stored figure of profiles/hbm_traffic.json is used then, and says so)."""
    import csv
    import glob
    import shutil
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="daco_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "p", "--"] + cmd,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
            got = []
            for f in glob.glob(os.path.join(tmp, "**", "p_counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if needle in row.get("Kernel_Name", "") and row.get("Counter_Name") == ctr:
                        got.append(float(row["Counter_Value"]))
            if not got:
                log(f"live counter pass {ctr}: no rows for {needle!r} (rc {r.returncode})")
                return None
            vals[ctr] = sum(got) / len(got)
        except Exception as e:
            log(f"live counter pass {ctr} failed: {e!r}")
            return None
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return vals["FETCH_SIZE"] * 1024.0 * 2.0 + vals["WRITE_SIZE"] * 1024.0


def load_counters():
    """profiles/counters.json: what the kernels' hardware counters said (rocprofv3 --pmc passes of the workloads below,
    tools/profile_r4.sh), stamped with the library version like hbm_traffic.json and dropped when it differs.  Pipe
    occupancies of the dominant kernel next to its roofline: which unit the kernel sits on is a counter, not a guess."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    try:
        cj = json.load(open(path))
        from deepaco_amd import _lib
        if int(cj.get("daco_version", -1)) != _lib.lib().daco_version():
            return {}
        return {k: dict(v, source=cj.get("source", "profiles/counters.json") + " (PMC passes of this workload and library "
                                                                                 "version, not collected in this run)")
                for k, v in cj.items() if isinstance(v, dict)}
    except Exception:
        return {}


def active_knobs():
    """DACO_* environment variables set for this run (three of them select which kernel a result comes from)."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("DACO_")}


# ---------------------------------------------------------------------------------------------- one rank
# ---------------------------------------------------------------------------------------------- best-cost gap
T975 = {1: 12.706, 2: 4.303, 3: 3.182, 4: 2.776, 5: 2.571, 6: 2.447, 7: 2.365, 8: 2.306, 9: 2.262, 10: 2.228, 11: 2.201, 12: 2.179,
        13: 2.160, 14: 2.145, 15: 2.131, 16: 2.120, 17: 2.110, 18: 2.101, 19: 2.093, 20: 2.086, 21: 2.080, 22: 2.074, 23: 2.069,
        24: 2.064, 25: 2.060, 26: 2.056, 27: 2.052, 28: 2.048, 29: 2.045, 30: 2.042, 40: 2.021, 60: 2.000, 120: 1.980}
GAP_UPPER_BOUND = 0.0025        # "equal or better": the 95 % interval of the paired relative difference ends at or below +0.25 %


def t975(dof):
    if dof in T975:
        return T975[dof]
    if dof < 1:
        return float("nan")
    below = max(k for k in T975 if k <= dof)            # (conservative: the quantile of the next smaller tabulated dof)
    return T975[below] if dof < 120 else 1.960 + (1.980 - 1.960) * 120 / dof


def gap_samplers(resolved):
    """The default sampler of the workload first, then the two samplers the reference fixtures pin bit for bit."""
    out = [resolved]
    for s_ in ("scan", "race"):
        if s_ not in out:
            out.append(s_)
    return out


def gap_statistics(gpu_best, cpu_best):
    """gpu_best [seeds][instances], cpu_best [seeds][instances] best costs after the same number of iterations: the paired
    per-instance relative difference of the seed means, its Student-t 95 % interval, and the verdict."""
    ni = len(cpu_best[0])
    g = [sum(run[i] for run in gpu_best) / len(gpu_best) for i in range(ni)]
    c = [sum(run[i] for run in cpu_best) / len(cpu_best) for i in range(ni)]
    rel = [(g[i] - c[i]) / c[i] for i in range(ni)]
    mean_rel = sum(rel) / ni
    sd = (sum((r - mean_rel) ** 2 for r in rel) / max(1, ni - 1)) ** 0.5
    half = t975(ni - 1) * sd / ni ** 0.5 if ni > 1 else None
    gm, cm = sum(g) / ni, sum(c) / ni
    return {"gap": (gm - cm) / cm, "gpu_mean_best": gm, "cpu_mean_best": cm,
            "mean_paired_relative_difference": mean_rel,
            "ci95": None if half is None else [mean_rel - half, mean_rel + half],
            "ci95_half_width": half,
            "equal_or_better": None if half is None else bool(mean_rel + half <= GAP_UPPER_BOUND),
            "gpu_better_or_equal_on": int(sum(g[i] <= c[i] for i in range(ni)))}


def best_cost_gap(dist_cpu, k_sparse, A, cpu_best, iters, dev, samplers, default, gpu_seeds=(99, 199, 299)):
    """(mean best cost of the GPU colonies - the CPU reference path's) / the CPU's, on the same instances after the same number
    of colony iterations, independent RNG streams, per sampler.  cpu_best: [cpu seeds][instances] best costs of the torch port
    (the reference's op sequence).  Every GPU colony is fresh (tau = 1), `gpu_seeds` runs per sampler; an instance's cost is the
    mean over the seeds on either side.  equal_or_better = the 95 % interval of the paired relative difference ends at or
    below +0.25 % (an interval that merely touches zero is not evidence of equality)."""
    from deepaco_amd import engine
    ni = len(cpu_best[0])
    d_dev = dist_cpu[:ni].to(dev)
    rows = {}
    for smp in samplers:
        runs = []
        for sd_ in gpu_seeds:
            col = engine.BatchedTSP(d_dev, n_ants=A, sampler=smp, seed=sd_)
            col.sparsify(k_sparse)
            col.run(iters)
            runs.append([float(x) for x in col.lowest_cost.cpu()])
        rows[smp] = gap_statistics(runs, cpu_best)
    out = dict(rows[default])
    out.update({"instances": ni, "iterations": iters, "gpu_seeds": len(gpu_seeds), "cpu_seeds": len(cpu_best), "sampler": default,
                "upper_bound_for_equal": GAP_UPPER_BOUND,
                "samplers": {k: {kk: v[kk] for kk in ("gap", "ci95", "equal_or_better", "gpu_better_or_equal_on")} for k, v in rows.items()},
                "note": "same instances, equal iterations, independent RNG streams; per instance the mean over the seeds on either side; "
                        "ci95 = mean +- t(0.975, n-1) s / sqrt(n) of the per-instance (gpu - cpu) / cpu; equal_or_better = the "
                        "interval's upper end <= +0.25 %"})
    return out


def gpu_topology():
    """Link types between the node's GPUs as rocm-smi reports them (best effort; one string for the record):
    e.g. "8 GPUs: 56 XGMI links" on an MI355X node, "1 GPU" on a single-GPU box."""
    import torch
    ng = torch.cuda.device_count()
    base = f"{ng} GPU{'s' if ng != 1 else ''}"
    if ng < 2:
        return base
    try:
        out = subprocess.run(["rocm-smi", "--showtopotype"], capture_output=True, text=True, timeout=20).stdout
        kinds = {}
        for line in out.splitlines():
            toks = line.split()
            if toks and toks[0].startswith("GPU") and len(toks) > 1:          # a row of the link-type matrix
                for t in toks[1:]:
                    if t.upper() in ("XGMI", "PCIE"):
                        kinds[t.upper()] = kinds.get(t.upper(), 0) + 1
        return base + (": " + ", ".join(f"{c} {t} links" for t, c in sorted(kinds.items())) if kinds else " (link types not reported)")
    except Exception as e:
        return base + f" (rocm-smi unavailable: {type(e).__name__})"


def worker(args):
    quiet_stdout()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"rank {rank}: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size is what runs "
            f"(n_gpus = {world})")
    distributed = world > 1 or args.force_dist
    if distributed:
        import torch.distributed as dist_pkg
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:           # --force-dist without a launcher
            s_ = socket.socket()
            s_.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
            s_.close()
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    dev_index = local_rank if args.force_device is None else args.force_device
    if dev_index >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: needs GPU {dev_index}, this node shows {torch.cuda.device_count()}")
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
        # the SAME sampler as the N = 1 line: "auto" resolves to the head / tail rows with the k of the
        # sparsified heuristic, exactly as BatchedTSP.sparsify(k) leaves it
        colony = engine.ant_sharded_tsp(d_dev, A, rank, world, heuristic=(1 / sparse).contiguous(),
                                        sampler=args.sampler, head_k=min(k_sparse, 127), seed=1234,
                                        exchange=args.exchange)
        _step = colony.step
        colony.step = lambda events=None: _step()
    streams = 1 if ant_sharded else max(1, min(args.streams, B))

    def make_colony(seed, gid0):
        if streams > 1:      # (StreamedTSP.sparsify keeps the parts' heuristics contiguous)
            col = engine.StreamedTSP(dist_cpu.to(dev), parts=streams, n_ants=A, sampler=args.sampler, seed=seed, ant_gid0=gid0)
            col.sparsify(k_sparse)
        else:
            col = engine.BatchedTSP(dist_cpu.to(dev), n_ants=A, sampler=args.sampler, seed=seed, ant_gid0=gid0)
            col.sparsify(k_sparse)
            col.heuristic = col.heuristic.contiguous()
        return col

    if not ant_sharded:
        colony = make_colony(1234, rank * B * A)

    if ant_sharded:
        resolved = engine.resolve_sampler(args.sampler, n, min(k_sparse, 127), None, {})[0]
    else:
        resolved = (colony.cols[0] if streams > 1 else colony).resolved_sampler()[0]
    log(f"rank {rank}/{world}: TSP-{n} x {A} ants x {B} instances, sampler={args.sampler} -> {resolved}")
    import gc
    gc.collect()                  # (before the warm-up, not between it and the timed region: see parallel.barrier_max_time)
    pre_steps = 0
    if args.precondition_seconds > 0 and not ant_sharded:
        # a throw-away colony of the same shape keeps the device busy until its clocks have settled; the measured colony
        # starts from its own initial state right after
        pre = make_colony(4321, 0)
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.precondition_seconds:
            for _ in range(20):
                pre.step()
            torch.cuda.synchronize()
            pre_steps += 20
        del pre
    # one event pair per launch of the construction kernel: per step, and per stream when the instances run as several colonies
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(streams)]
          for _ in range(args.steps)]
    for row in ev:       # create the handles; the library re-records them around the kernel
        for a, b in row:
            a.record(); b.record()
    torch.cuda.synchronize()
    # the W untimed warm-up steps run LAST before the timed region: the device is busy right up to the fence (host work between
    # them -- a collection, event set-up -- lets its clocks fall back, which the first timed steps then pay for)
    for _ in range(args.warmup):
        colony.step()

    def timed():
        for s in range(args.steps):
            colony.step(events=ev[s] if streams > 1 else ev[s][0])

    elapsed = barrier_max_time(timed, dev, distributed)
    log(f"timed region: {elapsed*1e3:.1f} ms for {args.steps} steps")
    # average duration of ONE launch of the construction kernel (B / streams instances each)
    kern_ms = sum(a.elapsed_time(b) for row in ev for a, b in row) / (args.steps * streams) if not ant_sharded else None
    tours_per_step = B * A if ant_sharded else world * B * A
    value = tours_per_step * args.steps / elapsed
    gpu_best = colony.lowest_cost.detach().cpu()

    # sustained loop (not the metric): the same steps for >= min-seconds
    sustained = None
    if args.min_seconds > 0:
        chunk = max(1, int(0.25 / max(elapsed / args.steps, 1e-5)))
        state = {"steps": 0}

        def sustain():
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < args.min_seconds:
                for _ in range(chunk):
                    colony.step()
                torch.cuda.synchronize()
                state["steps"] += chunk
                if distributed:          # every rank must run the same number of chunks (ant-sharded steps communicate)
                    flag = torch.tensor([time.perf_counter() - t0 < args.min_seconds], dtype=torch.int32,
                                        device=dev if args.dist_backend == "nccl" else "cpu")
                    dist_pkg.broadcast(flag, 0)
                    if not int(flag.item()):
                        break
        s_el = barrier_max_time(sustain, dev, distributed)
        sustained = {"seconds": s_el, "steps": state["steps"], "value": tours_per_step * state["steps"] / s_el,
                     "unit": "ant-tours/s"}
        log(f"sustained: {state['steps']} steps in {s_el:.2f} s")

    rccl = None
    if distributed:
        rccl = {"ranks": world, "backend": args.dist_backend,
                "data_path_collective": "none (instance-sharded)" if not ant_sharded else
                ("all-gather of int16 tours + f32 costs per iteration" if args.exchange == "tours"
                 else "all-reduce of delta-tau [B,n,n] f32 per iteration")}
        rccl["visible_gpus"] = torch.cuda.device_count()
        rccl["topology"] = gpu_topology()
        if args.dist_backend == "nccl":
            try:
                rccl["version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception:
                rccl["version"] = None
            buf = torch.ones((B, n, n), device=dev)
            for _ in range(2):
                dist_pkg.all_reduce(buf)
            reps = 5
            t_ar = barrier_max_time(lambda: [dist_pkg.all_reduce(buf) for _ in range(reps)], dev, True) / reps
            nbytes = buf.numel() * 4
            rccl["allreduce_bytes"] = nbytes
            rccl["allreduce_ms"] = t_ar * 1e3
            rccl["allreduce_busbw_GBps"] = 2 * (world - 1) / world * nbytes / t_ar / 1e9

    if rank == 0:
        traffic, tsrc = load_traffic().get(f"tsp{n}_a{A}_b{B}_{resolved}", (None, None))
        if world == 1 and not args.no_extras and not ant_sharded and streams == 1 and resolved in ("scan_sparse", "scan"):
            # the HBM-side bytes of the dominant kernel, collected in THIS run (a stored figure cannot show a
            # regression): two counter passes of the kernel alone at this workload, in subprocesses
            needle = "scan_sparse_kernel" if resolved == "scan_sparse" else "tsp_scan32_kernel"
            live = live_traffic(needle, [sys.executable, os.path.join(ROOT, "tools", "run_headline_kernel.py"), "5", str(B), str(A), str(n), resolved])
            if live is not None:
                traffic, tsrc = live, ("collected in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (kernel trace only) of "
                                       "tools/run_headline_kernel.py at this workload; FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, per launch")
        line = {
            "metric": "ant-tours/sec, TSP-500 n_ants=512", "value": value, "unit": "ant-tours/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if ant_sharded else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"TSP-{n} random-Euclidean, n_ants={A}, {B} instances per GPU, "
                                   f"AS update, heuristic 1/d sparsified k={k_sparse}, sampler={args.sampler}"
                                   + (f" -> {resolved}" if resolved != args.sampler else ""),
                       "nodes": n, "n_ants": A, "instances_per_gpu": B, "sampler": resolved,
                       "parallelism": f"{'ant' if ant_sharded else 'instance'}-sharded x{world}",
                       "streams_per_gpu": streams,
                       "streams": (f"{streams} colonies of {B // streams}-{-(-B // streams)} instances on {streams} HIP streams per GPU "
                                   f"(engine.StreamedTSP: the results of one colony over all instances, bit for bit)") if streams > 1
                       else "one colony over all instances"},
            # per LAUNCH of the construction kernel: B / streams instances (the counter passes are of one launch over all B
            # instances of the same kernel: their per-launch byte counts are scaled, their occupancies are ratios)
            "roofline": roofline_rows(n, A, B / streams, resolved, kern_ms,
                                      traffic=traffic / streams if traffic is not None else None,
                                      traffic_source=(tsrc + f"; one launch here covers 1/{streams} of the instances of that pass: scaled"
                                                      if tsrc and streams > 1 else tsrc),
                                      pipes=load_counters().get(f"tsp{n}_a{A}_b{B}_{resolved}"), head_k=k_sparse)
            if kern_ms else None,
            "knobs": active_knobs(),
            "preconditioning": {"steps": pre_steps, "seconds": args.precondition_seconds,
                                "note": "untimed steps of a throw-away colony of the same shape before the warm-up steps (device clocks "
                                        "settle: the first ~20 ms of a launch sequence run 5-8 % slower than the sustained loop)"},
            "sustained": sustained,
            "gpu_mean_best_cost": float(gpu_best.mean()),
        }
        if rccl:
            line["rccl"] = rccl
        if world == 1 and not args.no_extras and not ant_sharded:
            try:
                line["extras"] = extra_configs(dev, colony, cpu=not args.no_cpu)
            except Exception as e:          # the headline must still be reported
                log(f"extras failed: {e!r}")
                line["extras"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu and not ant_sharded:
            # 16 colonies x 2 threads: the best aggregate found on this host class (round 2: 2 103 ant-tours/s; 64 x 2 on the
            # same 128 cores gave 1 594 -- the op sequence is memory-bound and slows down as colonies are added)
            # (round 6: 32 colonies x 20 iterations -- the gap's interval needs the instances; the aggregate rate is the same)
            ncol = args.cpu_instances or max(1, min(B, 32, (os.cpu_count() or 1) // 4))
            cb, cpu_best, done = cpu_baseline(dist_cpu, k_sparse, A, ncol, args.cpu_iters, budget_s=args.cpu_seconds)
            line["cpu_baseline"] = cb
            # best-cost gap (BASELINE.json: "at equal or better best-cost gap"): the same instances, equal iterations, fresh GPU
            # colonies of the default sampler AND of the two reference-pinned ones, three seeds each
            line["best_cost_gap"] = best_cost_gap(dist_cpu[:len(cpu_best)], k_sparse, A, [cpu_best], done, dev,
                                                  samplers=gap_samplers(resolved), default=resolved)
            # ... and the full-power form of the same statistic (tools/best_cost_gap.py: 64 instances x 3 seeds on both sides, 20
            # iterations, ~22 minutes of CPU), from the committed record: the live sample above is what a few minutes allow
            try:
                st = json.load(open(os.path.join(ROOT, "profiles", "r06_best_cost_gap.json")))
                line["best_cost_gap"]["study"] = {
                    "source": "profiles/r06_best_cost_gap.json (tools/best_cost_gap.py on the GPU box, this round; not collected in this run)",
                    "instances": st["instances"], "iterations": st["iterations"], "cpu_seeds": st["cpu_seeds"], "gpu_seeds": st["gpu_seeds"],
                    "samplers": st["samplers"]}
            except Exception:
                pass
            line["speedup_vs_cpu"] = value / cb["value"]
        emit(line)
    if distributed:
        dist_pkg.barrier()
        dist_pkg.destroy_process_group()


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    worker(args)


if __name__ == "__main__":
    main()
