#!/usr/bin/env python3
"""Randomised soak: scan-draw tours (all lane layouts, TSP and CVRP) against the CPU oracle on random sizes,
ant counts, seeds and value distributions (uniform, heavy-tailed, sparse with exact zeros, tiny).
usage: python tests/soak_parity.py [cases] [seed]   -- prints one line per mismatch and a summary
(test infrastructure: tests/test_gpu_08_soak.py runs a short soak in the GPU suite)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from deepaco_amd import engine  # noqa: E402

def run(cases, seed, save_failures=True):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    bad = 0
    for c in range(cases):
        bad += 0 if one_case(c, rng, dev, save_failures) else 1
    return bad


def one_case(c, rng, dev, save_failures):
    n = int(rng.choice([rng.integers(2, 65), rng.integers(65, 257), rng.integers(257, 513), rng.integers(513, 700)]))
    A = int(rng.integers(1, 40))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        P = rng.random((n, n)) + 1e-3
    elif kind == 1:
        P = np.exp(rng.normal(0, 6, (n, n)))
    elif kind == 2:
        P = np.exp(rng.uniform(-60, 10, (n, n)))
        P[rng.random((n, n)) < 0.4] = 0.0
        idx = np.arange(n)
        for off in (1, 2, 3, 5):
            P[idx, (idx + off) % n] = np.maximum(P[idx, (idx + off) % n], 1e-30)
    else:
        P = rng.random((n, n)) * 1e-38 + 1e-40
    P = P.astype(np.float32)
    seed, it = int(rng.integers(1, 2 ** 40)), int(rng.integers(0, 1000))
    wave = bool(rng.integers(0, 4) == 0)
    race = bool(rng.integers(0, 5) == 0)
    mode = "race" if race else ("scan_wave" if wave else "scan")
    tau = torch.from_numpy(P)[None].contiguous().to(dev)
    eta = torch.ones(1, n, n, device=dev)
    cvrp = n >= 4 and rng.integers(0, 3) == 0
    if cvrp:
        demand = np.concatenate(([0.0], rng.integers(1, 10, n - 1))).astype(np.float32)
        cap = float(rng.integers(10, 60))
        paths, _, _, lens, flags = engine.cvrp_sample(tau, eta, torch.from_numpy(demand).to(dev), cap, A, mode=mode,
                                                      seed=seed, it=it)
        rp, _, L = oracle.cvrp_sample_rng(P, demand, cap, A, mode, seed, it)
        if L < 0:
            ok = int(flags[0]) != 0
        else:
            ok = int(flags[0]) == 0 and L == int(lens.max()) and np.array_equal(paths[0, :L].cpu().numpy(), rp)
            if ok:                                          # the fused outputs: route costs and the deposit's table
                d = (rng.random((n, n)) + 0.01).astype(np.float32)
                d[0, 0] = 1e-10
                D = torch.from_numpy(d)[None].to(dev)
                p2, _, _, _, _, costs, table = engine.cvrp_sample(tau, eta, torch.from_numpy(demand).to(dev), cap, A,
                                                                  mode=mode, seed=seed, it=it, dist=D, want_table=True)
                t1, t2 = tau.clone(), tau.clone()
                engine.pheromone_update_(t1, p2, costs, 0.9, False, False, floor=1e-10, nbr=table)
                engine.pheromone_update_(t2, p2[:, :L].contiguous(), costs, 0.9, False, False, floor=1e-10)
                ok = torch.equal(p2, paths) and torch.equal(t1, t2) and \
                    np.array_equal(costs[0].cpu().numpy(), oracle.tour_costs(d, rp, closed=False))
    else:
        paths, _, _, flags = engine.tsp_sample(tau, eta, A, mode=mode, seed=seed, it=it)
        rp, _, rc = oracle.tsp_sample_race(P, A, seed, it) if race else oracle.tsp_sample_scan(P, A, seed, it, wave=wave)
        if rc:                         # a draw without feasible candidate: flagged, and both move to node 0
            ok = int(flags[0]) == 1 and np.array_equal(paths[0].cpu().numpy(), rp)
        else:
            ok = int(flags[0]) == 0 and np.array_equal(paths[0].cpu().numpy(), rp)
            if ok and n >= 3:                               # the fused outputs: tour costs and the neighbour table
                d = (rng.random((n, n)) + 0.01).astype(np.float32)
                D = torch.from_numpy(d)[None].to(dev)
                p2, _, _, _, costs, nbr = engine.tsp_sample(tau, eta, A, mode=mode, seed=seed, it=it, dist=D, want_nbr=True)
                t1, t2 = tau.clone(), tau.clone()
                engine.pheromone_update_(t1, p2, costs, 0.9, nbr=nbr)
                engine.pheromone_update_(t2, p2, costs, 0.9)
                ok = torch.equal(p2, paths) and torch.equal(t1, t2) and \
                    np.array_equal(costs[0].cpu().numpy(), oracle.tour_costs(d, rp))
    if not ok:
        if save_failures:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez(os.path.join(ROOT, "gpurun_out", f"soak_fail_{c}.npz"), P=P, A=A, seed=seed, it=it, wave=wave,
                     cvrp=cvrp, gpu_paths=paths[0].cpu().numpy(), flags=flags.cpu().numpy(),
                     ref_paths=rp if rp is not None else np.zeros(1))
        print(f"MISMATCH case {c}: n={n} A={A} kind={kind} mode={mode} cvrp={cvrp} seed={seed} it={it}", flush=True)
    return ok


def random_routes(rng, n, A, demand=None, cap=None):
    """[L, A] int64: random permutations (TSP) or random feasible CVRP routes padded with the depot."""
    if demand is None:
        return np.stack([rng.permutation(n) for _ in range(A)], axis=1).astype(np.int64)
    cols = []
    for _ in range(A):
        order, route, load = rng.permutation(np.arange(1, n)), [0], 0.0
        for c in order:
            if load + demand[c] > cap:
                route.append(0)
                load = 0.0
            route.append(int(c))
            load += demand[c]
        route.append(0)
        cols.append(route)
    L = max(len(c) for c in cols)
    return np.stack([np.array(c + [0] * (L - len(c))) for c in cols], axis=1).astype(np.int64)


def update_case(c, rng, dev):
    """pheromone update (symmetric / directed, AS / elitist / MMAS clamp) and 2-opt against the oracle."""
    n = int(rng.choice([rng.integers(4, 40), rng.integers(40, 130), rng.integers(130, 520)]))
    A = int(rng.integers(1, 150))
    tau = (rng.random((n, n)) + 0.05).astype(np.float32)
    elitist = bool(rng.integers(0, 3) == 0)
    clamp = bool(rng.integers(0, 3) == 0)
    cmin, cmax = (0.2, 0.9) if clamp else (0.0, 0.0)
    decay = float(rng.choice([0.9, 0.5, 0.99]))
    directed = n >= 6 and bool(rng.integers(0, 2))
    if directed:
        demand = np.concatenate(([0.0], rng.integers(1, 10, n - 1))).astype(np.float32)
        paths = random_routes(rng, n, A, demand, float(rng.integers(10, 40)))
    else:
        paths = random_routes(rng, n, A)
    costs = (rng.random(A) * 10 + 1).astype(np.float32)
    t = torch.from_numpy(tau)[None].clone().contiguous().to(dev)
    kw = {}
    if clamp:
        kw = dict(clamp_min=torch.full((1,), cmin, device=dev), clamp_max=torch.full((1,), cmax, device=dev))
    engine.pheromone_update_(t, torch.from_numpy(paths)[None].contiguous().to(dev), torch.from_numpy(costs)[None].to(dev),
                             decay, elitist, not directed, floor=1e-10 if directed else 0.0, **kw)
    if directed:
        ref = oracle.pheromone_update_directed(tau, paths, costs, decay, None, elitist, cmin, cmax, 1e-10)
    else:
        ref = oracle.pheromone_update_tsp(tau, paths, costs, decay, elitist, cmin, cmax)
    ok = np.array_equal(t[0].cpu().numpy().view(np.uint32), ref.view(np.uint32))
    if ok and not directed and n <= 200:                   # 2-opt on a few of the tours, random (asymmetric) matrix
        d = (rng.random((n, n)) + 0.01).astype(np.float32)
        if rng.integers(0, 2):
            d = ((d + d.T) / 2).astype(np.float32)
        tours = paths[:, :min(A, 6)].T.astype(np.uint16).copy()
        cap = int(rng.choice([1, 5, 1000]))
        out, sweeps = engine.two_opt_(torch.from_numpy(d).to(dev), torch.from_numpy(tours.astype(np.int16)).to(dev), cap,
                                      want_sweeps=True)
        rt, rs = oracle.two_opt_batch(d, tours, cap)
        ok = np.array_equal(out.cpu().numpy().astype(np.uint16), rt) and np.array_equal(sweeps[0].cpu().numpy(), rs)
    if not ok:
        print(f"MISMATCH update case {c}: n={n} A={A} elitist={elitist} clamp={clamp} directed={directed}", flush=True)
    return ok


def batch_case(c, rng, dev):
    """several instances per launch (workgroup remap, batch strides, per-instance ant ids): every instance must
    equal its own single-instance oracle run."""
    n = int(rng.choice([rng.integers(2, 65), rng.integers(65, 257), rng.integers(257, 513), rng.integers(513, 600)]))
    A, B = int(rng.integers(1, 30)), int(rng.integers(2, 12))
    P = (rng.random((B, n, n)) ** 2 + 1e-3).astype(np.float32)
    seed, it, gid0 = int(rng.integers(1, 2 ** 40)), int(rng.integers(0, 1000)), int(rng.integers(0, 10 ** 6))
    mode = str(rng.choice(["scan", "scan_wave", "race"]))
    d = (rng.random((B, n, n)) + 0.01).astype(np.float32)
    tau = torch.from_numpy(P).to(dev)
    share_eta = bool(rng.integers(0, 2))                       # a shared (stride-0) matrix next to a batched one
    eta = torch.ones(n, n, device=dev) if share_eta else torch.ones(B, n, n, device=dev)
    paths, _, _, flags, costs, _ = engine.tsp_sample(tau, eta, A, mode=mode, seed=seed, it=it, ant_gid0=gid0,
                                                     dist=torch.from_numpy(d).to(dev), want_nbr=True, batch=B)
    ok = int(flags.sum()) == 0
    for b in range(B):
        fn = oracle.tsp_sample_race if mode == "race" else oracle.tsp_sample_scan
        kw = {} if mode == "race" else {"wave": mode == "scan_wave"}
        rp, _, rc = fn(P[b], A, seed, it, gid0 + b * A, **kw)
        ok = ok and rc == 0 and np.array_equal(paths[b].cpu().numpy(), rp)
        if n >= 3:
            ok = ok and np.array_equal(costs[b].cpu().numpy(), oracle.tour_costs(d[b], rp))
    if not ok:
        print(f"MISMATCH batch case {c}: n={n} A={A} B={B} mode={mode} seed={seed} it={it} gid0={gid0}", flush=True)
    return ok


def run_batches(cases, seed):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    return sum(0 if batch_case(c, rng, dev) else 1 for c in range(cases))


def sibling_case(c, rng, dev, grad=False):
    """fused sibling constructions (one launch) against their draw-by-draw service path: same Philox counters,
    identical solutions (SOP precedences, PCTSP prize threshold, OP length budget, MKP knapsacks)."""
    kind = int(rng.integers(0, 4))
    n = int(rng.choice([rng.integers(6, 40), rng.integers(40, 140), rng.integers(140, 260)]))
    A = int(rng.integers(2, 48))
    sampler = str(rng.choice(["scan", "race"]))
    seed = int(rng.integers(1, 2 ** 31))
    g = torch.Generator().manual_seed(int(rng.integers(1, 2 ** 31)))
    kw = {}
    if kind == 0:
        from deepaco_amd.sop.aco import ACO
        dist = torch.rand(n, n, generator=g) + 0.05
        prec = torch.zeros(n, n)
        order = torch.randperm(n - 1, generator=g) + 1
        for _ in range(int(rng.integers(0, 3 * n))):
            i, j = sorted(torch.randint(0, n - 1, (2,), generator=g).tolist())
            if i != j:
                prec[order[j], order[i]] = 1
        prec[1:, 0] = 1
        make = lambda: ACO(dist.to(dev), prec.to(dev), n_ants=A, device="cuda:0", sampler=sampler, seed=seed)
        name = "gen_path"
    elif kind == 1:
        from deepaco_amd.pctsp.aco import ACO
        coor = torch.rand(n, 2, generator=g)
        dist = torch.cdist(coor, coor)
        prizes = torch.cat((torch.zeros(1), torch.rand(n - 1, generator=g)))
        pen = torch.cat((torch.zeros(1), torch.rand(n - 1, generator=g) * 0.3))
        make = lambda: ACO(dist.to(dev), prizes.to(dev), pen.to(dev), n_ants=A, device="cuda:0", sampler=sampler, seed=seed)
        name = "gen_sol"
    elif kind == 2:
        from deepaco_amd.op.aco import ACO
        from deepaco_amd.tsp.utils import gen_distance_matrix
        coor = torch.rand(n, 2, generator=g)
        dist = gen_distance_matrix(coor)
        dd = (coor - coor[0]).norm(dim=-1)
        prizes = 1 + torch.floor(99 * dd / dd.max())
        prizes = prizes / prizes.max()
        max_len = float(rng.uniform(1.5, 5.0))
        make = lambda: ACO(dist.to(dev), prizes.to(dev), max_len, n_ants=A, k_sparse=max(5, n // 5), device="cuda:0",
                           sampler=sampler, seed=seed)
        name = "gen_sol"
    else:
        from deepaco_amd.mkp.aco import ACO
        m = int(rng.integers(1, 7))
        n = min(n, 120)
        prize = torch.rand(n, generator=g)
        w = torch.rand(n, m, generator=g)
        cons = w.max(0).values + torch.rand(m, generator=g) * (w.sum(0) - w.max(0).values)
        w = w * (n // 2) / cons.unsqueeze(0)
        kw = {"_start": torch.randint(0, n, (A,), generator=g).to(dev)}
        make = lambda: ACO(prize.to(dev), w.to(dev), n_ants=A, device="cuda:0", sampler=sampler, seed=seed)
        name = "gen_sol"
    a1, a2 = make(), make()
    if grad:                                                # gradient to the heuristic: fused replay vs per-draw autograd
        h = (torch.rand(a1.heuristic.shape, generator=g) + 0.05).to(dev)
        a1.heuristic, a2.heuristic = h.clone().requires_grad_(True), h.clone().requires_grad_(True)
    s1, l1 = getattr(a1, name)(True, **kw)
    s2, l2 = getattr(a2, name)(True, _stepwise=True, **kw)
    ok = s1.shape == s2.shape and torch.equal(s1, s2) and torch.allclose(l1, l2, rtol=1e-5, atol=2e-6)
    if grad and ok:
        w = torch.linspace(-1, 1, l1.shape[1], device=dev) if l1.shape[1] > 1 else torch.ones(1, device=dev)
        (l1.sum(0) * w).sum().backward()
        (l2.sum(0) * w).sum().backward()
        g1, g2 = a1.heuristic.grad, a2.heuristic.grad
        scale = float(g2.abs().max()) + 1e-30
        ok = torch.allclose(g1, g2, rtol=3e-4, atol=3e-6 * scale)
    if not ok:
        print(f"MISMATCH sibling case {c}: kind={kind} n={n} A={A} sampler={sampler} seed={seed} grad={grad}", flush=True)
    return ok


def run_siblings(cases, seed, grad=False):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    return sum(0 if sibling_case(c, rng, dev, grad) else 1 for c in range(cases))


def grad_case(c, rng, dev):
    """gradient of the log-probabilities w.r.t. the heuristic (sampler forward with log-probs + replay backward)
    against the closed form on the drawn tours, all lane layouts."""
    from oracle import grad as ograd
    from deepaco_amd.tsp.aco import ACO
    n = int(rng.choice([rng.integers(4, 65), rng.integers(65, 257), rng.integers(257, 400)]))
    A = int(rng.integers(1, 12))
    beta = int(rng.integers(1, 3))
    mode = str(rng.choice(["scan", "scan_wave", "race"]))
    seed = int(rng.integers(1, 2 ** 31))
    g = torch.Generator().manual_seed(seed)
    cds = torch.rand(n, 2, generator=g)
    d = torch.cdist(cds, cds)
    d[torch.arange(n), torch.arange(n)] = 1e9
    tau = torch.rand(n, n, generator=g) + 0.2
    eta = torch.rand(n, n, generator=g) + 1e-2
    heu = eta.to(dev).requires_grad_(True)
    aco = ACO(d.to(dev), n_ants=A, heuristic=heu, pheromone=tau.to(dev), beta=beta, device="cuda:0", sampler=mode, seed=seed)
    costs, logp = aco.sample()
    w = torch.linspace(-1, 1, A, device=dev) if A > 1 else torch.ones(1, device=dev)
    (logp.sum(0) * w).sum().backward()
    paths = ACO(d.to(dev), n_ants=A, heuristic=eta.to(dev), pheromone=tau.to(dev), beta=beta, device="cuda:0", sampler=mode,
                seed=seed).gen_path()
    G = np.tile(w.cpu().numpy()[None, :], (n - 1, 1))
    ref = ograd.tsp_grad(tau.numpy(), eta.numpy(), 1, beta, paths.cpu().numpy(), G)
    scale = np.abs(ref).max()
    ok = np.allclose(heu.grad.cpu().numpy(), ref, rtol=3e-4, atol=3e-6 * scale)
    if not ok:
        print(f"MISMATCH grad case {c}: n={n} A={A} beta={beta} mode={mode} seed={seed}", flush=True)
    return ok


def run_grads(cases, seed):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    return sum(0 if grad_case(c, rng, dev) else 1 for c in range(cases))


def run_updates(cases, seed):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    return sum(0 if update_case(c, rng, dev) else 1 for c in range(cases))


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    t0 = time.time()
    n_bad = run(n_cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{n_cases} sampler cases, {n_bad} mismatches, {time.time() - t0:.1f} s")
    t0 = time.time()
    u_bad = run_updates(n_cases // 4, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{n_cases // 4} update / 2-opt cases, {u_bad} mismatches, {time.time() - t0:.1f} s")
    n_bad += u_bad
    t0 = time.time()
    b_bad = run_batches(n_cases // 4, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{n_cases // 4} multi-instance cases, {b_bad} mismatches, {time.time() - t0:.1f} s")
    n_bad += b_bad
    t0 = time.time()
    s_bad = run_siblings(n_cases // 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{n_cases // 20} sibling cases (fused vs draw-by-draw), {s_bad} mismatches, {time.time() - t0:.1f} s")
    n_bad += s_bad
    t0 = time.time()
    g_bad = run_grads(n_cases // 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{n_cases // 20} gradient cases, {g_bad} mismatches, {time.time() - t0:.1f} s")
    n_bad += g_bad
    sys.exit(1 if n_bad else 0)
