#!/usr/bin/env python3
"""Generate golden input/output vectors by IMPORTING the reference (this container only).

The reference (henry-yeh/DeepACO, mounted read-only at /root/reference) ships no
tests and no golden vectors, so parity is pinned by running the
reference itself here and committing the captured I/O as small .npz fixtures.
Nothing of the reference's source travels: fixtures are data only.

Noise contract: Categorical.sample() ->
torch.multinomial(probs, 1, True) -> argmax(probs / q), q = empty_like(probs).exponential_(1).
torch's CPU exponential_ stream (MKL VSL) cannot be re-implemented, so we RECORD q by
tapping torch.multinomial with a functionally identical replacement, and verify that the
tapped run returns exactly what the untapped reference returns under the same seed.

Run:  python tests/golden/gen_golden.py   (writes tests/golden/*.npz)
"""
import importlib.util
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DEEPACO_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "shims"))


def load_ref(subdir, name, alias):
    """Import /root/reference/<subdir>/<name>.py under a unique module alias."""
    path = os.path.join(REF, subdir, name + ".py")
    spec = importlib.util.spec_from_file_location(alias, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[alias] = mod
    spec.loader.exec_module(mod)
    return mod


class NoiseTap:
    """Replacement for torch.multinomial(probs, 1, True) that records q.

    Same arithmetic as aten multinomial's n_sample==1 fast path: q ~ Exp(1) drawn with
    empty_like(probs).exponential_(1); result = argmax(probs / q, -1, keepdim=True).
    """

    def __init__(self):
        self.q = []
        self.orig = torch.multinomial

    def __call__(self, probs, num_samples, replacement=False, *, generator=None):
        assert num_samples == 1
        q = torch.empty_like(probs).exponential_(1)
        self.q.append(q.clone())
        return torch.argmax(probs / q, dim=-1, keepdim=True)

    def __enter__(self):
        torch.multinomial = self
        return self

    def __exit__(self, *a):
        torch.multinomial = self.orig


def rand_instance(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, 2, generator=g)


def tsp_dist(coords):
    # same arithmetic as tsp/utils.py:4-14 (norm of differences, diag = 1e9)
    d = torch.norm(coords[:, None] - coords, dim=2, p=2)
    n = len(coords)
    d[torch.arange(n), torch.arange(n)] = 1e9
    return d


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  ({os.path.getsize(path)/1024:.0f} KB)")


# --------------------------------------------------------------------------- G1 / G2 TSP
def gen_tsp_sampler(tsp_aco):
    cases = [
        # name, n, A, seed, heuristic kind, pheromone kind
        ("tsp_n20_a8_inv", 20, 8, 11, "inv", "ones"),
        ("tsp_n20_a8_learned", 20, 8, 12, "learned", "rand"),
        ("tsp_n50_a16_sparse", 50, 16, 13, "sparse", "rand"),
        ("tsp_n100_a16_learned", 100, 16, 14, "learned", "rand"),
        ("tsp_n37_a5_inv", 37, 5, 15, "inv", "rand"),     # odd sizes: ragged lanes / vectors
    ]
    for name, n, A, seed, hk, pk in cases:
        coords = rand_instance(n, seed)
        dist = tsp_dist(coords)
        g = torch.Generator().manual_seed(seed + 1000)
        if hk == "inv":
            heu = None
        elif hk == "learned":
            heu = torch.rand(n, n, generator=g) ** 3 + 1e-10
        else:
            heu = None
        phe = None if pk == "ones" else (torch.rand(n, n, generator=g) * 1.5 + 0.05)

        def build():
            aco = tsp_aco.ACO(dist.clone(), n_ants=A, heuristic=None if heu is None else heu.clone(),
                              pheromone=None if phe is None else phe.clone())
            if hk == "sparse":
                aco.sparsify(max(5, n // 5))
            return aco

        # untapped reference run
        torch.manual_seed(seed)
        aco = build()
        ref_paths, ref_logp = aco.gen_path(require_prob=True)
        ref_costs = aco.gen_path_costs(ref_paths)
        # tapped run, same seed
        torch.manual_seed(seed)
        aco2 = build()
        with NoiseTap() as tap:
            paths, logp = aco2.gen_path(require_prob=True)
        assert torch.equal(paths, ref_paths), name
        assert torch.equal(logp, ref_logp), name
        # require_prob=False consumes the same stream and returns the same tours
        torch.manual_seed(seed)
        assert torch.equal(build().gen_path(require_prob=False), ref_paths)
        q = torch.stack(tap.q)                                   # [n-1, A, n]
        save("g1_" + name, distances=dist, heuristic=aco2.heuristic, pheromone=aco2.pheromone,
             alpha=np.float32(1), beta=np.float32(1), start=ref_paths[0], noise=q, seed=np.int64(seed),
             sparsify_k=np.int32(max(5, n // 5) if hk == "sparse" else 0),
             paths=ref_paths, log_probs=ref_logp, costs=ref_costs)


def gen_tsp_update(tsp_aco):
    """G2: update_pheronome (tsp/aco.py:95-118) for AS / elitist / MMAS, bitwise."""
    for name, n, A, seed, kw in [
        ("as_n20_a8", 20, 8, 21, {}),
        ("as_n50_a64", 50, 64, 22, {}),
        ("elitist_n50_a16", 50, 16, 23, dict(elitist=True)),
        ("mmas_n50_a16", 50, 16, 24, dict(min_max=True, min=0.05)),
        ("as_n3_a4", 3, 4, 25, {}),           # smallest tour with distinct prev/next
    ]:
        coords = rand_instance(n, seed)
        dist = tsp_dist(coords)
        g = torch.Generator().manual_seed(seed + 1000)
        phe = torch.rand(n, n, generator=g) + 0.1
        phe = (phe + phe.T) / 2
        torch.manual_seed(seed)
        aco = tsp_aco.ACO(dist, n_ants=A, pheromone=phe.clone(), **kw)
        paths = aco.gen_path()
        costs = aco.gen_path_costs(paths)
        extra = {}
        if kw.get("min_max"):
            aco.max = float(n / costs.min())
            extra = dict(clamp_min=np.float32(aco.min), clamp_max=np.float32(aco.max))
        aco.update_pheronome(paths, costs)
        save("g2_tsp_" + name, pheromone_in=phe, paths=paths, costs=costs,
             decay=np.float32(aco.decay), elitist=np.int32(bool(kw.get("elitist"))),
             pheromone_out=aco.pheromone, **extra)


def gen_tsp_run(tsp_aco):
    """U2 trace: run(T) with per-iteration (tau_in, start, noise) -> (paths, costs, tau_out)."""
    for name, n, A, T, seed, kw in [
        ("as_n30_a12", 30, 12, 4, 31, {}),
        ("mmas_n30_a12", 30, 12, 4, 32, dict(min_max=True)),
        ("elitist_n30_a12", 30, 12, 4, 33, dict(elitist=True)),
    ]:
        coords = rand_instance(n, seed)
        dist = tsp_dist(coords)
        torch.manual_seed(seed)
        ref = tsp_aco.ACO(dist, n_ants=A, **kw)
        ref_low = ref.run(T)
        torch.manual_seed(seed)
        aco = tsp_aco.ACO(dist, n_ants=A, **kw)
        rec = dict(tau_in=[], start=[], noise=[], paths=[], costs=[], tau_out=[], lowest=[])
        for _ in range(T):
            rec["tau_in"].append(aco.pheromone.clone())
            orig_gen = aco.gen_path
            cap = {}

            def tapped(require_prob=False, _o=orig_gen, _c=cap):
                with NoiseTap() as tap:
                    p = _o(require_prob)
                _c["q"] = torch.stack(tap.q)
                _c["p"] = p
                return p
            aco.gen_path = tapped
            aco.run(1)
            aco.gen_path = orig_gen
            rec["start"].append(cap["p"][0])
            rec["noise"].append(cap["q"])
            rec["paths"].append(cap["p"])
            rec["costs"].append(aco.gen_path_costs(cap["p"]))
            rec["tau_out"].append(aco.pheromone.clone())
            rec["lowest"].append(torch.as_tensor(aco.lowest_cost))
        assert torch.equal(torch.as_tensor(ref_low), torch.as_tensor(aco.lowest_cost)), name
        assert torch.equal(ref.pheromone, aco.pheromone), name
        save("u2_tsp_" + name, distances=dist, decay=np.float32(aco.decay),
             elitist=np.int32(bool(kw.get("elitist"))), min_max=np.int32(bool(kw.get("min_max"))),
             clamp_min=np.float32(getattr(aco, "min", 0.0)),
             shortest_path=aco.shortest_path,
             **{k: torch.stack(v) for k, v in rec.items()})


# --------------------------------------------------------------------------- tsp_nls
def gen_nls(nls_aco, two_opt):
    # G1 for the tsp_nls sampler (start 0, precomputed prob matrix, explicit renorm)
    for name, n, A, seed in [("nls_n20_a8", 20, 8, 41), ("nls_n50_a16", 50, 16, 42)]:
        coords = rand_instance(n, seed)
        dist = tsp_dist(coords)
        g = torch.Generator().manual_seed(seed + 1000)
        heu = torch.rand(n, n, generator=g) ** 2 + 1e-10
        phe = torch.rand(n, n, generator=g) + 0.1
        torch.manual_seed(seed)
        ref_paths, ref_logp = nls_aco.ACO(dist, n_ants=A, heuristic=heu, pheromone=phe).gen_path(True)
        torch.manual_seed(seed)
        aco = nls_aco.ACO(dist, n_ants=A, heuristic=heu, pheromone=phe)
        with NoiseTap() as tap:
            paths, logp = aco.gen_path(True)
        assert torch.equal(paths, ref_paths) and torch.equal(logp, ref_logp)
        save("g1_" + name, distances=dist, heuristic=heu, pheromone=phe, start=paths[0],
             noise=torch.stack(tap.q), paths=paths, log_probs=logp,
             costs=aco.gen_path_costs(paths))

    # G4: 2-opt (tsp_nls/two_opt.py:6-49)
    for name, n, T, seed in [("n20_t8", 20, 8, 51), ("n50_t8", 50, 8, 52), ("n100_t4", 100, 4, 53)]:
        coords = rand_instance(n, seed)
        dist = tsp_dist(coords).numpy().astype(np.float32)
        rng = np.random.default_rng(seed)
        tours = np.stack([rng.permutation(n) for _ in range(T)]).astype(np.uint16)
        one = np.stack([t.copy() for t in tours])
        deltas = []
        for t in one:
            deltas.append(two_opt.two_opt_once(dist, t, 0))
        full = two_opt.batched_two_opt_python(dist, tours, max_iterations=10000)
        capped = two_opt.batched_two_opt_python(dist, tours, max_iterations=5)
        # sweep counts to convergence
        iters = []
        for t in tours:
            t = t.copy()
            it = 0
            mc = -1.0
            while mc < -1e-6 and it < 10000:
                mc = two_opt.two_opt_once(dist, t, 0)
                it += 1
            iters.append(it)
        save("g4_twoopt_" + name, dist=dist, tours=tours, after_one=one,
             delta_one=np.asarray(deltas, dtype=np.float32), after_full=full,
             after_cap5=capped, sweeps=np.asarray(iters, dtype=np.int32))

    # O4: the NLS driver (tsp_nls/aco.py:241-258) on a small case, training setting
    n, A, seed = 40, 6, 61
    coords = rand_instance(n, seed)
    dist = tsp_dist(coords)
    g = torch.Generator().manual_seed(seed + 1000)
    heu = torch.rand(n, n, generator=g) + 1e-10
    torch.manual_seed(seed)
    aco = nls_aco.ACO(dist, n_ants=A, heuristic=heu)
    paths = aco.gen_path()
    out_nls = aco.nls(paths.clone())
    out_2opt = aco.two_opt(paths.clone())
    save("o4_nls_n40_a6", distances=dist, heuristic=heu, paths=paths, nls_paths=out_nls,
         twoopt_paths=out_2opt, nls_costs=aco.gen_path_costs(out_nls),
         heuristic_dist=aco.heuristic_dist)

    # G6: roulette sampler (tsp_nls/aco.py:260-275) with an injected uniform stream
    n, seed = 30, 71
    coords = rand_instance(n, seed)
    dist = tsp_dist(coords)
    prob = (1.0 / dist).numpy().astype(np.float32)
    rng = np.random.default_rng(seed)
    routes, us = [], []
    for r in range(6):
        u = rng.random(n - 1)
        it = iter(u.tolist())
        orig = random.random
        nls_aco.random.random = lambda: np.float64(next(it))   # numba types random() as float64
        try:
            routes.append(nls_aco._inference_sample(prob, 0))
        finally:
            nls_aco.random.random = orig
        us.append(u)
    save("g6_roulette_n30", probmat=prob, uniforms=np.stack(us), routes=np.stack(routes))


# --------------------------------------------------------------------------- CVRP
def gen_cvrp(cvrp_aco):
    for name, n, A, seed, cap in [("cvrp_n20_a8", 20, 8, 81, 50), ("cvrp_n50_a8", 50, 8, 82, 50),
                                  ("cvrp_n20_a8_cap20", 20, 8, 83, 20)]:
        g = torch.Generator().manual_seed(seed)
        loc = torch.rand(n, 2, generator=g)
        dem = torch.randint(1, 10, (n,), generator=g)
        allloc = torch.cat((torch.tensor([[0.5, 0.5]]), loc), 0)
        demand = torch.cat((torch.zeros(1), dem.float()))
        dist = torch.norm(allloc[:, None] - allloc, dim=2, p=2)
        dist[torch.arange(n + 1), torch.arange(n + 1)] = 1e-10       # cvrp/utils.py:18-22
        heu = torch.rand(n + 1, n + 1, generator=g) + 1e-10
        phe = torch.rand(n + 1, n + 1, generator=g) + 0.1
        torch.manual_seed(seed)
        ref = cvrp_aco.ACO(dist, demand, n_ants=A, heuristic=heu, pheromone=phe.clone(), capacity=cap)
        ref_paths, ref_logp = ref.gen_path(True)
        torch.manual_seed(seed)
        aco = cvrp_aco.ACO(dist, demand, n_ants=A, heuristic=heu, pheromone=phe.clone(), capacity=cap)
        with NoiseTap() as tap:
            paths, logp = aco.gen_path(True)
        assert torch.equal(paths, ref_paths) and torch.equal(logp, ref_logp)
        costs = aco.gen_path_costs(paths)
        aco.update_pheronome(paths, costs)
        tau_as = aco.pheromone.clone()
        el = cvrp_aco.ACO(dist, demand, n_ants=A, heuristic=heu, pheromone=phe.clone(),
                          capacity=cap, elitist=True)
        el.update_pheronome(paths, costs)
        save("g1_" + name, distances=dist, demand=demand, capacity=np.float32(cap), heuristic=heu,
             pheromone=phe, noise=torch.stack(tap.q), paths=paths, log_probs=logp, costs=costs,
             decay=np.float32(aco.decay), pheromone_as=tau_as, pheromone_elitist=el.pheromone)


def gen_grads(tsp_aco, cvrp_aco):
    """G3: gradient of the REINFORCE loss (tsp/train.ipynb:45-49) w.r.t. the heuristic matrix."""
    for name, n, A, seed, beta in [("tsp_n20_a8", 20, 8, 91, 1), ("tsp_n40_a10_beta2", 40, 10, 92, 2)]:
        coords = rand_instance(n, seed)
        dist = tsp_dist(coords)
        g = torch.Generator().manual_seed(seed + 1000)
        heu = (torch.rand(n, n, generator=g) ** 2 + 1e-3).requires_grad_(True)
        phe = torch.rand(n, n, generator=g) + 0.2
        torch.manual_seed(seed)
        aco = tsp_aco.ACO(dist, n_ants=A, heuristic=heu, pheromone=phe, beta=beta)
        with NoiseTap() as tap:
            costs, logp = aco.sample()
        paths_q = torch.stack(tap.q)
        loss = torch.sum((costs - costs.mean()) * logp.sum(dim=0)) / A
        loss.backward()
        # recover the tours (sample() does not return them): replay with the recorded noise
        torch.manual_seed(seed)
        aco2 = tsp_aco.ACO(dist, n_ants=A, heuristic=heu.detach(), pheromone=phe, beta=beta)
        paths = aco2.gen_path()
        save("g3_grad_" + name, distances=dist, heuristic=heu.detach(), pheromone=phe, beta=np.float32(beta),
             start=paths[0], noise=paths_q, paths=paths, log_probs=logp.detach(), costs=costs,
             loss=loss.detach(), grad=heu.grad)
    n, A, seed, cap = 20, 8, 95, 30
    g = torch.Generator().manual_seed(seed)
    loc = torch.rand(n, 2, generator=g)
    dem = torch.randint(1, 10, (n,), generator=g)
    allloc = torch.cat((torch.tensor([[0.5, 0.5]]), loc), 0)
    demand = torch.cat((torch.zeros(1), dem.float()))
    dist = torch.norm(allloc[:, None] - allloc, dim=2, p=2)
    dist[torch.arange(n + 1), torch.arange(n + 1)] = 1e-10
    heu = (torch.rand(n + 1, n + 1, generator=g) + 1e-3).requires_grad_(True)
    torch.manual_seed(seed)
    aco = cvrp_aco.ACO(dist, demand, n_ants=A, heuristic=heu, capacity=cap)
    with NoiseTap() as tap:
        paths, logp = aco.gen_path(True)
    costs = aco.gen_path_costs(paths)
    loss = torch.sum((costs - costs.mean()) * logp.sum(dim=0)) / A
    loss.backward()
    save("g3_grad_cvrp_n20_a8", distances=dist, demand=demand, capacity=np.float32(cap), heuristic=heu.detach(),
         pheromone=aco.pheromone, noise=torch.stack(tap.q), paths=paths, log_probs=logp.detach(), costs=costs,
         loss=loss.detach(), grad=heu.grad)


def gen_net():
    """G5: Net.forward (tsp/net.py:27-45,59-66,84-88) with the shipped checkpoints, eval and train mode.
    Checkpoints are converted to plain float arrays (data, not code)."""
    cases = [("tsp", "tsp100", "tsp", 100, 20), ("tsp", "tsp20", "tsp", 20, 10),
             ("tsp_nls", "tsp100", "tsp_nls", 100, 10), ("cvrp", "cvrp20", "cvrp", 20, None)]
    for sub, ck, kind, n, k in cases:
        net_mod = load_ref(sub, "net", f"ref_net_{sub}")
        utils = load_ref(sub, "utils", f"ref_utils_{sub}")
        sd = torch.load(os.path.join(REF, "pretrained", sub, ck + ".pt"), map_location="cpu")
        model = net_mod.Net()
        model.load_state_dict(sd)
        torch.manual_seed(7)
        if kind == "cvrp":
            demands, distances = utils.gen_instance(n, "cpu")
            pyg = utils.gen_pyg_data(demands, distances, "cpu")
            extra = dict(demand=demands, distances=distances)
        elif kind == "tsp_nls":
            coords = torch.rand(n, 2)
            pyg, distances = utils.gen_pyg_data(coords, k_sparse=k, start_node=0)
            extra = dict(coords=coords, distances=distances)
        else:
            coords = torch.rand(n, 2)
            pyg, distances = utils.gen_pyg_data(coords, k_sparse=k)
            extra = dict(coords=coords, distances=distances)
        model.eval()
        with torch.no_grad():
            heu_eval = model(pyg)
            emb_eval = model.emb_net(pyg.x, pyg.edge_index, pyg.edge_attr)
            mat = net_mod.Net.reshape(pyg, heu_eval) if kind != "cvrp" else heu_eval.reshape(n + 1, n + 1)
        model.train()
        with torch.no_grad():
            heu_train = model(pyg)
        weights = {"w__" + kk: v.float().numpy() for kk, v in sd.items() if v.numel() > 0 and v.dtype.is_floating_point}
        save(f"g5_net_{sub}_{ck}", x=pyg.x, edge_index=pyg.edge_index, edge_attr=pyg.edge_attr, heu_eval=heu_eval,
             emb_eval=emb_eval, heu_train=heu_train, heu_mat=mat, k_sparse=np.int32(k or 0), **extra, **weights)


def gen_netgrad():
    """G7: one training step of the heuristic network (tsp_nls/train.py:15-44 as far as the network goes): train-mode
    forward (BatchNorm on the statistics of the one graph), a scalar loss sum(heu * coef) with a recorded coefficient
    vector, loss.backward().  Stored: heu, every parameter's gradient, the BatchNorm running statistics after the
    forward.  Inputs (graph, weights) are those of the matching g5 fixture (same seed, same checkpoint)."""
    cases = [("tsp", "tsp20", "tsp", 20, 10), ("tsp_nls", "tsp100", "tsp_nls", 100, 10), ("cvrp", "cvrp20", "cvrp", 20, None)]
    for sub, ck, kind, n, k in cases:
        net_mod = load_ref(sub, "net", f"ref_net_{sub}")
        utils = load_ref(sub, "utils", f"ref_utils_{sub}")
        sd = torch.load(os.path.join(REF, "pretrained", sub, ck + ".pt"), map_location="cpu")
        model = net_mod.Net()
        model.load_state_dict(sd)
        torch.manual_seed(7)
        if kind == "cvrp":
            demands, distances = utils.gen_instance(n, "cpu")
            pyg = utils.gen_pyg_data(demands, distances, "cpu")
        elif kind == "tsp_nls":
            pyg, _ = utils.gen_pyg_data(torch.rand(n, 2), k_sparse=k, start_node=0)
        else:
            pyg, _ = utils.gen_pyg_data(torch.rand(n, 2), k_sparse=k)
        model.train()
        heu = model(pyg)
        coef = torch.randn(heu.shape, generator=torch.Generator().manual_seed(99))
        loss = torch.sum(heu * coef)
        loss.backward()
        grads = {"g__" + kk: v.grad.numpy() for kk, v in model.named_parameters() if v.grad is not None and v.numel() > 0}
        stats = {"rs__" + kk: v.numpy() for kk, v in model.state_dict().items() if "running_" in kk}
        save(f"g7_netgrad_{sub}_{ck}", heu_train=heu.detach(), coef=coef, loss=loss.detach(), **grads, **stats)


def gen_bench_weights():
    """The checkpoint the reference ships for the headline size (pretrained/tsp/tsp500.pt) as plain float arrays (data,
    not code): bench.py's learned-heuristic variant loads them into deepaco_amd's Net."""
    sd = torch.load(os.path.join(REF, "pretrained", "tsp", "tsp500.pt"), map_location="cpu")
    save("w_tsp_tsp500", **{"w__" + k: v.float().numpy() for k, v in sd.items() if v.numel() > 0 and v.dtype.is_floating_point})
    # ... and the one SURVEY 8(d) names for config 5's learned variant: pretrained/tsp_nls/tsp1000.pt
    sd = torch.load(os.path.join(REF, "pretrained", "tsp_nls", "tsp1000.pt"), map_location="cpu")
    save("w_tsp_nls_tsp1000", **{"w__" + k: v.float().numpy() for k, v in sd.items() if v.numel() > 0 and v.dtype.is_floating_point})


def load_ref_dir(subdir, alias):
    """Import <subdir>/aco.py with <subdir> on sys.path (smtwtp/aco.py does `import utils`)."""
    d = os.path.join(REF, subdir)
    sys.path.insert(0, d)
    for m in ("utils",):
        sys.modules.pop(m, None)
    try:
        aco = load_ref(subdir, "aco", alias)
        utils = load_ref(subdir, "utils", alias + "_utils")
    finally:
        sys.path.remove(d)
        sys.modules.pop("utils", None)
    return aco, utils


def tapped(fn, *a, **k):
    with NoiseTap() as tap:
        out = fn(*a, **k)
    return out, torch.stack(tap.q)


def gen_siblings():
    """S1-S6: one construction (recorded noise) + objective + pheromone update per sibling problem."""
    A = 8
    # ---- S4 SMTWTP (smtwtp/aco.py)
    aco_m, utils = load_ref_dir("smtwtp", "ref_smtwtp")
    torch.manual_seed(101)
    _, due, wts, proc = utils.instance_gen(20, "cpu")
    aco = aco_m.ACO(due, wts, proc, n_ants=A)
    aco.pheromone = torch.rand(21, 21) + 0.2
    tau0, heu = aco.pheromone.clone(), aco.heuristic.clone()
    (paths, logp), q = tapped(aco.gen_path, True)
    costs = aco.gen_path_costs(paths)
    aco.update_pheronome(paths, costs)
    el = aco_m.ACO(due, wts, proc, n_ants=A, elitist=True, pheromone=tau0.clone())
    el.update_pheronome(paths, costs)
    save("s4_smtwtp_n20", due_time=due, weights=wts, processing_time=proc, pheromone=tau0, heuristic=heu, noise=q,
         paths=paths, log_probs=logp, costs=costs, decay=np.float32(aco.decay), pheromone_as=aco.pheromone,
         pheromone_elitist=el.pheromone)
    # ---- S3 SOP (sop/aco.py)
    aco_m, utils = load_ref_dir("sop", "ref_sop")
    torch.manual_seed(102)
    dist, adj, prec = utils.training_instance_gen(20, "cpu")
    aco = aco_m.ACO(dist, prec, n_ants=A, pheromone=torch.rand(20, 20) + 0.2)
    tau0 = aco.pheromone.clone()
    (paths, logp), q = tapped(aco.gen_path, True)
    costs = aco.gen_path_costs(paths)
    aco.update_pheronome(paths, costs)
    save("s3_sop_n20", distances=dist, prec_cons=prec, pheromone=tau0, heuristic=aco.heuristic, noise=q, paths=paths,
         log_probs=logp, costs=costs, decay=np.float32(aco.decay), pheromone_as=aco.pheromone)
    # ---- S2 PCTSP (pctsp/aco.py)
    aco_m, utils = load_ref_dir("pctsp", "ref_pctsp")
    torch.manual_seed(103)
    dist, prizes, pen = utils.gen_inst(20, "cpu")
    aco = aco_m.ACO(dist, prizes, pen, n_ants=A)
    aco.pheromone = torch.rand(21, 21) + 0.2
    tau0 = aco.pheromone.clone()
    (out, q) = tapped(aco.gen_sol, True)
    sols, logp = out
    objs = aco.gen_sol_obj(sols)
    best_obj, best_idx = objs.max(dim=0)          # as run() does (pctsp/aco.py:73)
    aco.update_pheronome(sols.T, objs, best_obj, best_idx)
    save("s2_pctsp_n20", distances=dist, prizes=prizes, penalties=pen, pheromone=tau0, heuristic=aco.heuristic,
         noise=q, sols=sols, log_probs=logp, objs=objs, decay=np.float32(aco.decay), pheromone_as=aco.pheromone)
    # ---- S1 OP (op/aco.py)
    aco_m, utils = load_ref_dir("op", "ref_op")
    torch.manual_seed(104)
    coor = torch.rand(30, 2)
    _, dist, prizes = utils.gen_pyg_data(coor, k_sparse=8)
    aco = aco_m.ACO(dist.clone(), prizes.clone(), 3.0, n_ants=A, k_sparse=8)
    tau0 = aco.pheromone.clone()
    (out, q) = tapped(aco.gen_sol, True)
    sols, logp = out
    objs = aco.gen_sol_obj(sols)
    best_obj, best_idx = objs.max(dim=0)
    aco.update_pheronome(sols.T, objs, best_obj, best_idx)
    save("s1_op_n30", distances_in=dist, prizes_in=prizes, max_len=np.float32(3.0), k_sparse=np.int32(8),
         distances=aco.distances, prizes=aco.prizes, pheromone=tau0, heuristic=aco.heuristic, Q=aco.Q, noise=q,
         sols=sols, log_probs=logp, objs=objs, decay=np.float32(aco.decay), pheromone_as=aco.pheromone)
    # ---- S5 BPP (bpp/aco.py)
    aco_m, utils = load_ref_dir("bpp", "ref_bpp")
    torch.manual_seed(105)
    demand = utils.gen_instance(24, "cpu")
    aco = aco_m.ACO(demand, n_ants=A)
    # numba types the fitness accumulators as float64 (int64 + float32 -> float64); under the identity
    # shim numpy-2 scalars would stay float32, so hand the jitted functions a float64 demand array
    aco.__dict__["demand_numpy"] = demand.numpy().astype(np.float64)
    aco.pheromone = torch.rand(25, 25) + 0.2
    tau0 = aco.pheromone.clone()
    (out, q) = tapped(aco.gen_path, True)
    paths, logp = out
    costs = aco.gen_path_costs(paths)
    aco.update_pheronome(paths, -costs)           # as run() does (bpp/aco.py:96)
    save("s5_bpp_n24", demand=demand, capacity=np.float32(aco.capacity), pheromone=tau0, heuristic=aco.heuristic,
         noise=q, paths=paths, log_probs=logp, costs=costs, decay=np.float32(aco.decay), pheromone_as=aco.pheromone)
    # ---- S6 MKP (mkp/aco.py)
    aco_m, utils = load_ref_dir("mkp", "ref_mkp")
    torch.manual_seed(106)
    np.random.seed(106)
    prize, weight = utils.gen_instance(20, 3, "cpu")
    aco = aco_m.ACO(prize.clone(), weight.clone(), n_ants=A)
    aco.pheromone = torch.rand(21, 21) + 0.2
    tau0 = aco.pheromone.clone()
    (out, q) = tapped(aco.gen_sol, True)
    sols, logp = out
    objs = aco.gen_sol_obj(sols)
    best_obj, best_idx = objs.max(dim=0)
    aco.update_pheronome(sols.T, objs, best_obj.item(), best_idx.item())
    save("s6_mkp_n20", prize_in=prize, weight_in=weight, prize=aco.prize, weight=aco.weight, pheromone=tau0,
         heuristic=aco.heuristic, Q=aco.Q, noise=q, start=sols[0], sols=sols, log_probs=logp, objs=objs,
         decay=np.float32(aco.decay), pheromone_as=aco.pheromone)


def main():
    torch.set_num_threads(1)
    print("reference:", REF)
    tsp_aco = load_ref("tsp", "aco", "ref_tsp_aco")
    print("TSP sampler (G1)"); gen_tsp_sampler(tsp_aco)
    print("TSP update (G2)"); gen_tsp_update(tsp_aco)
    print("TSP run trace (U2)"); gen_tsp_run(tsp_aco)
    two_opt = load_ref("tsp_nls", "two_opt", "two_opt")          # name the reference imports
    nls_aco = load_ref("tsp_nls", "aco", "ref_nls_aco")
    print("tsp_nls (G1/G4/O4/G6)"); gen_nls(nls_aco, two_opt)
    cvrp_aco = load_ref("cvrp", "aco", "ref_cvrp_aco")
    print("CVRP (G1/G2)"); gen_cvrp(cvrp_aco)
    print("gradients (G3)"); gen_grads(tsp_aco, cvrp_aco)
    print("Net forward (G5)"); gen_net()
    print("Net training step (G7)"); gen_netgrad()
    print("bench weights"); gen_bench_weights()
    print("siblings (S1-S6)"); gen_siblings()


if __name__ == "__main__":
    main()
