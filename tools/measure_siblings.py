#!/usr/bin/env python3
"""Solutions/s of the sibling constructions: fused one-launch kernels vs the draw-by-draw service."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    A, n = 512, 100
    g = torch.Generator().manual_seed(0)
    out = []
    from deepaco_amd.smtwtp.aco import ACO as SMTWTP
    a = SMTWTP((torch.rand(n, generator=g) * n).to(dev), torch.rand(n, generator=g).to(dev), torch.rand(n, generator=g).to(dev), n_ants=A, device="cuda:0")
    out.append(("smtwtp", timeit(lambda: a.gen_path()), timeit(lambda: a.gen_path(_stepwise=True), 2)))
    from deepaco_amd.sop.aco import ACO as SOP
    dist = torch.rand(n, n, generator=g) + 0.05
    prec = torch.zeros(n, n); prec[1:, 0] = 1
    a = SOP(dist.to(dev), prec.to(dev), n_ants=A, device="cuda:0")
    out.append(("sop", timeit(lambda: a.gen_path()), timeit(lambda: a.gen_path(_stepwise=True), 2)))
    from deepaco_amd.pctsp.aco import ACO as PCTSP
    coor = torch.rand(n + 1, 2, generator=g); d = torch.cdist(coor, coor)
    a = PCTSP(d.to(dev), torch.cat((torch.zeros(1), torch.rand(n, generator=g))).to(dev), torch.cat((torch.zeros(1), torch.rand(n, generator=g) * 0.12)).to(dev), n_ants=A, device="cuda:0")
    out.append(("pctsp", timeit(lambda: a.gen_sol()), timeit(lambda: a.gen_sol(_stepwise=True), 2)))
    from deepaco_amd.op.aco import ACO as OP
    from deepaco_amd.tsp.utils import gen_distance_matrix
    coor = torch.rand(n, 2, generator=g); dd = (coor - coor[0]).norm(dim=-1); pr = 1 + torch.floor(99 * dd / dd.max()); pr = pr / pr.max()
    a = OP(gen_distance_matrix(coor).to(dev), pr.to(dev), 4.0, n_ants=A, k_sparse=20, device="cuda:0")
    out.append(("op", timeit(lambda: a.gen_sol()), timeit(lambda: a.gen_sol(_stepwise=True), 2)))
    from deepaco_amd.mkp.aco import ACO as MKP
    w = torch.rand(n, 5, generator=g); cons = w.max(0).values + torch.rand(5, generator=g) * (w.sum(0) - w.max(0).values); w = w * (n // 2) / cons
    a = MKP(torch.rand(n, generator=g).to(dev), w.to(dev), n_ants=A, device="cuda:0")
    out.append(("mkp", timeit(lambda: a.gen_sol()), timeit(lambda: a.gen_sol(_stepwise=True), 2)))
    from deepaco_amd.bpp.aco import ACO as BPP
    dem = torch.cat((torch.zeros(1), torch.randint(20, 101, (120,), generator=g).float()))
    a = BPP(dem.to(dev), n_ants=A, device="cuda:0")
    out.append(("bpp", timeit(lambda: a.gen_path()), None))
    for name, tf, ts in out:
        print(json.dumps({"problem": name, "n": n, "ants": A, "fused_ms": tf * 1e3, "fused_solutions_per_s": A / tf,
                          "stepwise_ms": None if ts is None else ts * 1e3}))


if __name__ == "__main__":
    main()
