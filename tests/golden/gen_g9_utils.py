#!/usr/bin/env python3
"""g9: captured I/O of the reference's per-problem utils.py (instance and graph construction) for
cvrp_nls and the six sibling directories, by importing them from /root/reference (this container only; torch_geometric
is the attribute-bag shim under tests/golden/shims).  Seeds are stored with the outputs: tests/test_utils_siblings.py
re-seeds torch / numpy and calls deepaco_amd/<problem>/utils.py the same way.

Run:  python tests/golden/gen_g9_utils.py   (writes tests/golden/g9_utils.npz)
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DEEPACO_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "shims"))


def load_ref(subdir):
    spec = importlib.util.spec_from_file_location(f"ref_{subdir}_utils", os.path.join(REF, subdir, "utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def graph(out, key, data):
    out[key + ".x"] = data.x.numpy()
    out[key + ".edge_index"] = data.edge_index.numpy()
    out[key + ".edge_attr"] = data.edge_attr.numpy()


def main():
    out = {}
    SEED = 2468
    # ---- cvrp_nls
    m = load_ref("cvrp_nls")
    out["cvrp_nls.capacity"] = np.array([m.get_capacity(n) for n in (1, 19, 20, 49, 50, 100, 399, 400, 1000, 2000, 5000)])
    for n in (20, 50):
        torch.manual_seed(SEED + n)
        dem, dist, pos = m.gen_instance(n, "cpu", True)
        out[f"cvrp_nls.n{n}.demands"], out[f"cvrp_nls.n{n}.distances"], out[f"cvrp_nls.n{n}.positions"] = dem.numpy(), dist.numpy(), pos.numpy()
        graph(out, f"cvrp_nls.n{n}.pyg", m.gen_pyg_data(dem, dist, "cpu", k_sparse=max(n // 5, 4)))
    # ---- op
    m = load_ref("op")
    torch.manual_seed(SEED)
    coor = torch.rand(30, 2)
    out["op.coor"] = coor.numpy()
    data, dist, prizes = m.gen_pyg_data(coor, 7)
    graph(out, "op.pyg", data)
    out["op.distances"], out["op.prizes"] = dist.numpy(), prizes.numpy()
    # ---- pctsp
    m = load_ref("pctsp")
    torch.manual_seed(SEED)
    dist, prizes, pen = m.gen_inst(20, "cpu")
    out["pctsp.dist"], out["pctsp.prizes"], out["pctsp.penalties"] = dist.numpy(), prizes.numpy(), pen.numpy()
    graph(out, "pctsp.pyg", m.gen_pyg_data(prizes, pen, dist))
    # ---- sop
    m = load_ref("sop")
    torch.manual_seed(SEED)
    dist, adj, mask = m.training_instance_gen(14, "cpu")
    out["sop.dist"], out["sop.adj"], out["sop.mask"] = dist.numpy(), adj.numpy(), mask.numpy()
    graph(out, "sop.pyg", m.gen_pyg_data(dist, adj, "cpu"))
    # ---- smtwtp
    m = load_ref("smtwtp")
    torch.manual_seed(SEED)
    data, due, wts, proc = m.instance_gen(9, "cpu")
    graph(out, "smtwtp.pyg", data)
    out["smtwtp.due"], out["smtwtp.weights"], out["smtwtp.processing"] = due.numpy(), wts.numpy(), proc.numpy()
    # ---- bpp
    m = load_ref("bpp")
    torch.manual_seed(SEED)
    dem = m.gen_instance(11, "cpu")
    out["bpp.demands"] = dem.numpy()
    graph(out, "bpp.pyg", m.gen_pyg_data(dem))
    # ---- mkp
    m = load_ref("mkp")
    torch.manual_seed(SEED)
    np.random.seed(SEED)
    prize, wm = m.gen_instance(12, 5, "cpu")
    out["mkp.prize"], out["mkp.weights"] = prize.numpy(), wm.numpy()
    graph(out, "mkp.pyg", m.gen_pyg_data(prize, wm))
    out["seed"] = np.int64(SEED)
    np.savez_compressed(os.path.join(HERE, "g9_utils.npz"), **out)
    print("g9_utils.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
