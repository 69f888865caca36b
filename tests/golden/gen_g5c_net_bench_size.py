#!/usr/bin/env python3
"""G5c: the reference's heuristic network at the sizes bench.py runs it (VERDICT r5 missing 3 / next 4).

  g5c_net_tsp_tsp500       tsp/net.py Net + pretrained/tsp/tsp500.pt        n = 500,  k = 50   (tsp/train.ipynb:268)   E = 25 000
  g5c_net_tsp_nls_tsp1000  tsp_nls/net.py Net + pretrained/tsp_nls/tsp1000.pt  n = 1000, k = 100  (tsp_nls/test.py:50)  E = 100 000

The IMPORTED reference (PyG shims of gen_golden.py) builds the graph with its own utils.gen_pyg_data and evaluates
Net.forward in eval and in train mode (tsp/net.py:27-45,59-66,84-88).  Stored: the coordinates, the reference's edge list
(int16: n <= 1000) and edge attributes, heu[E] of both modes, and 2 048 evenly spaced rows of the embedding (the whole
[E, 32] would be 12 MB).  The weights are those of w_tsp_tsp500.npz / w_tsp_nls_tsp1000.npz (gen_golden.gen_bench_weights).
Run (this container only):  python tests/golden/gen_g5c_net_bench_size.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import REF, load_ref, save  # noqa: E402  (also puts the shims on sys.path)


def main():
    torch.set_num_threads(8)
    for sub, ck, n, k in (("tsp", "tsp500", 500, 50), ("tsp_nls", "tsp1000", 1000, 100)):
        net_mod = load_ref(sub, "net", f"ref_net_{sub}_c")
        utils = load_ref(sub, "utils", f"ref_utils_{sub}_c")
        sd = torch.load(os.path.join(REF, "pretrained", sub, ck + ".pt"), map_location="cpu")
        model = net_mod.Net()
        model.load_state_dict(sd)
        torch.manual_seed(2026)
        coords = torch.rand(n, 2)
        if sub == "tsp_nls":
            pyg, distances = utils.gen_pyg_data(coords, k_sparse=k, start_node=0)
        else:
            pyg, distances = utils.gen_pyg_data(coords, k_sparse=k)
        model.eval()
        with torch.no_grad():
            heu_eval = model(pyg)
            emb = model.emb_net(pyg.x, pyg.edge_index, pyg.edge_attr)
        model.train()
        with torch.no_grad():
            heu_train = model(pyg)
        E = pyg.edge_index.shape[1]
        rows = np.linspace(0, E - 1, 2048).astype(np.int64)
        assert int(pyg.edge_index.max()) < 32768
        save(f"g5c_net_{sub}_{ck}", coords=coords, x=pyg.x, edge_index=pyg.edge_index.to(torch.int16), edge_attr=pyg.edge_attr.view(-1),
             heu_eval=heu_eval.view(-1), heu_train=heu_train.view(-1), emb_rows=rows, emb_eval_rows=emb[rows], k_sparse=np.int32(k),
             diag=np.float32(distances[0, 0]))


if __name__ == "__main__":
    main()
