#!/usr/bin/env python3
"""Train-mode (batch-statistics) forward of the HIP kernels against the reference fixtures g5_net_* (heu_train): the largest
absolute error and the share of elements beyond 1e-5 + 1e-4 |ref| -- what tests/test_gpu_07_net.py's tolerance is set from."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from test_net_host import make_net, load_weights, names  # noqa: E402
from deepaco_amd.net import GraphData  # noqa: E402

dev = torch.device("cuda:0")
for name in names("g5_net"):
    g = load_golden(name)
    if "heu_train" not in g:
        continue
    net = make_net(name)
    load_weights(net, g)
    net = net.to(dev).train()
    pyg = GraphData(x=torch.from_numpy(g["x"]), edge_index=torch.from_numpy(g["edge_index"]),
                    edge_attr=torch.from_numpy(g["edge_attr"])).to(dev)
    with torch.no_grad():
        heu = net(pyg).cpu().numpy().astype(np.float64).reshape(-1)
    ref = g["heu_train"].astype(np.float64).reshape(-1)
    err = np.abs(heu - ref)
    print(json.dumps({"fixture": name, "elements": int(err.size), "max_abs_err": float(err.max()),
                      "beyond_1e-5": int((err > 1e-5 + 1e-4 * np.abs(ref)).sum()), "ref_absmax": float(np.abs(ref).max())}), flush=True)
