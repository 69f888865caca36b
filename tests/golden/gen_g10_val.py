#!/usr/bin/env python3
"""g10: the reference's TSP validation sets (data/tsp/valDataset-{20,100,500}.pt: 100 instances of coordinates each) as a
plain .npz, next to the numbers tsp/train.ipynb prints for them with the networks it then saves as pretrained/tsp/tsp*.pt
(cell outputs, final epoch: avg sample cost, best sample cost, best ACO cost after T = 5 iterations).  Data only.
tests/test_gpu_07_net.py runs the same protocol on the drop-in classes: SURVEY 8(c) names this as the end-to-end sanity
check of the torch_geometric stand-in the network fixtures were generated with.

Run:  python tests/golden/gen_g10_val.py
"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DEEPACO_REFERENCE", "/root/reference")
out = {}
for n in (20, 100, 500):
    out[f"coords{n}"] = torch.load(os.path.join(REF, "data", "tsp", f"valDataset-{n}.pt")).numpy()
# tsp/train.ipynb cell outputs, "epoch 4" lines: (avg sample obj., best sample obj., best ACO obj.)
out["notebook20"] = np.array([4.38464241027832, 3.9300931763648985, 3.84234961271286])
out["notebook100"] = np.array([9.711450233459473, 9.026623315811158, 8.63921733379364])
out["notebook500"] = np.array([21.94193012237549, 20.73219964981079, 19.791852359771728])
np.savez_compressed(os.path.join(HERE, "g10_val_tsp.npz"), **out)
print({k: v.shape for k, v in out.items()})
