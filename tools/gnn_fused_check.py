#!/usr/bin/env python3
"""Net.forward_batch (eval) on 64 graphs of TSP-500 (k = 50) with the fused layer kernel (DACO_GNN_FUSED_NPW = -1: chosen to fill the device, or a fixed count of nodes
per wave) and with the split edge | node kernels (0): time per forward and the largest difference of the heuristic between
the fused and the split path.  One subprocess per setting (the knob is read once per process)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(out_path):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from deepaco_amd import engine
    from deepaco_amd.tsp.net import Net
    dev = torch.device("cuda:0")
    B, n, k = int(os.environ.get("GNN_B", 64)), int(os.environ.get("GNN_N", 500)), int(os.environ.get("GNN_K", 50))
    torch.manual_seed(0)
    net = Net().to(dev).eval()
    coords = torch.rand(B, n, 2, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    _, ei, ea = engine.tsp_knn_graph(coords, k, want_dist=False)
    heu = net.forward_batch(coords, ei, ea, k_sparse=k)
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        heu = net.forward_batch(coords, ei, ea, k_sparse=k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    np.save(out_path, heu.cpu().numpy())
    print(json.dumps({"npw": os.environ.get("DACO_GNN_FUSED_NPW"), "ms_per_forward": ms, "graphs": B, "n": n, "k": k}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2])
        sys.exit(0)
    import numpy as np
    outs = {}
    for npw in ("0", "-1", "8", "11", "16"):
        path = f"/tmp/gnn_fused_{npw}.npy"
        env = dict(os.environ, DACO_GNN_FUSED_NPW=npw)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", path], env=env, capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-2000:])
        if os.path.exists(path):
            outs[npw] = np.load(path)
    ref = outs.get("0")
    for npw, h in outs.items():
        if ref is not None and npw != "0":
            print(json.dumps({"npw": npw, "max_abs_diff_vs_split": float(np.abs(h - ref).max()), "finite": bool(np.isfinite(h).all())}))
