#!/bin/bash
# Round 6, GPU session U: soaks of what the second half of the round touched: the route-exact local search in latency mode,
# the head-row sampler (loop index no longer masked), the captured training step over many replays.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06u
mkdir -p $OUT
cd $R
DACO_HGS_LATENCY=1 timeout 200 python tools/soak_hgs_ls.py 90 11 > $OUT/soak_hgs_latency.txt 2>&1; tail -2 $OUT/soak_hgs_latency.txt | cut -c1-400
timeout 200 python tools/soak_scan_sparse.py 60 9 > $OUT/soak_scan_sparse.txt 2>&1; tail -1 $OUT/soak_scan_sparse.txt | cut -c1-400
timeout 300 python - > $OUT/soak_trainer.txt 2>&1 <<'PY'
import json, sys, os, time, torch
sys.path.insert(0, os.getcwd())
from deepaco_amd.pipeline import TspNlsTrainer
from deepaco_amd.tsp_nls.net import Net
dev = torch.device("cuda:0")
torch.manual_seed(3)
out = {}
for (B, n, A, k) in ((20, 100, 30, 10), (4, 200, 20, 20)):
    net = Net().to(dev)
    tr = TspNlsTrainer(net, B, n, A, k, lr=3e-4, seed=2, graph=True)
    hist = []
    t0 = time.time()
    for s in range(400):
        loss, c, cls = tr.step(torch.rand(B, n, 2, device=dev))
        if s % 50 == 49:
            hist.append((round(float(c), 4), round(float(cls), 4)))
    ok = all(bool(torch.isfinite(p).all()) for p in net.parameters())
    out[f"tsp{n}"] = {"steps": 400, "seconds": round(time.time() - t0, 2), "finite": ok, "it_dev": int(tr.it_dev), "graph": tr._graph is not None,
                      "mean_cost_every_50": hist, "block_is_module": bool(torch.equal(net.pack_params_train().detach(), tr.block.detach()))}
print(json.dumps(out))
PY
tail -1 $OUT/soak_trainer.txt | cut -c1-900
