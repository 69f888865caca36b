import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle
from deepaco_amd import engine
from test_gpu_00_tsp import make_instance
B, n, A = 4, 500, 512
dist, tau, eta = make_instance(n, 2024, B)
dev = torch.device("cuda:0")
paths, _, _, flags = engine.tsp_sample(tau.to(dev), eta.to(dev), A, mode="scan", seed=11, it=0)
print("flags", flags.tolist())
p = paths.cpu().numpy()
bad = [(b, a) for b in range(B) for a in range(A) if len(set(p[b, :, a])) != n]
print("ants with duplicates:", len(bad), bad[:20])
for b in range(B):
    P = oracle.prob_matrix(tau[b].numpy(), eta[b].numpy())
    rp, _, _ = oracle.tsp_sample_scan(P, A, 11, 0, b * A)
    diff = [a for a in range(A) if not np.array_equal(rp[:, a], p[b, :, a])]
    print("instance", b, "ants differing from oracle:", len(diff), diff[:20])
    for a in diff[:3]:
        t = int(np.argmax(rp[:, a] != p[b, :, a]))
        print("  ant", a, "first diff at step", t, "oracle", rp[t-1:t+2, a], "gpu", p[b, t-1:t+2, a])
b, a = 1, 45
P = oracle.prob_matrix(tau[b].numpy(), eta[b].numpy())
rp, _, _ = oracle.tsp_sample_scan(P, A, 11, 0, b * A)
print("positions of 495: gpu", np.where(p[b, :, a] == 495)[0], "oracle", np.where(rp[:, a] == 495)[0])
print("gpu 330..345", p[b, 330:345, a]); print("orc 330..345", rp[330:345, a])
t0 = int(np.where(rp[:, a] == 495)[0][0])
print("around oracle visit of 495:", rp[t0-2:t0+3, a], p[b, t0-2:t0+3, a])
# lane 27 candidates
lane = 27
cands = [c*128 + lane*4 + v for c in range(4) for v in range(4)]
print("lane-27 slots:", cands)
vis = set(rp[:337, a].tolist())
row = P[rp[336, a]]
print("row values / open:", [(k, float(row[k]), k not in vis) for k in cands])
