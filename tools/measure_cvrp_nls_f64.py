#!/usr/bin/env python3
"""After the float64 load bookkeeping: how many cvrp_nls routes does float32 PROBABILITY arithmetic still change?
(This container only: imports the reference.)

cvrp_nls/utils.py:19-30 keeps every tensor in float64, so cvrp_nls/aco.py:205-232 draws argmax(p / q) on float64 p.
The kernels (and oracle.cvrp_sample_noise with a float64 demand) keep the load bookkeeping in double and the
probabilities in float32.  This script runs the reference's gen_path on recorded Exp(1) noise (float64, as it is) and
replays the same noise (cast to float32, as the recorded-noise fixtures do) through the oracle's rule, and counts the
ants whose route sequences differ.  VERDICT r3 item 2: "commit the count that still differ"."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(os.environ.get("DEEPACO_REFERENCE", "/root/reference"), "cvrp_nls")
LIB = os.path.join(ROOT, "oracle", "_ref", "libhgscvrp.so")
scratch = tempfile.mkdtemp(prefix="f64_")
os.makedirs(os.path.join(scratch, "HGS-CVRP-main", "build"))
os.symlink(LIB, os.path.join(scratch, "HGS-CVRP-main", "build", "libhgscvrp.so"))
os.chdir(scratch)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden", "shims"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
import aco as ref_aco  # noqa: E402
import utils as ref_utils  # noqa: E402
import oracle  # noqa: E402


class Tap:
    def __init__(self):
        self.q = []

    def __call__(self, probs, num_samples, replacement=False, *, generator=None):
        q = torch.empty_like(probs).exponential_(1)
        self.q.append(q.clone())
        return torch.argmax(probs / q, dim=-1, keepdim=True)


def main():
    n, A, instances = 100, 1000, 10
    differing = total = steps = 0
    first = []
    for inst in range(instances):
        torch.manual_seed(500 + inst)
        demands, distances = ref_utils.gen_instance(n, "cpu")                 # float64
        g = torch.Generator().manual_seed(900 + inst)
        heu = (1.0 / distances).float() if inst % 2 == 0 else (torch.rand(n + 1, n + 1, generator=g) + 1e-5).float()
        tap = Tap()
        orig = torch.multinomial
        torch.multinomial = tap
        try:
            p64 = ref_aco.ACO(distances, demands, n_ants=A, heuristic=heu).gen_path(require_prob=False)
        finally:
            torch.multinomial = orig
        noise = torch.stack(tap.q).float().numpy()                             # [steps][A][n+1], the fixtures' cast
        P = oracle.prob_matrix(np.ones((n + 1, n + 1), dtype=np.float32), heu.numpy())
        paths, _, L = oracle.cvrp_sample_noise(P, demands.numpy(), float(ref_aco.CAPACITY), noise, require_prob=False)
        ref = p64.numpy()
        Lc = min(L, ref.shape[0])
        diff = (ref[:Lc] != paths[:Lc]).any(axis=0)
        if L != ref.shape[0]:
            longer = ref if ref.shape[0] > L else paths
            diff |= (longer[Lc:] != 0).any(axis=0)
        differing += int(diff.sum())
        total += A
        steps += ref.shape[0] * A
        for a in np.nonzero(diff)[0].tolist():
            first.append(int(np.nonzero(ref[:Lc, a] != paths[:Lc, a])[0][0]) if (ref[:Lc, a] != paths[:Lc, a]).any() else Lc)
    print(f"cvrp_nls sampler, n = {n}, capacity {ref_aco.CAPACITY} (demands k/50), {instances} instances x {A} ants (even instances: "
          f"heuristic 1/d, odd: random positive), the reference's float64 gen_path on recorded Exp(1) noise\n  vs the "
          f"float64-bookkeeping / float32-probability rule (oracle.cvrp_sample_noise, what the kernels compute) on the same noise "
          f"cast to float32:\n  {differing} of {total} route sequences differ ({100.0 * differing / total:.3f} %), "
          f"{steps} draws in all" + (f"; first differing steps: {sorted(first)[:20]}" if first else ""))


if __name__ == "__main__":
    main()
