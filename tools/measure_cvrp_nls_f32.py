#!/usr/bin/env python3
"""How often does float32 arithmetic change a cvrp_nls tour?  (This container only: imports the reference.)

cvrp_nls/ keeps its instance data in float64 (cvrp_nls/utils.py:19-30) and its sampler therefore draws in float64;
deepaco_amd's kernels compute in float32.  This script runs the REFERENCE's own gen_path (cvrp_nls/aco.py:205-232) twice
on the same instances and the same exponential noise -- once as it is (float64) and once with every input cast to
float32 (the arithmetic of the kernels, which the g1_cvrp fixtures hold bit-exact against the float32 reference) -- and
counts the ants whose route sequences differ.  The noise is injected by replacing torch.multinomial with the
arithmetic of its one-sample path (argmax(p / q), as tests/golden/gen_golden.py does), q drawn once in float64.
"""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(os.environ.get("DEEPACO_REFERENCE", "/root/reference"), "cvrp_nls")
LIB = os.path.join(ROOT, "oracle", "_ref", "libhgscvrp.so")
scratch = tempfile.mkdtemp(prefix="f32_")
os.makedirs(os.path.join(scratch, "HGS-CVRP-main", "build"))
os.symlink(LIB, os.path.join(scratch, "HGS-CVRP-main", "build", "libhgscvrp.so"))
os.chdir(scratch)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden", "shims"))
sys.path.insert(0, REF)
import aco as ref_aco  # noqa: E402
import utils as ref_utils  # noqa: E402


class Replay:
    """torch.multinomial(p, 1) = argmax(p / q): q recorded in float64 on the first run, replayed (cast) on the second."""

    def __init__(self):
        self.q, self.pos, self.record = [], 0, True

    def __call__(self, probs, num_samples, replacement=False, *, generator=None):
        if self.record:
            q = torch.empty(probs.shape, dtype=torch.float64).exponential_(1)
            self.q.append(q)
        elif self.pos < len(self.q):
            q = self.q[self.pos]
            self.pos += 1
        else:                                                   # the float32 run needs more steps than the float64 run did:
            q = torch.empty(probs.shape, dtype=torch.float64).exponential_(1)   # some ant already differs, fresh noise
        return torch.argmax(probs / q.to(probs.dtype), dim=-1, keepdim=True)


def main():
    n, A, instances = 100, 1000, 10
    differing = total = steps = 0
    first_step = []
    for inst in range(instances):
        torch.manual_seed(500 + inst)
        demands, distances = ref_utils.gen_instance(n, "cpu")
        heu = 1.0 / distances
        tap = Replay()
        orig = torch.multinomial
        torch.multinomial = tap
        try:
            p64 = ref_aco.ACO(distances, demands, n_ants=A, heuristic=heu).gen_path(require_prob=False)
            tap.record = False
            p32 = ref_aco.ACO(distances.float(), demands.float(), n_ants=A, heuristic=heu.float()).gen_path(require_prob=False)
        finally:
            torch.multinomial = orig
        L = min(p64.shape[0], p32.shape[0])
        diff = (p64[:L] != p32[:L]).any(dim=0)
        differing += int(diff.sum())
        total += A
        steps += L * A
        for a in torch.nonzero(diff).flatten().tolist():
            first_step.append(int(torch.nonzero(p64[:L, a] != p32[:L, a])[0]))
    print(f"cvrp_nls sampler, n = {n}, capacity 1.0 (demands k/50), {instances} instances x {A} ants, the reference's gen_path "
          f"with recorded Exp(1) noise:\n  float64 (as the reference runs it) vs float32 inputs (the kernels' arithmetic): "
          f"{differing} of {total} route sequences differ ({100.0 * differing / total:.3f} %), i.e. one draw in "
          f"{steps // max(differing, 1)} falls on a float32 rounding boundary (argmax of p/q, or demand > remaining capacity)"
          f"\n  first differing step of those ants: {sorted(first_step)[:20]}")


if __name__ == "__main__":
    main()
