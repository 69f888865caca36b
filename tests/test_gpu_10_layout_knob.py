"""The scan draw's measurement knob DACO_SCAN_LAYOUT (4 | 8 | 16 lanes per ant for small n): the library reads it once per
process and the oracle honours it, so every layout of scan16_kernel -- not only the default rule's -- is held bit-exact
against the restatement.  Each layout runs in its own process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import numpy as np, torch, oracle
from deepaco_amd import engine
dev = torch.device("cuda:0")
seed, it = 424242, 2
for n, A in ((100, 33), (50, 17), (128, 9), (17, 40), (200, 5), (256, 12)):
    g = torch.Generator().manual_seed(n)
    tau = torch.rand(1, n, n, generator=g) + 0.1
    eta = torch.rand(1, n, n, generator=g) ** 2 + 1e-10
    paths, logp, _, flags = engine.tsp_sample(tau.to(dev), eta.to(dev), A, mode="scan", seed=seed, it=it, require_prob=True)
    P = oracle.prob_matrix(tau[0].numpy(), eta[0].numpy())
    rp, rl, rc = oracle.tsp_sample_scan(P, A, seed, it, require_prob=True)
    assert rc == 0 and int(flags.sum()) == 0
    assert np.array_equal(paths[0].cpu().numpy(), rp), ("tsp", n)
    np.testing.assert_allclose(logp[0].cpu().numpy(), rl, atol=3e-6, rtol=1e-5)
    dem = torch.cat((torch.zeros(1), torch.randint(1, 10, (n - 1,), generator=g).float()))
    cp, _, _, lens, cf = engine.cvrp_sample(tau.to(dev), eta.to(dev), dem.to(dev), 40.0, A, seed=seed, it=it)
    rp, _, L = oracle.cvrp_sample_rng(P, dem.numpy(), 40.0, A, "scan", seed, it)
    assert int(cf.sum()) == 0 and L == int(lens.max()) and np.array_equal(cp[0, :L].cpu().numpy(), rp), ("cvrp", n)
print("layout ok")
"""


@pytest.mark.parametrize("layout", ["4", "8", "16"])
def test_every_small_n_layout_equals_the_oracle(layout):
    env = dict(os.environ, DACO_SCAN_LAYOUT=layout, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, "-c", SCRIPT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "layout ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
