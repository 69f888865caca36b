import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def assert_close_mostly(got, ref, atol=1e-5, rtol=1e-4, frac=1e-4, cap=1e-4, err_msg=""):
    """|got - ref| <= atol + rtol |ref| for all but a fraction `frac` of the elements, and <= cap + rtol |ref| for every one.
    For the heuristic network at E = 25 000 ... 100 000 edges: twelve residual layers leave a handful of mid-sigmoid outputs a few
    1e-5 apart between ANY two float32 evaluations (the float64 restatement against the reference's own float32 output: 2 of
    100 000 elements, 3.9e-5), so SURVEY G5's 1e-5 is held for 99.99 % of the elements and 1e-4 for all of them."""
    got, ref = np.asarray(got, dtype=np.float64).reshape(-1), np.asarray(ref, dtype=np.float64).reshape(-1)
    err = np.abs(got - ref)
    over = err > atol + rtol * np.abs(ref)
    assert err.size == ref.size and over.mean() <= frac, (err_msg, int(over.sum()), err.size, float(err.max()))
    assert (err <= cap + rtol * np.abs(ref)).all(), (err_msg, float(err.max()))
