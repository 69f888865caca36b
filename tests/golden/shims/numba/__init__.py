"""Oracle-side API shim for `numba` (absent in this image; README pins 0.56.4).

Test infrastructure only -- used by tests/golden/gen_golden.py when importing the
reference's tsp_nls/{aco,two_opt}.py.  numba is only a JIT: the semantics of a
jitted function are those of its Python source, so identity decorators preserve
the arithmetic (numpy float32 scalars add/subtract in float32, left to right).
Never imported by the product package.
"""
import numpy as _np


class _Type:
    def __init__(self, name):
        self.name = name

    def __getitem__(self, item):          # nb.float32[:, :]
        return self

    def __call__(self, *a, **k):          # nb.float32(nb.float32[:, :], ...)
        return self


float32 = _Type("float32")
float64 = _Type("float64")
uint8 = _Type("uint8")
uint16 = _Type("uint16")
int32 = _Type("int32")
int64 = _Type("int64")


def _identity_decorator(*dargs, **dkwargs):
    if len(dargs) == 1 and callable(dargs[0]) and not isinstance(dargs[0], _Type) and not dkwargs:
        return dargs[0]

    def wrap(fn):
        return fn
    return wrap


njit = _identity_decorator
jit = _identity_decorator
