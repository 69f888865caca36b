"""deepaco_amd -- DeepACO's ant-rollout hot path on AMD Instinct MI355X (gfx950).

Python host over libdeepaco_hip.so (hand-written HIP kernels behind a C ABI, see
include/deepaco_hip.h).  The package keeps the reference's class surface:

    deepaco_amd.tsp.aco.ACO        <- tsp/aco.py        (henry-yeh/DeepACO)
    deepaco_amd.engine             batched (B instances) functional layer the classes sit on

There is no CPU fallback: tensors must live on a HIP device and the shared library must
have been built (python -c "import __graft_entry__ as g; g.build()").
"""
from . import _lib  # noqa: F401

__all__ = ["engine", "tsp"]
__version__ = "0.1.0"
