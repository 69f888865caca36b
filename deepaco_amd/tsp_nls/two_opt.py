"""2-opt with the call surface of the reference's tsp_nls/two_opt.py, running on MI355X.

`batched_two_opt_python(dist, tours, max_iterations)` keeps the reference's name and argument
meaning (tsp_nls/two_opt.py:41-49: `dist` [n, n], `tours` [count, n] one ROW per tour, returns
the improved tours).  numpy inputs are accepted like in the reference (host buffers are copied
to the current HIP device and the result copied back); torch tensors on the device are
processed in place without leaving the GPU.  Results are bit-identical to the reference's
numba code (same f32 expression order and tie-breaking), see csrc/daco_two_opt.hip.
"""
import os
import sys

import numpy as np
import torch

try:
    from deepaco_amd import engine
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd import engine


def two_opt_device(dist, tours_i16, max_iterations=1000, dist_t=None, tables=None):
    """dist [n,n] or [B,n,n] f32 on device, tours [T,n] or [B,T,n] int16 on device -> new tensor.
    dist_t / tables: engine.two_opt_'s (how the kernels read the matrix; never the result)."""
    out = tours_i16.clone().contiguous()
    engine.two_opt_(dist, out, max_iterations, dist_t=dist_t, tables=tables)
    return out


def batched_two_opt_python(dist, tours, max_iterations=1000):
    if isinstance(dist, np.ndarray) or isinstance(tours, np.ndarray):
        dev = torch.device("cuda", torch.cuda.current_device())
        d = torch.as_tensor(np.asarray(dist, dtype=np.float32)).to(dev)
        t = torch.as_tensor(np.asarray(tours).astype(np.int16)).to(dev)
        out = two_opt_device(d, t, max_iterations)
        return out.cpu().numpy().astype(np.uint16)
    return two_opt_device(dist.float(), tours.to(torch.int16), max_iterations)
