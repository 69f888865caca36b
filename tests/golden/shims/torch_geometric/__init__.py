"""Oracle-side API shim for `torch_geometric` (absent in this image; README pins 2.0.4).

Test infrastructure only -- lets tests/golden/gen_golden.py import the reference's
net.py / utils.py.  Only the three symbols the reference uses are provided
(net.py:15,21 and utils.py:2).  Never imported by the product package.
"""
from . import nn, data  # noqa: F401
