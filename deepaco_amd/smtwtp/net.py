"""Net of smtwtp/net.py: two node features (normalised due time, weight), par_net_phe present, and NO node update
(smtwtp/net.py:42 is commented out in the reference).  `from net import Net`."""
import os
import sys

try:
    from deepaco_amd.net import Net as _Net, EmbNet, MLP, ParNet  # noqa: F401
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd.net import Net as _Net, EmbNet, MLP, ParNet  # noqa: F401


class Net(_Net):
    def __init__(self):
        super().__init__(feats=2, with_phe=True, node_update=False)
