"""Net of pctsp/net.py: two node features (prize, penalty), par_net_phe present.  `from net import Net`."""
import os
import sys

try:
    from deepaco_amd.net import Net as _Net, EmbNet, MLP, ParNet  # noqa: F401
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd.net import Net as _Net, EmbNet, MLP, ParNet  # noqa: F401


class Net(_Net):
    def __init__(self):
        super().__init__(feats=2, with_phe=True)
