"""Net with one node feature (start-node one-hot in tsp_nls/, demand in cvrp/) and no par_net_phe
head (tsp_nls/net.py:78-83, cvrp/net.py).  Same import surface: `from net import Net`."""
import os
import sys

try:
    from deepaco_amd.net import Net as _Net, EmbNet, MLP, ParNet  # noqa: F401
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd.net import Net as _Net, EmbNet, MLP, ParNet  # noqa: F401


class Net(_Net):
    def __init__(self):
        super().__init__(feats=1, with_phe=False)
