"""Instance / graph construction with the surface of the reference's bpp/utils.py (bin packing, Falkenauer-style sizes)."""
import os
import sys

import torch

try:
    from deepaco_amd.net import GraphData as Data
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd.net import GraphData as Data

DEMAND_LOW = 20
DEMAND_HIGH = 100


def gen_instance(n, device):
    """Item sizes: integers in [20, 100], with a leading 0 for the dummy node (bpp/utils.py:9-12)."""
    demands = torch.randint(low=DEMAND_LOW, high=DEMAND_HIGH + 1, size=(n,), device=device)
    return torch.cat((torch.zeros((1,), device=device), demands))


def gen_pyg_data(demands, device='cpu'):
    """Complete graph with unit edge attributes, edge e = (e % N, e // N); node feature = item size (bpp/utils.py:14-23)."""
    n = demands.size(0)
    nodes = torch.arange(n, device=device)
    edge_index = torch.stack((nodes.repeat(n), torch.repeat_interleave(nodes, n)))
    return Data(x=demands.unsqueeze(1), edge_attr=torch.ones((edge_index.size(1), 1)), edge_index=edge_index)


def load_test_dataset(problem_size, device):
    return torch.load(f'../data/bpp/testDataset-{problem_size}.pt', map_location=device)


if __name__ == "__main__":      # writes ../data/bpp/* as the reference's utils.py does when run as a script
    import sys
    from deepaco_amd.datasets import write_datasets
    print("\n".join(write_datasets("bpp", sys.modules[__name__])))
