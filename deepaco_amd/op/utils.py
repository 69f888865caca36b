"""Instance / graph construction with the surface of the reference's op/utils.py (orienteering: node 0 = depot)."""
import os
import sys

import torch

try:
    from deepaco_amd.net import GraphData as Data
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd.net import GraphData as Data


def gen_prizes(coordinates):
    """Prize grows with the distance to the depot: 1 + floor(99 d / d_max), scaled to (0, 1] (op/utils.py:5-11)."""
    away = (coordinates - coordinates[0]).norm(p=2, dim=-1)
    prizes = 1 + torch.floor(99 * away / away.max())
    return prizes / prizes.max()


def gen_distance_matrix(coordinates):
    """Euclidean distances with 1e9 on the diagonal (op/utils.py:13-23)."""
    n_nodes = len(coordinates)
    distances = torch.norm(coordinates[:, None] - coordinates, dim=2, p=2)
    distances[torch.arange(n_nodes), torch.arange(n_nodes)] = 1e9
    return distances


def gen_pyg_data(tsp_coordinates, k_sparse):
    """(pyg_data, distances, prizes): node features (distance to the depot, prize), k nearest neighbours per node
    (op/utils.py:26-49)."""
    n_nodes = len(tsp_coordinates)
    prizes = gen_prizes(tsp_coordinates)
    x = torch.stack(((tsp_coordinates - tsp_coordinates[0]).norm(dim=-1), prizes)).T
    distances = gen_distance_matrix(tsp_coordinates)
    near_d, near_i = torch.topk(distances, k=k_sparse, dim=1, largest=False)
    edge_index = torch.stack((torch.repeat_interleave(torch.arange(n_nodes).to(near_i.device), repeats=k_sparse),
                              torch.flatten(near_i)))
    return Data(x=x, edge_index=edge_index, edge_attr=near_d.reshape(-1, 1)), distances, prizes


def _load(path, k_sparse, device):
    return [gen_pyg_data(coor.to(device), k_sparse=k_sparse) for coor in torch.load(path)]


def load_val_dataset(n_node, k_sparse, device):
    """[(pyg_data, distances, prizes)] from ../data/op/valDataset-<n>.pt (a tensor of coordinates)."""
    return _load(f'../data/op/valDataset-{n_node}.pt', k_sparse, device)


def load_test_dataset(n_node, k_sparse, device):
    return _load(f'./data/op/testDataset-{n_node}.pt', k_sparse, device)


if __name__ == "__main__":      # writes ../data/op/* as the reference's utils.py does when run as a script
    import sys
    from deepaco_amd.datasets import write_datasets
    print("\n".join(write_datasets("op", sys.modules[__name__])))
