"""Instance / graph construction with the surface of the reference's pctsp/utils.py (prize-collecting TSP)."""
import os
import sys

import torch

try:
    from deepaco_amd.net import GraphData as Data
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd.net import GraphData as Data

K_n = {20: 2, 100: 4, 500: 9}


def gen_prizes(n, device):
    """U(0,1) prizes, 0 for the depot (pctsp/utils.py:10-12)."""
    return torch.cat((torch.tensor([0.], device=device), torch.rand(size=(n,), device=device)))


def gen_penalties(n, device):
    """U(0,1) * 3 K_n / n penalties, 0 for the depot (pctsp/utils.py:14-17)."""
    beta = torch.rand(size=(n,), device=device) * 3 * K_n[n] / n
    return torch.cat((torch.tensor([0.], device=device), beta))


def gen_distance_matrix(coordinates):
    """Plain Euclidean distances (zero diagonal; pctsp/utils.py:19-22)."""
    return torch.norm(coordinates[:, None] - coordinates, dim=2, p=2)


def gen_inst(n, device):
    """(dist_mat [n+1, n+1], prizes [n+1], penalties [n+1]); draws in the reference's order: coordinates, prizes, penalties."""
    dist_mat = gen_distance_matrix(torch.rand((n + 1, 2), device=device))
    return dist_mat, gen_prizes(n, device), gen_penalties(n, device)


def gen_pyg_data(prizes, penalties, dist_mat):
    """Complete graph, edge e = (e // N, e % N) with the row-major distances as attributes; node features (prize, penalty)
    (pctsp/utils.py:31-40)."""
    n_nodes = prizes.size(0)
    nodes = torch.arange(n_nodes, device=prizes.device)
    edge_index = torch.stack((torch.repeat_interleave(nodes, n_nodes), nodes.repeat(n_nodes)))
    return Data(x=torch.stack((prizes, penalties)).permute(1, 0), edge_index=edge_index,
                edge_attr=dist_mat.reshape(-1,).unsqueeze(-1))


def load_test_dataset(n_node, device):
    """[(dist_mat, prizes, penalties)] from ./data/pctsp/testDataset-<n>.pt (rows: distances, then prizes, then penalties)."""
    dataset = torch.load(f'./data/pctsp/testDataset-{n_node}.pt', map_location=device)
    return [(inst[:-2], inst[-2], inst[-1]) for inst in dataset]


if __name__ == "__main__":      # writes ../data/pctsp/* as the reference's utils.py does when run as a script
    import sys
    from deepaco_amd.datasets import write_datasets
    print("\n".join(write_datasets("pctsp", sys.modules[__name__])))
