"""Instance / graph construction with the surface of the reference's cvrp/utils.py (H1)."""
import os
import sys

import torch

try:
    from deepaco_amd.net import GraphData as Data
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd.net import GraphData as Data

CAPACITY = 50
DEMAND_LOW = 1
DEMAND_HIGH = 9
DEPOT_COOR = [0.5, 0.5]


def gen_distance_matrix(tsp_coordinates):
    '''Euclidean distances with 1e-10 on the diagonal (cvrp/utils.py:18-22).'''
    n_nodes = len(tsp_coordinates)
    distances = torch.norm(tsp_coordinates[:, None] - tsp_coordinates, dim=2, p=2)
    distances[torch.arange(n_nodes), torch.arange(n_nodes)] = 1e-10
    return distances


def gen_instance(n, device):
    '''Depot at (0.5, 0.5), n customers uniform in the unit square, integer demands 1..9
    (cvrp/utils.py:9-16).  Returns (demands [n+1], distances [n+1, n+1]).'''
    locations = torch.rand(size=(n, 2), device=device)
    demands = torch.randint(low=DEMAND_LOW, high=DEMAND_HIGH + 1, size=(n,), device=device)
    depot = torch.tensor([DEPOT_COOR], device=device)
    all_locations = torch.cat((depot, locations), dim=0)
    all_demands = torch.cat((torch.zeros((1,), device=device), demands))
    return all_demands, gen_distance_matrix(all_locations)


def gen_pyg_data(demands, distances, device):
    '''Complete graph: edge (u, v) for all pairs, u cycling fastest (cvrp/utils.py:24-33).'''
    n = demands.size(0)
    nodes = torch.arange(n, device=device)
    edge_index = torch.stack((nodes.repeat(n), torch.repeat_interleave(nodes, n)))
    return Data(x=demands.unsqueeze(1), edge_attr=distances.reshape((n ** 2, 1)), edge_index=edge_index)


def load_test_dataset(problem_size, device):
    dataset = torch.load(f'./data/cvrp/testDataset-{problem_size}.pt', map_location=device)
    return [(dataset[i, 0, :], dataset[i, 1:, :]) for i in range(len(dataset))]


if __name__ == "__main__":      # writes ../data/cvrp/* as the reference's utils.py does when run as a script
    import sys
    from deepaco_amd.datasets import write_datasets
    print("\n".join(write_datasets("cvrp", sys.modules[__name__])))
