"""Instance / graph construction with the surface of the reference's cvrp_nls/utils.py (H1): float64 instance data,
capacity normalised to 1, k-nearest-neighbour customer graph plus every depot edge."""
import os
import sys

import torch

try:
    from deepaco_amd.net import GraphData as Data
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd.net import GraphData as Data

CAPACITY_LIST = [(1, 10), (20, 30), (50, 40), (100, 50), (400, 150), (1000, 200), (2000, 300)]   # (from n nodes on, capacity)
DEMAND_LOW = 1
DEMAND_HIGH = 9


def get_capacity(n: int):
    """Vehicle capacity of an n-customer instance (cvrp_nls/utils.py:9-10): the last table row whose size is <= n."""
    return [cap for size, cap in CAPACITY_LIST if size <= n][-1]


def gen_distance_matrix(tsp_coordinates):
    """float64 Euclidean distances, 1e-10 on the diagonal (cvrp_nls/utils.py:28-32)."""
    n_nodes = len(tsp_coordinates)
    distances = torch.norm(tsp_coordinates[:, None] - tsp_coordinates, dim=2, p=2, dtype=torch.double)
    distances[torch.arange(n_nodes), torch.arange(n_nodes)] = 1e-10
    return distances


def gen_instance(n, device, position=False):
    """n + 1 uniform locations (node 0 = depot), integer demands 1..9 divided by the capacity of that size, all float64
    (cvrp_nls/utils.py:12-26).  Returns (demands [n+1], distances [n+1, n+1]) and, with position=True, the locations."""
    locations = torch.rand(size=(n + 1, 2), device=device, dtype=torch.double)
    demands = torch.randint(low=DEMAND_LOW, high=DEMAND_HIGH + 1, size=(n,), device=device, dtype=torch.double)
    all_demands = torch.cat((torch.zeros((1,), device=device, dtype=torch.double), demands / get_capacity(n)))
    distances = gen_distance_matrix(locations)
    return (all_demands, distances, locations) if position else (all_demands, distances)


def gen_pyg_data(demands, distances, device, k_sparse=5):
    """Sparse graph of cvrp_nls/utils.py:34-60: every customer's k nearest customers (edge list in customer order, then
    nearest first), then depot -> customer and customer -> depot for all customers; node feature = demand (float32)."""
    n = demands.size(0)
    near_d, near_i = torch.topk(distances[1:, 1:], k=k_sparse, dim=1, largest=False)
    customers = torch.arange(1, n, device=device, dtype=torch.long)
    knn = torch.stack((torch.repeat_interleave(torch.arange(n - 1).to(near_i.device), repeats=k_sparse),
                       torch.flatten(near_i))) + 1
    depot = torch.zeros(n - 1, device=device, dtype=torch.long)
    edge_index = torch.concat((knn, torch.stack((depot, customers)), torch.stack((customers, depot))), dim=1)
    to_depot = distances[1:, 0].reshape(-1, 1)
    edge_attr = torch.concat((near_d.reshape(-1, 1), to_depot, to_depot))
    return Data(x=demands.unsqueeze(1).float(), edge_attr=edge_attr.float(), edge_index=edge_index)


def _unpack(dataset, n_node, device):
    out = []
    for inst in dataset:
        demands, position, distances = inst[0, :], inst[1:3, :], inst[3:, :]
        out.append((gen_pyg_data(demands, distances, device, k_sparse=max(n_node // 5, 4)), demands, distances, position.T))
    return out


def load_test_dataset(n_node, k_sparse, device, start_node=None):
    """[(pyg_data, demands, distances, positions)] from ../data/cvrp_nls/testDataset-<n>.pt (rows: demand, x, y, distances)."""
    return _unpack(torch.load(f'../data/cvrp_nls/testDataset-{n_node}.pt', map_location=device), n_node, device)


def load_val_dataset(n_node, k_sparse, device, start_node=None):
    """As load_test_dataset for valDataset-<n>.pt; the file is generated (100 instances) if it does not exist
    (cvrp_nls/utils.py:71-90)."""
    path = f'../data/cvrp_nls/valDataset-{n_node}.pt'
    if not os.path.isfile(path):
        rows = []
        for _ in range(100):
            demand, dist, position = gen_instance(n_node, device, True)
            rows.append(torch.vstack([demand, position.T, dist]))
        dataset = torch.stack(rows)
        torch.save(dataset, path)
    else:
        dataset = torch.load(path, map_location=device)
    return _unpack(dataset, n_node, device)


if __name__ == "__main__":      # writes ../data/cvrp_nls/* as the reference's utils.py does when run as a script
    import sys
    from deepaco_amd.datasets import write_datasets
    print("\n".join(write_datasets("cvrp_nls", sys.modules[__name__])))
