"""Instance / graph construction with the surface of the reference's tsp/utils.py (H1)."""
import os
import sys

import torch

try:
    from deepaco_amd.net import GraphData as Data
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd.net import GraphData as Data


def gen_distance_matrix(tsp_coordinates):
    '''Euclidean distances [n, n] with 1e9 on the diagonal (tsp/utils.py:4-14).'''
    n_nodes = len(tsp_coordinates)
    distances = torch.norm(tsp_coordinates[:, None] - tsp_coordinates, dim=2, p=2)
    distances[torch.arange(n_nodes), torch.arange(n_nodes)] = 1e9
    return distances


def gen_pyg_data(tsp_coordinates, k_sparse, start_node=None):
    '''k-nearest-neighbour graph of one instance (tsp/utils.py:16-36, tsp_nls/utils.py:17-45):
    x = coordinates (or the start-node one-hot), edge_index [2, n*k] with sources sorted,
    edge_attr = the k smallest distances of each row.'''
    n_nodes = len(tsp_coordinates)
    distances = gen_distance_matrix(tsp_coordinates)
    topk_values, topk_indices = torch.topk(distances, k=k_sparse, dim=1, largest=False)
    edge_index = torch.stack([
        torch.repeat_interleave(torch.arange(n_nodes, device=topk_indices.device), repeats=k_sparse),
        torch.flatten(topk_indices)])
    edge_attr = topk_values.reshape(-1, 1)
    if start_node is None:
        x = tsp_coordinates
    else:
        x = torch.zeros((n_nodes, 1), device=tsp_coordinates.device, dtype=tsp_coordinates.dtype)
        x[start_node, 0] = 1.0
    return Data(x=x, edge_index=edge_index, edge_attr=edge_attr), distances


def gen_pyg_data_batch(coords, k_sparse, start_node=None):
    '''B instances at once on the device (one kernel launch): coords [B, n, 2] ->
    list of (Data, distances) exactly as gen_pyg_data returns them per instance.'''
    from deepaco_amd import engine
    dist, ei, ea = engine.tsp_knn_graph(coords, k_sparse)
    out = []
    for b in range(coords.shape[0]):
        if start_node is None:
            x = coords[b]
        else:
            x = torch.zeros((coords.shape[1], 1), device=coords.device, dtype=coords.dtype)
            x[start_node, 0] = 1.0
        out.append((Data(x=x, edge_index=ei[b], edge_attr=ea[b]), dist[b]))
    return out


def _load(path, k_sparse, device, start_node=None):
    out = []
    for instance in torch.load(path):
        out.append(gen_pyg_data(instance.to(device), k_sparse=k_sparse, start_node=start_node))
    return out


def load_val_dataset(n_node, k_sparse, device, start_node=None):
    return _load(f'../data/tsp/valDataset-{n_node}.pt', k_sparse, device, start_node)


def load_test_dataset(n_node, k_sparse, device, start_node=None, filename=None):
    return _load(filename or f'../data/tsp/testDataset-{n_node}.pt', k_sparse, device, start_node)
