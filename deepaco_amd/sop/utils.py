"""Instance / graph construction with the surface of the reference's sop/utils.py (sequential ordering problem)."""
import pickle
import os
import sys

import torch

try:
    from deepaco_amd.net import GraphData as Data
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd.net import GraphData as Data


def ordering_constraint_gen(n, rand=0.2):
    """Random precedence pairs (i precedes j), transitively closed, node 0 before everything (sop/utils.py:5-21).
    One torch.rand draw per (i, j) candidate, in the reference's loop order."""
    pairs = [(0, i) for i in range(1, n)]
    after = [set() for _ in range(1, n)]                       # after[i]: jobs that must follow job i + 1
    for i in range(n - 3, -1, -1):
        for j in range(i + 1, n - 1):
            if torch.rand(size=(1,)) > rand:
                continue
            after[i].add(j)
            after[i].update(after[j])
        pairs.extend((i + 1, j + 1) for j in after[i])
    return pairs


def adjacency_mat_gen(n, r):
    """1 where the edge u -> v may be used: not a self loop, and v does not have to precede u (sop/utils.py:23-28)."""
    c = torch.ones(size=(n, n))
    c[torch.arange(n), torch.arange(n)] = 0
    for i, j in r:
        c[j][i] = 0
    return c


def preceding_mat_gen(n, r):
    """prec_mat[i, :] marks the nodes that must precede node i (sop/utils.py:30-37)."""
    prec_mat = torch.zeros(size=(n, n))
    for i, j in r:
        prec_mat[j, i] = 1
    return prec_mat


def cost_mat_gen(n):
    """U(0,1) set-up costs plus the processing cost of the job entered (row 0), for every row but the start's
    (sop/utils.py:39-43)."""
    distances = torch.rand(size=(n, n))
    distances[1:, :] += distances[0, :]
    return distances


def training_instance_gen(n, device):
    distance = cost_mat_gen(n).to(device)
    r = ordering_constraint_gen(n)
    mask = preceding_mat_gen(n, r).to(device)
    return distance, adjacency_mat_gen(n, r).to(device), mask


def gen_pyg_data(distances, adj, device):
    """Edges = the admissible transitions in row-major order, attribute = their cost; node feature = cost from the start
    node (sop/utils.py:52-57)."""
    return Data(x=distances[0, :].unsqueeze(-1), edge_index=torch.nonzero(adj).T,
                edge_attr=distances[adj.bool()].unsqueeze(-1))


def load_test_dataset(n_node, device):
    with open(f"../data/sop/test{n_node}.pkl", "rb") as f:
        loaded = pickle.load(f)
    return [[t.to(device) for t in inst] for inst in loaded]


if __name__ == "__main__":      # writes ../data/sop/* as the reference's utils.py does when run as a script
    import sys
    from deepaco_amd.datasets import write_datasets
    print("\n".join(write_datasets("sop", sys.modules[__name__])))
