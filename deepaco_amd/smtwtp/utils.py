"""Instance construction with the surface of the reference's smtwtp/utils.py (single-machine total weighted tardiness)."""
import pickle
import os
import sys

import torch

try:
    from deepaco_amd.net import GraphData as Data
except ImportError:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from deepaco_amd.net import GraphData as Data


def instance_gen(n, device):
    """(pyg_data, due_time [n], weights [n], processing_time [n]); job 0 of the graph is the dummy start.  Node features
    (due time / n, weight); complete graph whose edge e has attribute processing_time[e // (n+1)] and runs
    (e % (n+1)) -> (e // (n+1)) (smtwtp/utils.py:5-22).  Draws in the reference's order: due times, weights, processing."""
    due_norm = torch.rand(size=(n,), device=device)
    weights = torch.rand(size=(n,), device=device)
    processing_time = torch.rand(size=(n,), device=device)
    x = torch.cat((torch.zeros(size=(1, 2), device=device), torch.stack((due_norm, weights)).T), dim=0)
    padded = torch.cat((torch.zeros(size=(1,), device=device), processing_time))
    nodes = torch.arange(n + 1, device=device)
    edge_index = torch.stack((nodes.repeat(n + 1), torch.repeat_interleave(nodes, n + 1)))
    data = Data(x=x, edge_attr=torch.repeat_interleave(padded, n + 1).unsqueeze(-1), edge_index=edge_index)
    return data, due_norm * n, weights, processing_time


def load_test_dataset(n_node, device):
    with open(f"../data/smtwtp/test{n_node}.pkl", "rb") as f:
        loaded = pickle.load(f)
    return [[t.to(device) for t in inst] for inst in loaded]


if __name__ == "__main__":      # writes ../data/smtwtp/* as the reference's utils.py does when run as a script
    import sys
    from deepaco_amd.datasets import write_datasets
    print("\n".join(write_datasets("smtwtp", sys.modules[__name__])))
