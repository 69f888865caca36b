"""DeepACO inference for a batch of TSP instances, end to end on the GPU.

The batched counterpart of `infer_instance` in the reference's test harnesses
(tsp/test.ipynb:31-52, tsp_nls/test.py:16-34): coordinates -> kNN graph -> heuristic (the GNN, or
the vanilla 1/d) -> ACO colonies.  Every stage is one pass of the HIP kernels over all B instances
and nothing synchronises with the host until the caller reads the result.
"""
import torch

from . import engine

EPS = 1e-10        # added to the learned heuristic (tsp/train.ipynb:35, tsp_nls/test.py:28)


@torch.no_grad()
def infer_tsp_batch(coords, n_ants, t_aco, k_sparse, net=None, node_feature="coords", local_search=None,
                    sampler="scan", seed=0, **aco_kw):
    """coords [B,n,2] on a HIP device; t_aco: iteration checkpoints, e.g. [1, 10, 20] (the reference's schedule
    `t_aco_diff`); net: a deepaco_amd Net in eval mode or None for the vanilla heuristic 1/d on the kNN edges
    (ACO.sparsify, tsp/aco.py:52-67); node_feature: 'coords' (tsp/) or 'onehot0' (tsp_nls/utils.py:38-44: a
    one-hot of the start node).  Returns (best costs [len(t_aco), B], colony)."""
    B, n, _ = coords.shape
    dist, ei, ea = engine.tsp_knn_graph(coords, k_sparse)
    heuristic = None
    if net is not None:
        if node_feature == "coords":
            x = coords
        else:
            x = torch.zeros((B, n, 1), device=coords.device)
            x[:, 0] = 1.0
        heu = net.forward_batch(x, ei, ea)
        heuristic = net.reshape_batch(n, ei, heu) + EPS
    colony = engine.BatchedTSP(dist, n_ants=n_ants, heuristic=heuristic, sampler=sampler, seed=seed,
                               local_search=local_search, fixed_start=0 if local_search else -1,
                               inference=True,      # tsp_nls/test.py:30 aco.run(t, inference=True): 2-opt to convergence
                               **aco_kw)
    if net is None:
        colony.sparsify(k_sparse)
    out, done = [], 0
    for t in t_aco:
        colony.run(t - done)
        done = t
        out.append(colony.lowest_cost.clone())
    return torch.stack(out), colony
