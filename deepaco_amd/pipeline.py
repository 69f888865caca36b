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
                    sampler="auto", seed=0, **aco_kw):
    """coords [B,n,2] on a HIP device; t_aco: iteration checkpoints, e.g. [1, 10, 20] (the reference's schedule
    `t_aco_diff`); net: a deepaco_amd Net in eval mode or None for the vanilla heuristic 1/d on the kNN edges
    (ACO.sparsify, tsp/aco.py:52-67); node_feature: 'coords' (tsp/) or 'onehot0' (tsp_nls/utils.py:38-44: a
    one-hot of the start node); sampler: BatchedTSP's ("auto": the network's k-sparse heuristic and the sparsified 1/d both
    run on head / tail rows for 129 <= n <= 1024).  Returns (best costs [len(t_aco), B], colony)."""
    B, n, _ = coords.shape
    dist, ei, ea = engine.tsp_knn_graph(coords, k_sparse)
    heuristic = None
    if net is not None:
        if node_feature == "coords":
            x = coords
        else:
            x = torch.zeros((B, n, 1), device=coords.device)
            x[:, 0] = 1.0
        heu = net.forward_batch(x, ei, ea, k_sparse=k_sparse)
        heuristic = net.reshape_batch(n, ei, heu, eps=EPS)
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


W_2OPT = 0.95      # tsp_nls/train.py:13: weight of the locally-searched costs in the REINFORCE signal


def train_tsp_nls_batch(net, optimizer, coords, n_ants, k_sparse, seed=0, it=0, max_norm=3.0, local_search="nls"):
    """One optimisation step of tsp_nls/train.py:15-44 (`train_instance`) for a batch of B instances, on the device end
    to end: kNN graphs (one launch) -> Net in training mode (HIP kernels, per-graph BatchNorm statistics) -> heuristic
    matrices -> B colonies sampled with log-probabilities (one launch) -> NLS / 2-opt costs -> REINFORCE loss with the
    mixed baseline -> backward (sampler backward + GNN backward kernels) -> gradient clipping -> optimizer step.
    coords [B,n,2].  Returns (loss, mean sampled cost, mean locally-searched cost) as tensors."""
    from .autograd import TspBatchSampleFn
    B, n, _ = coords.shape
    dev = coords.device
    net.train()
    dist, ei, ea = engine.tsp_knn_graph(coords, k_sparse)
    x = torch.zeros((B, n, 1), device=dev)
    x[:, 0] = 1.0                                                     # tsp_nls/utils.py:38-44: one-hot of the start node
    heu = net.forward_batch_train(x, ei, ea, k_sparse=k_sparse)
    heu_mat = net.reshape_batch(n, ei, heu) + EPS
    tau = torch.ones((B, n, n), device=dev)
    paths, log_probs, flags = TspBatchSampleFn.apply(heu_mat, tau, n_ants, 1.0, 1.0, "scan", 2, 0, seed, it)
    with torch.no_grad():
        costs = engine.tour_costs(dist, paths)                        # [B, A]
        tours = paths.permute(0, 2, 1).to(torch.int16).contiguous()
        maxt = n // 4                                                 # tsp_nls/aco.py:235,242 (training)
        if local_search == "nls":
            h = heu_mat.detach()
            hdist = (1 / (h / h.amax(dim=-1, keepdim=True) + 1e-5)).contiguous()
            tours = engine.nls_(dist, hdist, tours, maxt, dist_t="symmetric")
        else:
            engine.two_opt_(dist, tours, maxt, dist_t="symmetric")
        costs_ls = engine.tour_costs(dist, tours.permute(0, 2, 1).to(torch.int64).contiguous())
        cost = (costs_ls - costs_ls.mean(dim=1, keepdim=True)) * W_2OPT + (costs - costs.mean(dim=1, keepdim=True)) * (1 - W_2OPT)
    # sum over instances of sum_a cost_a * sum_t logp[t,a] / A, averaged over the batch (train.py:35-40)
    loss = torch.sum(cost.unsqueeze(1) * log_probs) / n_ants / B
    optimizer.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(parameters=net.parameters(), max_norm=max_norm, norm_type=2)
    optimizer.step()
    return loss.detach(), costs.mean(), costs_ls.mean()


@torch.no_grad()
def cvrp_nls_graph_batch(demands, distances, k_sparse):
    """cvrp_nls/utils.py:34-60 (gen_pyg_data) for B instances at once: every customer's k nearest customers (customer order,
    nearest first), then depot -> customer and customer -> depot; node feature = demand.  demands [B,n+1], distances
    [B,n+1,n+1] -> (x [B,n+1,1] f32, edge_index [B,2,E] int64 with ids local to each graph, edge_attr [B,E,1] f32)."""
    B, n1 = demands.shape
    dev = demands.device
    near_d, near_i = torch.topk(distances[:, 1:, 1:], k=k_sparse, dim=2, largest=False)
    src = torch.repeat_interleave(torch.arange(1, n1, device=dev), repeats=k_sparse).unsqueeze(0).expand(B, -1)
    knn = torch.stack((src, near_i.reshape(B, -1) + 1), dim=1)                                  # [B, 2, n k]
    customers = torch.arange(1, n1, device=dev).unsqueeze(0).expand(B, -1)
    depot = torch.zeros_like(customers)
    edge_index = torch.cat((knn, torch.stack((depot, customers), dim=1), torch.stack((customers, depot), dim=1)), dim=2)
    to_depot = distances[:, 1:, 0]
    edge_attr = torch.cat((near_d.reshape(B, -1), to_depot, to_depot), dim=1).unsqueeze(2)
    return demands.unsqueeze(2).float(), edge_index.contiguous(), edge_attr.float().contiguous()


@torch.no_grad()
def infer_cvrp_nls_batch(locations, demands, n_ants, t_aco, k_sparse, net=None, seed=0, ls_ants=8, **aco_kw):
    """The batched counterpart of cvrp_nls/test.py:40-60 (`infer_instance`): locations [B,n+1,2] (node 0 = depot) and
    demands [B,n+1] normalised to capacity 1, float64 as cvrp_nls/utils.py:12-26 builds them -> distances -> sparse graph ->
    heuristic (the network in eval mode, + 1e-10; None: 1 / distance) -> B colonies whose `ls_ants` cheapest ants go through the
    reference's local search every iteration (engine.BatchedCVRP(local_search="hgs", inference=True)).
    Returns (best costs [len(t_aco), B], colony); colony.shortest_path holds the best route sequences."""
    B, n1, _ = locations.shape
    loc = locations.double()
    dist = torch.norm(loc[:, :, None] - loc[:, None], dim=3, p=2)
    ar = torch.arange(n1, device=loc.device)
    dist[:, ar, ar] = 1e-10                                                                         # cvrp_nls/utils.py:28-32
    heuristic = None
    if net is not None:
        x, ei, ea = cvrp_nls_graph_batch(demands, dist, k_sparse)
        heu = net.forward_batch(x, ei, ea)
        heuristic = net.reshape_batch(n1, ei, heu, eps=EPS)
    colony = engine.BatchedCVRP(dist, demands.double(), n_ants=n_ants, capacity=1.0, heuristic=heuristic, seed=seed,
                                local_search="hgs", ls_ants=ls_ants, inference=True, **aco_kw)
    out, done = [], 0
    for t in t_aco:
        colony.run(t - done)
        done = t
        out.append(colony.lowest_cost.clone())
    return torch.stack(out), colony
