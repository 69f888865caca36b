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


def _tsp_nls_loss(net, coords, n_ants, k_sparse, seed, it, iter_dev, local_search):
    """The forward half of tsp_nls/train.py:15-44 for B instances: graphs -> Net (training mode) -> heuristic matrices ->
    B colonies sampled with log-probabilities -> NLS / 2-opt costs -> REINFORCE loss with the mixed baseline.
    iter_dev: optional int64 device scalar added to `it` inside the sampler (a captured step advances it).
    Returns (loss, mean sampled cost, mean locally-searched cost)."""
    from .autograd import TspBatchSampleFn
    B, n, _ = coords.shape
    dev = coords.device
    dist, ei, ea = engine.tsp_knn_graph(coords, k_sparse)
    x = torch.zeros((B, n, 1), device=dev)
    x[:, 0] = 1.0                                                     # tsp_nls/utils.py:38-44: one-hot of the start node
    heu = net.forward_batch_train(x, ei, ea, k_sparse=k_sparse)
    heu_mat = net.reshape_batch(n, ei, heu) + EPS
    tau = torch.ones((B, n, n), device=dev)
    paths, log_probs, flags = TspBatchSampleFn.apply(heu_mat, tau, n_ants, 1.0, 1.0, "scan", 2, 0, seed, it, iter_dev)
    with torch.no_grad():
        costs = engine.tour_costs(dist, paths)                        # [B, A]
        tours = paths.permute(0, 2, 1).to(torch.int16).contiguous()
        maxt = n // 4                                                 # tsp_nls/aco.py:235,242 (training)
        if local_search == "nls":
            h = heu_mat.detach()
            hdist = (1 / (h / h.amax(dim=-1, keepdim=True) + 1e-5)).contiguous()
            # (the perturbation matrix of a learned heuristic is not symmetric in general; handing its transpose over spares
            # nls_ the device comparison -- a host round trip -- that would find that out, and changes no result)
            tours = engine.nls_(dist, hdist, tours, maxt, dist_t="symmetric", heuristic_dist_t=hdist.transpose(1, 2).contiguous())
        else:
            engine.two_opt_(dist, tours, maxt, dist_t="symmetric")
        costs_ls = engine.tour_costs(dist, tours.permute(0, 2, 1).to(torch.int64).contiguous())
        cost = (costs_ls - costs_ls.mean(dim=1, keepdim=True)) * W_2OPT + (costs - costs.mean(dim=1, keepdim=True)) * (1 - W_2OPT)
    # sum over instances of sum_a cost_a * sum_t logp[t,a] / A, averaged over the batch (train.py:35-40)
    loss = torch.sum(cost.unsqueeze(1) * log_probs) / n_ants / B
    return loss, costs.mean(), costs_ls.mean()


def train_tsp_nls_batch(net, optimizer, coords, n_ants, k_sparse, seed=0, it=0, max_norm=3.0, local_search="nls"):
    """One optimisation step of tsp_nls/train.py:15-44 (`train_instance`) for a batch of B instances, on the device end
    to end: kNN graphs (one launch) -> Net in training mode (HIP kernels, per-graph BatchNorm statistics) -> heuristic
    matrices -> B colonies sampled with log-probabilities (one launch) -> NLS / 2-opt costs -> REINFORCE loss with the
    mixed baseline -> backward (sampler backward + GNN backward kernels) -> gradient clipping -> optimizer step.
    coords [B,n,2].  Returns (loss, mean sampled cost, mean locally-searched cost) as tensors.
    (After net.flatten_parameters() the clipping runs over the flat block: the same norm in one launch.)"""
    net.train()
    loss, c, c_ls = _tsp_nls_loss(net, coords, n_ants, k_sparse, seed, it, None, local_search)
    optimizer.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(parameters=net.train_parameters(), max_norm=max_norm, norm_type=2)
    optimizer.step()
    return loss.detach(), c, c_ls


class TspNlsTrainer:
    """The step of train_tsp_nls_batch as ONE captured HIP graph (the step was hundreds of launches of a few
    microseconds of work each, bound by launch latency on the host).

    The network's parameters become views of one flat block (Net.flatten_parameters), the optimizer -- AdamW as
    tsp_nls/train.py:95-100 builds it, here on the block: one fused elementwise update, capturable -- belongs to the trainer,
    and from the third call on `step(coords)` copies the coordinates into a static buffer and replays the graph: graph
    construction, the network's training forward, the colonies' construction with log-probabilities, the NLS, the loss, both
    backward passes, gradient clipping and the optimizer update, without a host round trip in between.  The sampler's Philox
    iteration counter lives in device memory and is advanced inside the graph, so step s draws what the eager
    train_tsp_nls_batch(..., seed, it=s) draws: the same tours, the same loss (tests/test_gpu_07_net.py).
    step() returns (loss, mean sampled cost, mean locally-searched cost) as device scalars that the next step overwrites."""

    def __init__(self, net, B, n, n_ants, k_sparse, lr=3e-4, seed=0, max_norm=3.0, local_search="nls", graph=True,
                 optimizer=None, device=None):
        dev = torch.device(device) if device is not None else next(net.parameters()).device
        if dev.type != "cuda":
            raise engine._lib.DacoError("TspNlsTrainer runs on a HIP device only")
        self.net, self.B, self.n, self.n_ants, self.k = net, int(B), int(n), int(n_ants), int(k_sparse)
        self.seed, self.max_norm, self.local_search, self.use_graph = int(seed), float(max_norm), local_search, bool(graph)
        self.block = net.flatten_parameters()
        # (capturable: the step count is a device tensor; fused: one launch for the whole update)
        self.optimizer = optimizer if optimizer is not None else torch.optim.AdamW([self.block], lr=lr, capturable=True, fused=True)
        self.coords = torch.zeros((self.B, self.n, 2), dtype=torch.float32, device=dev)
        self.it_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.steps_done = 0
        self._graph = None
        self._out = None
        self._side = torch.cuda.Stream(device=dev)
        self.dev = dev

    def _body(self):
        self.net.train()
        loss, c, c_ls = _tsp_nls_loss(self.net, self.coords, self.n_ants, self.k, self.seed, 0, self.it_dev, self.local_search)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([self.block], max_norm=self.max_norm, norm_type=2)
        self.optimizer.step()
        self.it_dev += 1
        return loss.detach(), c, c_ls

    def step(self, coords):
        cur = torch.cuda.current_stream(self.dev)
        self.coords.copy_(coords, non_blocking=True)
        if self._graph is not None:
            self._graph.replay()
        elif not self.use_graph or self.steps_done < 2:
            # eager (also the first two steps of a captured run: they size the workspaces and create the optimizer state) --
            # on the stream the graph will be captured on, so that the per-stream scratch is the same
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                self._out = self._body()
            cur.wait_stream(self._side)
        else:
            self._side.wait_stream(cur)
            g = torch.cuda.CUDAGraph()
            self.optimizer.zero_grad(set_to_none=True)
            with torch.cuda.graph(g, stream=self._side):
                self._out = self._body()
            cur.wait_stream(self._side)
            self._graph = g
            g.replay()                                                # (a capture records; this runs the step)
        self.steps_done += 1
        return self._out


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
