"""Heuristic network (edge GNN + MLP head) with the module tree of the reference's net.py.

`Net` keeps the reference's parameter names, so its checkpoints load unchanged
(tsp/net.py:78-102: emb_net.{v_lin0, v_lins1..4.{i}, v_bns.{i}.module.*, e_lin0, e_lins0.{i},
e_bns.{i}.module.*}, par_net_heu.{_dummy, lins.{i}}, and the unused par_net_phe of tsp/).

forward(pyg):
  * inference (module in eval mode and no gradient required) -> one call into
    libdeepaco_hip.so (daco_gnn_forward: 14 kernel launches, MFMA edge linears, BatchNorm folded);
  * training mode -> daco_gnn_train_forward / daco_gnn_train_backward behind a torch.autograd.Function
    (csrc/daco_gnn_train.hip: BatchNorm on the statistics of the single graph, as in the reference
    tsp/net.py:21,24,43-44; MFMA linears and weight gradients; no library GEMM).  The BatchNorm running
    statistics are updated from the statistics the kernels report (every BatchNorm1d configuration: momentum, cumulative
    average, no running statistics);
  * eval mode with a gradient required -> the same kernels, the running statistics as constants (`fixed_stats`).
There is no torch-op path in forward(): the module tree (EmbNet / ParNet.forward) stays callable for cross-checks only.
`pyg` only needs `.x`, `.edge_index`, `.edge_attr` (torch_geometric is not required).
"""
import torch
from torch import nn
from torch.nn import functional as F

from . import _lib
from . import engine

DEPTH = 12
UNITS = 32


class GraphData:
    """Minimal stand-in for torch_geometric.data.Data: an attribute bag (x, edge_index, edge_attr)."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self


class GraphBatchNorm(nn.Module):
    """Same parameter path as torch_geometric.nn.BatchNorm: the BatchNorm1d lives at `.module`."""

    def __init__(self, channels):
        super().__init__()
        self.module = nn.BatchNorm1d(channels)

    def forward(self, x):
        return self.module(x)


def mean_by_source(values, src, n):
    """global_mean_pool(values, src): mean of the rows of `values` grouped by `src` (0 for empty groups)."""
    total = torch.zeros((n, values.shape[1]), dtype=values.dtype, device=values.device).index_add_(0, src, values)
    count = torch.bincount(src, minlength=n).clamp_(min=1).to(values.dtype)
    return total / count.unsqueeze(1)


class EmbNet(nn.Module):
    """12 residual layers updating node states x and edge states w (tsp/net.py:8-45)."""

    def __init__(self, depth=DEPTH, feats=2, units=UNITS, act_fn='silu', agg_fn='mean', node_update=True):
        super().__init__()
        assert act_fn == 'silu' and agg_fn == 'mean' and units == UNITS and depth == DEPTH
        self.depth, self.feats, self.units = depth, feats, units
        # sop/net.py:43 and smtwtp/net.py:42 have the node update commented out: x stays the input embedding, v_lins1/2 and
        # v_bns exist (checkpoints hold them) but nothing reads them
        self.node_update = node_update
        self.v_lin0 = nn.Linear(feats, units)
        self.v_lins1 = nn.ModuleList([nn.Linear(units, units) for _ in range(depth)])
        self.v_lins2 = nn.ModuleList([nn.Linear(units, units) for _ in range(depth)])
        self.v_lins3 = nn.ModuleList([nn.Linear(units, units) for _ in range(depth)])
        self.v_lins4 = nn.ModuleList([nn.Linear(units, units) for _ in range(depth)])
        self.v_bns = nn.ModuleList([GraphBatchNorm(units) for _ in range(depth)])
        self.e_lin0 = nn.Linear(1, units)
        self.e_lins0 = nn.ModuleList([nn.Linear(units, units) for _ in range(depth)])
        self.e_bns = nn.ModuleList([GraphBatchNorm(units) for _ in range(depth)])

    def forward(self, x, edge_index, edge_attr):
        src, dst = edge_index[0], edge_index[1]
        n = x.shape[0]
        x = F.silu(self.v_lin0(x))
        w = F.silu(self.e_lin0(edge_attr))
        for i in range(self.depth):
            gate = torch.sigmoid(w)
            msg = mean_by_source(gate * self.v_lins2[i](x)[dst], src, n) if self.node_update else None
            w_new = w + F.silu(self.e_bns[i](self.e_lins0[i](w) + self.v_lins3[i](x)[src] + self.v_lins4[i](x)[dst]))
            if self.node_update:
                x = x + F.silu(self.v_bns[i](self.v_lins1[i](x) + msg))
            w = w_new
        return w


class MLP(nn.Module):
    @property
    def device(self):
        return self._dummy.device

    def __init__(self, units_list, act_fn):
        super().__init__()
        assert act_fn == 'silu'
        self._dummy = nn.Parameter(torch.empty(0), requires_grad=False)
        self.units_list = units_list
        self.depth = len(units_list) - 1
        self.lins = nn.ModuleList([nn.Linear(units_list[i], units_list[i + 1]) for i in range(self.depth)])

    def forward(self, x):
        for i, lin in enumerate(self.lins):
            x = lin(x)
            x = F.silu(x) if i < self.depth - 1 else torch.sigmoid(x)
        return x


class ParNet(MLP):
    def __init__(self, depth=3, units=UNITS, preds=1, act_fn='silu'):
        self.units, self.preds = units, preds
        super().__init__([units] * depth + [preds], act_fn)

    def forward(self, x):
        return super().forward(x).squeeze(dim=-1)


def _csr_graph(pyg, n, device):
    """(src, dst, rowptr, perm) int32 for the kernels; cached on the graph object."""
    graph = getattr(pyg, "_daco_graph", None)
    if graph is None:
        ei = pyg.edge_index
        src64 = ei[0]
        sorted_already = bool((src64[1:] >= src64[:-1]).all()) if src64.numel() > 1 else True
        perm = None if sorted_already else torch.argsort(src64, stable=True).to(torch.int32).contiguous()
        rowptr = torch.zeros(n + 1, dtype=torch.int32, device=device)
        rowptr[1:] = torch.cumsum(torch.bincount(src64, minlength=n), 0).to(torch.int32)
        graph = (ei[0].to(torch.int32).contiguous(), ei[1].to(torch.int32).contiguous(), rowptr, perm)
        try:
            pyg._daco_graph = graph
        except Exception:
            pass
    return graph


_ROWPTR_CACHE = {}


def _regular_rowptr(nodes, k, device):
    """arange(nodes + 1) * k as int32: the CSR row pointer of a graph whose every node has k out-edges (read-only, cached)."""
    key = (int(nodes), int(k), str(device))
    rp = _ROWPTR_CACHE.get(key)
    if rp is None:
        if len(_ROWPTR_CACHE) > 64:
            _ROWPTR_CACHE.clear()
        rp = _ROWPTR_CACHE[key] = torch.arange(0, nodes + 1, device=device, dtype=torch.int32) * int(k)
    return rp


def _merge_graphs(x, edge_index, edge_attr, k_sparse=None):
    """B equal-sized graphs as one block-diagonal GraphData.  k_sparse: the caller's promise that every graph is the regular
    k-nearest-neighbour layout engine.tsp_knn_graph / gen_pyg_data build (edge j*k + t leaves node j): the CSR arrays are then
    written down directly instead of being derived (sortedness check, bincount, cumsum and a host sync per call) -- and if the
    edge_index is the very tensor engine.tsp_knn_graph returned, untouched since, the merged int32 arrays that launch wrote are
    taken as they are (no elementwise launch at all; the merged graph then has no int64 `edge_index` of its own)."""
    B, n, feats = x.shape
    E = edge_index.shape[2]
    if k_sparse is not None and E != n * int(k_sparse):
        raise _lib.DacoError(f"k_sparse = {k_sparse} does not describe graphs of {n} nodes and {E} edges")
    csr = getattr(edge_index, "_daco_csr", None)
    if (k_sparse is not None and csr is not None and csr[2] == n and csr[3] == int(k_sparse) and csr[4] == edge_index._version
            and csr[0].numel() == B * E and csr[0].device == x.device):
        merged = GraphData(x=x.reshape(B * n, feats), edge_index=None, edge_attr=edge_attr.reshape(B * E, 1))
        merged._daco_graph = (csr[0], csr[1], _regular_rowptr(B * n, k_sparse, x.device), None)
        return merged
    off = (torch.arange(B, device=x.device, dtype=edge_index.dtype) * n).view(B, 1, 1)
    ei = (edge_index + off).permute(1, 0, 2).reshape(2, B * E)
    merged = GraphData(x=x.reshape(B * n, feats), edge_index=ei, edge_attr=edge_attr.reshape(B * E, 1))
    if k_sparse is not None:
        merged._daco_graph = (ei[0].to(torch.int32).contiguous(), ei[1].to(torch.int32).contiguous(),
                              _regular_rowptr(B * n, k_sparse, x.device), None)
    return merged


class _GnnTrainFn(torch.autograd.Function):
    """heu = Net(graph) in training mode; the backward returns d loss / d (flat parameter block)."""

    @staticmethod
    def forward(ctx, flat, x, attr, src, dst, rowptr, perm, feats, G, fixed=None):
        # (perm may be None; the destination CSR is derived here, once per forward, for the backward)
        n, E = x.shape[0], src.numel()
        L = _lib.lib()
        dev = x.device
        with torch.cuda.device(dev):
            heu = torch.empty(E, dtype=torch.float32, device=dev)
            stats = torch.empty((DEPTH, 2, G, UNITS, 2), dtype=torch.float32, device=dev)
            # the activations the backward reads stay in this block: one per forward call, kept by ctx
            ws = torch.empty(L.daco_gnn_train_workspace_bytes(n, E, G), dtype=torch.uint8, device=dev)
            flat = flat.detach().contiguous()
            rc = L.daco_gnn_train_forward(engine._stream(dev), n, E, feats, G, x.data_ptr(), src.data_ptr(), dst.data_ptr(),
                                          rowptr.data_ptr(), perm.data_ptr() if perm is not None else None, attr.data_ptr(),
                                          flat.data_ptr(), heu.data_ptr(), stats.data_ptr(),
                                          fixed.data_ptr() if fixed is not None else None, ws.data_ptr(), ws.numel())
        _lib.check(rc, "daco_gnn_train_forward")
        ctx.save_for_backward(flat, x, attr, src, dst, rowptr, heu)
        ctx.ws, ctx.feats, ctx.G, ctx.perm, ctx.fixed = ws, feats, G, perm, fixed is not None
        ctx.mark_non_differentiable(stats)
        return heu, stats

    @staticmethod
    def backward(ctx, gheu, _gstats):
        if ctx.ws is None:
            raise RuntimeError("Net (HIP training path): the saved activations of this forward were already consumed by a "
                               "backward pass -- backward through the same forward a second time is not supported "
                               "(run the forward again)")
        flat, x, attr, src, dst, rowptr, heu = ctx.saved_tensors
        n, E = x.shape[0], src.numel()
        L = _lib.lib()
        dev = x.device
        with torch.cuda.device(dev):
            gflat = torch.empty_like(flat)
            gheu = gheu.float().contiguous()
            perm = ctx.perm                     # (destination CSR: None -> the library builds it in the workspace)
            rc = L.daco_gnn_train_backward(engine._stream(dev), n, E, ctx.feats, ctx.G, x.data_ptr(), src.data_ptr(),
                                           dst.data_ptr(), rowptr.data_ptr(), perm.data_ptr() if perm is not None else None,
                                           None, None, attr.data_ptr(), flat.data_ptr(),
                                           heu.data_ptr(), gheu.data_ptr(), gflat.data_ptr(), int(ctx.fixed), ctx.ws.data_ptr(),
                                           ctx.ws.numel())
        _lib.check(rc, "daco_gnn_train_backward")
        ctx.ws = None
        return (gflat,) + (None,) * 9


class Net(nn.Module):
    """feats: node-feature width (2 = coordinates in tsp/, 1 in tsp_nls/ and cvrp/);
    with_phe: also create the unused par_net_phe head that tsp/ checkpoints contain;
    node_update=False: the sop/ and smtwtp/ variant whose node states are never updated (sop/net.py:43).  The kernels
    run it as the same layer with the node BatchNorm's scale and shift set to zero: x + silu(0) = x exactly."""

    def __init__(self, feats=2, with_phe=True, node_update=True):
        super().__init__()
        self.emb_net = EmbNet(feats=feats, node_update=node_update)
        if with_phe:
            self.par_net_phe = ParNet()
        self.par_net_heu = ParNet()
        self._packed = None
        self._packed_key = None

    # ------------------------------------------------------------------ reference surface
    def forward(self, pyg):
        """tsp/net.py:84-88.  Every mode runs on the HIP kernels:
          training mode                        -> daco_gnn_train_forward / _backward, per-graph batch statistics (+ the running
                                                  statistics' update, for every BatchNorm1d configuration)
          eval mode under autograd             -> the same kernels with the running statistics as constants (fixed_stats)
          eval mode without a gradient         -> daco_gnn_forward (BatchNorm folded into scale / shift)
        (The module tree itself -- EmbNet / ParNet.forward as torch ops -- stays callable for cross-checks; forward() never is.)"""
        x = pyg.x
        if not x.is_cuda:
            raise _lib.DacoError("deepaco_amd.Net runs on a HIP device only (got CPU tensors)")
        needs_graph = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        # BatchNorm1d(track_running_stats=False) normalises with the batch statistics in eval mode too: there is nothing to fold
        # into scale / shift, the statistics kernels run (without a graph under no_grad) -- ADVICE r5
        if self.training or needs_graph or not self._bn_config()[0]:
            return self.forward_train_hip(pyg)
        return self.forward_hip(pyg)

    def _bn_config(self):
        """(tracks running statistics, momentum | None) -- one configuration for all BatchNorm modules of the network."""
        bns = [bn.module for bn in list(self.emb_net.v_bns) + list(self.emb_net.e_bns)]
        cfg = {(bool(bn.track_running_stats and bn.running_mean is not None), bn.momentum) for bn in bns}
        if len(cfg) != 1:
            raise _lib.DacoError("deepaco_amd.Net: the BatchNorm modules of one network must share track_running_stats / momentum")
        return next(iter(cfg))

    def freeze_gnn(self):
        for param in self.emb_net.parameters():
            param.requires_grad = False

    @staticmethod
    def reshape(pyg, vector):
        '''Turn phe/heu vector into matrix with zero padding (tsp/net.py:94-102)'''
        n_nodes = pyg.x.shape[0]
        matrix = torch.zeros(size=(n_nodes, n_nodes), device=pyg.x.device, dtype=vector.dtype)
        matrix[pyg.edge_index[0], pyg.edge_index[1]] = vector
        return matrix

    # ------------------------------------------------------------------ HIP training path
    def pack_params_train(self):
        """Flat parameter block for the training kernels, built from the live parameters with differentiable ops
        (views + one cat), so that autograd hands the flat gradient back to every nn.Parameter.  Layout of
        csrc/daco_gnn.hip with gamma / beta in the BatchNorm slots."""
        e = self.emb_net
        parts = [e.v_lin0.weight.reshape(-1), e.v_lin0.bias, e.e_lin0.weight.reshape(-1), e.e_lin0.bias]
        for i in range(DEPTH):
            # without the node update (sop / smtwtp: the reference never calls v_lins1, v_lins2, v_bns, so their .grad stays
            # None and AdamW's decoupled weight decay leaves them alone) those slices enter the block detached
            live = (lambda t: t) if e.node_update else (lambda t: t.detach())
            Wv = torch.cat([live(e.v_lins1[i].weight), live(e.v_lins2[i].weight), e.v_lins3[i].weight, e.v_lins4[i].weight], 0)   # [128, 32]
            bv = torch.cat([live(e.v_lins1[i].bias), live(e.v_lins2[i].bias), e.v_lins3[i].bias, e.v_lins4[i].bias], 0)
            vg, vb = e.v_bns[i].module.weight, e.v_bns[i].module.bias
            if not e.node_update:                              # gamma = beta = 0: the node state passes through unchanged
                vg, vb = torch.zeros_like(vg), torch.zeros_like(vb)
            parts += [Wv.t().reshape(-1), bv, e.e_lins0[i].weight.reshape(-1), e.e_lins0[i].bias,
                      vg, vb, e.e_bns[i].module.weight, e.e_bns[i].module.bias]
        h = self.par_net_heu.lins
        parts += [h[0].weight.reshape(-1), h[0].bias, h[1].weight.reshape(-1), h[1].bias, h[2].weight.reshape(-1), h[2].bias]
        return torch.cat([p.float().reshape(-1) for p in parts])

    # ------------------------------------------------------------------ one flat parameter block (round 6)
    def _flat_specs(self):
        """(parameter, offset, shape, stride) of every parameter the kernels read inside the training block (the layout
        pack_params_train writes; csrc/daco_gnn.hip).  The four node linears of a layer are one [32 in][128 out] matrix there:
        linear q's weight [out][in] is the strided view (1, 128) at column 32 q."""
        e = self.emb_net
        specs, off = [], 0

        def put(p, shape, stride):
            nonlocal off
            specs.append((p, off, shape, stride))
            off += p.numel()

        put(e.v_lin0.weight, (UNITS, e.feats), (e.feats, 1))
        put(e.v_lin0.bias, (UNITS,), (1,))
        put(e.e_lin0.weight, (UNITS, 1), (1, 1))
        put(e.e_lin0.bias, (UNITS,), (1,))
        for i in range(DEPTH):
            lins = (e.v_lins1[i], e.v_lins2[i], e.v_lins3[i], e.v_lins4[i])
            for q, lin in enumerate(lins):
                specs.append((lin.weight, off + UNITS * q, (UNITS, UNITS), (1, 4 * UNITS)))
            off += 4 * UNITS * UNITS
            for q, lin in enumerate(lins):
                specs.append((lin.bias, off + UNITS * q, (UNITS,), (1,)))
            off += 4 * UNITS
            put(e.e_lins0[i].weight, (UNITS, UNITS), (UNITS, 1))
            put(e.e_lins0[i].bias, (UNITS,), (1,))
            for bn in (e.v_bns[i].module, e.e_bns[i].module):
                put(bn.weight, (UNITS,), (1,))
                put(bn.bias, (UNITS,), (1,))
        for lin in self.par_net_heu.lins:
            put(lin.weight, tuple(lin.weight.shape), (lin.weight.shape[1], 1))
            put(lin.bias, tuple(lin.bias.shape), (1,))
        return specs, off

    def flatten_parameters(self):
        """Re-seat every parameter the kernels read as a VIEW of one flat f32 block in the training kernels' layout and return
        that block as a single leaf nn.Parameter (the training step re-packed ~100 tensors with torch.cat every
        step and autograd split the flat gradient back into ~100 pieces -- two hundred launches of a few microseconds each).
        After this call
          * the training forward hands the block to daco_gnn_train_forward as it is (no pack), the backward's flat gradient IS
            the block's .grad, and every parameter's .grad is a view of it (set when the gradient arrives);
          * an elementwise optimizer may be built on [block] alone -- `torch.optim.AdamW(net.train_parameters(), ...)` -- and is
            then one fused update of one tensor: AdamW's update and decoupled weight decay are elementwise, so this is the
            update `AdamW(net.parameters())` makes, element for element; clip_grad_norm_ over [block] is the same norm;
          * state_dict / load_state_dict / eval-mode inference see the same module tree (load copies into the views).
        Moving or casting the module afterwards (.to / .double / .cuda) gives the parameters storages of their own again: the
        next training forward notices and raises; call flatten_parameters() again (and rebuild the optimizer).
        Only for networks whose every covered parameter is trained (node_update=True, nothing frozen): AdamW leaves a parameter
        without a gradient alone, a flat block cannot."""
        e = self.emb_net
        specs, total = self._flat_specs()
        if not e.node_update or any(not p.requires_grad for p, _, _, _ in specs):
            raise _lib.DacoError("Net.flatten_parameters: every parameter of the block must be trained (node_update=True, none "
                                 "frozen): the optimizer's weight decay would move the ones the reference never touches")
        if any(p.dtype != torch.float32 for p, _, _, _ in specs):
            raise _lib.DacoError("Net.flatten_parameters: float32 parameters only")
        with torch.no_grad():
            flat = self.pack_params_train().detach().clone().contiguous()
        assert flat.numel() == total, (flat.numel(), total)
        for p, off, shape, stride in specs:
            view = flat.as_strided(shape, stride, off)
            assert torch.equal(view, p.data), "flat layout does not match the parameter"
            p.data = view
        block = nn.Parameter(flat)
        self._flat_holder = [block, specs]          # (a list: not registered -- parameters() keeps listing the module tree only)

        def spread(param):                          # the pieces of the flat gradient, where torch utilities look for them
            g = param.grad
            if g is not None:
                for p, off, shape, stride in specs:
                    p.grad = g.as_strided(shape, stride, off)
        block.register_post_accumulate_grad_hook(spread)
        self._packed = None
        return block

    def __deepcopy__(self, memo):
        # (a copy's parameters get storages of their own: a flattened network is flattened again, on its own block)
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k != "_flat_holder":
                new.__dict__[k] = copy.deepcopy(v, memo)
        if getattr(self, "_flat_holder", None) is not None:
            new.flatten_parameters()
        return new

    def _flat_block(self):
        """The flat block if flatten_parameters() was called and the parameters still are its views, else None."""
        holder = getattr(self, "_flat_holder", None)
        if holder is None:
            return None
        block, specs = holder
        esz = block.element_size()
        for p, off, _, _ in (specs[0], specs[len(specs) // 2], specs[-1]):
            if p.data_ptr() != block.data_ptr() + off * esz or p.device != block.device:
                raise _lib.DacoError("Net: the parameters no longer alias the block flatten_parameters() made (the module was "
                                     "moved or cast afterwards): call flatten_parameters() again and rebuild the optimizer")
        return block

    def train_parameters(self):
        """What to hand the optimizer: [the flat block] after flatten_parameters(), else the parameters that receive a gradient."""
        block = self._flat_block()
        return [block] if block is not None else [p for p in self.parameters() if p.requires_grad]

    @torch.no_grad()
    def _update_running_stats(self, stats, count_e, count_v):
        """BatchNorm1d's training-mode side effect, from the statistics the kernels report, as G successive reference forwards
        (one per graph) would leave it: momentum m: running = (1 - m) * running + m * batch per graph (variance unbiased);
        momentum None (cumulative average): running = (k * running + sum of the batches) / (k + G), k = num_batches_tracked."""
        tracks, momentum = self._bn_config()
        if not tracks:
            return
        G = stats.shape[2]
        # all 24 BatchNorms at once (a dozen launches instead of eight per module: the step is made of small kernels)
        e = self.emb_net
        which = [0, 1] if e.node_update else [0]                       # (the reference never calls the node BatchNorms of sop / smtwtp)
        bns = [(i, w, (e.e_bns[i] if w == 0 else e.v_bns[i]).module) for i in range(DEPTH) for w in which]
        mean = stats[..., 0]                                           # [12, 2, G, 32]
        var = torch.stack((stats[:, 0, :, :, 1] * (count_e / max(count_e - 1, 1)),
                           stats[:, 1, :, :, 1] * (count_v / max(count_v - 1, 1))), dim=1)      # unbiased, as BatchNorm1d tracks it
        rm, rv = [bn.running_mean for _, _, bn in bns], [bn.running_var for _, _, bn in bns]
        nbt = [bn.num_batches_tracked for _, _, bn in bns]
        if momentum is None:
            k = float(nbt[0])                                          # (one host read; the cumulative average's weight)
            add_m, add_v = mean.sum(2) / (k + G), var.sum(2) / (k + G)
            keep = k / (k + G)
        else:
            m = momentum
            decay = ((1 - m) ** torch.arange(G - 1, -1, -1, device=stats.device, dtype=torch.float32)).view(1, 1, G, 1)   # oldest graph first
            add_m, add_v = m * (decay * mean).sum(2), m * (decay * var).sum(2)
            keep = (1 - m) ** G
        torch._foreach_mul_(rm + rv, keep)
        torch._foreach_add_(rm, [add_m[i, w] for i, w, _ in bns])
        torch._foreach_add_(rv, [add_v[i, w] for i, w, _ in bns])
        torch._foreach_add_(nbt, G)

    def _running_stats_block(self):
        """[12][2][32][2] (mean, variance) of the edge (0) and node (1) BatchNorm of every layer: fixed_stats of the kernels."""
        e = self.emb_net
        rows = []
        for i in range(DEPTH):
            for bn in (e.e_bns[i].module, e.v_bns[i].module):
                rows.append(torch.stack((bn.running_mean.float(), bn.running_var.float()), dim=1))
        return torch.stack(rows).view(DEPTH, 2, UNITS, 2).contiguous()

    def forward_train_hip(self, pyg, graphs=1):
        """Differentiable forward through the HIP kernels (graphs > 1: that many equal-sized graphs side by side).  Training
        mode: every graph is normalised with its own batch statistics; eval mode (a gradient through a module in eval()):
        with the running statistics, as constants."""
        x = pyg.x.float().contiguous()
        n, feats = x.shape
        src, dst, rowptr, perm = _csr_graph(pyg, n, x.device)
        attr = pyg.edge_attr.float().contiguous().view(-1)
        flat = self._flat_block()
        if flat is None or not torch.is_grad_enabled():
            flat = self.pack_params_train() if flat is None else flat.detach()
        tracks, _ = self._bn_config()
        # BatchNorm1d normalises with the batch statistics in training mode -- and in eval mode too when it tracks none
        fixed = self._running_stats_block() if (not self.training and tracks) else None
        heu, stats = _GnnTrainFn.apply(flat, x, attr, src, dst, rowptr, perm, feats, graphs, fixed)
        if self.training:
            self._update_running_stats(stats, src.numel() // graphs, n // graphs)
        return heu

    def forward_batch_train(self, x, edge_index, edge_attr, k_sparse=None):
        """Training forward for B equal-sized graphs in one pass (tsp_nls/train.py's batch of instances): x [B,n,feats],
        edge_index [B,2,E] (ids local to each graph), edge_attr [B,E(,1)] -> heu [B,E]; graph b is normalised with its own
        BatchNorm statistics, exactly as B separate training forwards.  k_sparse: see _merge_graphs."""
        B, E = x.shape[0], edge_index.shape[2]
        return self.forward_train_hip(_merge_graphs(x, edge_index, edge_attr, k_sparse), graphs=B).view(B, E)

    # ------------------------------------------------------------------ HIP inference path
    def pack_params(self):
        """Flat f32 parameter block in the layout csrc/daco_gnn.hip documents (BatchNorm folded)."""
        key = tuple(t._version for t in list(self.parameters()) + list(self.buffers())) + (next(self.parameters()).device,)
        if self._packed is not None and self._packed_key == key:
            return self._packed
        e = self.emb_net
        with torch.no_grad():
            parts = [e.v_lin0.weight.reshape(-1), e.v_lin0.bias, e.e_lin0.weight.reshape(-1), e.e_lin0.bias]
            if not self._bn_config()[0]:
                raise _lib.DacoError("Net.pack_params folds the BatchNorm running statistics; this network tracks none "
                                     "(track_running_stats=False): forward() / forward_batch() run the statistics kernels instead")
            for i in range(DEPTH):
                Wv = torch.cat([m[i].weight for m in (e.v_lins1, e.v_lins2, e.v_lins3, e.v_lins4)], 0)   # [128, 32]
                bv = torch.cat([m[i].bias for m in (e.v_lins1, e.v_lins2, e.v_lins3, e.v_lins4)], 0)
                parts += [Wv.t().contiguous().reshape(-1), bv, e.e_lins0[i].weight.reshape(-1), e.e_lins0[i].bias]
                for bn in (e.v_bns[i].module, e.e_bns[i].module):
                    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                    shift = bn.bias - bn.running_mean * scale
                    if bn is e.v_bns[i].module and not e.node_update:
                        scale, shift = torch.zeros_like(scale), torch.zeros_like(shift)
                    parts += [scale, shift]
            h = self.par_net_heu.lins
            parts += [h[0].weight.reshape(-1), h[0].bias, h[1].weight.reshape(-1), h[1].bias, h[2].weight.reshape(-1),
                      h[2].bias]
            flat = torch.cat([p.float().reshape(-1) for p in parts]).contiguous()
        assert flat.numel() == _lib.lib().daco_gnn_param_floats(e.feats)
        self._packed, self._packed_key = flat, key
        return flat

    @torch.no_grad()
    def forward_batch(self, x, edge_index, edge_attr, k_sparse=None):
        """Eval-mode forward for B graphs of equal size in ONE pass of the HIP kernels: the graphs are laid side
        by side as one block-diagonal graph (eval-mode BatchNorm is a per-feature affine map, so instances do not
        interact).  x [B,n,feats], edge_index [B,2,E] (node ids local to each graph, e.g. engine.tsp_knn_graph),
        edge_attr [B,E,1] or [B,E] -> heu [B,E], row b equal to forward() on graph b alone.  k_sparse: see _merge_graphs."""
        if self.training:
            raise _lib.DacoError("Net.forward_batch is an inference path (BatchNorm running statistics): call .eval()")
        B, E = x.shape[0], edge_index.shape[2]
        if not self._bn_config()[0]:             # no running statistics: every graph is normalised with its own batch statistics
            return self.forward_train_hip(_merge_graphs(x, edge_index, edge_attr, k_sparse), graphs=B).view(B, E)
        return self.forward_hip(_merge_graphs(x, edge_index, edge_attr, k_sparse)).view(B, E)

    @staticmethod
    def reshape_batch(n_nodes, edge_index, heu, eps=0.0):
        """Batched Net.reshape (tsp/net.py:94-102): heu [B,E] -> [B,n,n] with zeros off the graph, `+ eps` included (what every
        caller adds: tsp/train.ipynb:35, tsp_nls/test.py:28).  Outside autograd on the device: one fill and one scatter launch
        (daco_heu_matrix), the values of zeros / indexed assignment / add bit for bit; under autograd the torch ops themselves."""
        B, E = heu.shape
        if heu.is_cuda and heu.dtype == torch.float32 and not (heu.requires_grad and torch.is_grad_enabled()):
            # the scatter kernel skips an edge whose node id lies outside [0, n) where the indexed assignment below (and the
            # reference's) raises: the ids are validated once per edge_index tensor -- a graph built by
            # engine.tsp_knn_graph is in range by construction, anything else costs one aminmax and one host read, cached on
            # the tensor like `_daco_csr`
            if getattr(edge_index, "_daco_ids_ok", None) != (n_nodes, edge_index._version):
                csr = getattr(edge_index, "_daco_csr", None)
                if not (csr is not None and csr[2] == n_nodes and csr[4] == edge_index._version):
                    lo, hi = (int(v) for v in torch.stack(torch.aminmax(edge_index)).tolist())
                    if lo < 0 or hi >= n_nodes:
                        raise IndexError(f"Net.reshape_batch: edge_index holds node ids in [{lo}, {hi}], outside [0, {n_nodes})")
                edge_index._daco_ids_ok = (n_nodes, edge_index._version)
            return engine.heu_matrix(n_nodes, edge_index, heu, fill=eps, add=eps)
        out = torch.zeros((B, n_nodes, n_nodes), dtype=heu.dtype, device=heu.device)
        bidx = torch.arange(B, device=heu.device).view(B, 1).expand(B, E)
        out[bidx, edge_index[:, 0], edge_index[:, 1]] = heu
        return out + eps if eps else out

    @torch.no_grad()
    def forward_hip(self, pyg, return_embedding=False):
        x = pyg.x.float().contiguous()
        n, feats = x.shape
        graph = _csr_graph(pyg, n, x.device)
        src, dst, rowptr, perm = graph
        E = src.numel()
        attr = pyg.edge_attr.float().contiguous().view(-1)
        params = self.pack_params()
        L = _lib.lib()
        dev = x.device
        with torch.cuda.device(dev):
            heu = torch.empty(E, dtype=torch.float32, device=dev)
            emb = torch.empty((E, UNITS), dtype=torch.float32, device=dev) if return_embedding else None
            nbytes = L.daco_gnn_workspace_bytes(n, E)
            ws = engine._workspace(dev, nbytes, "gnn")
            rc = L.daco_gnn_forward(engine._stream(dev), n, E, feats, x.data_ptr(), src.data_ptr(), dst.data_ptr(),
                                    rowptr.data_ptr(), perm.data_ptr() if perm is not None else None, attr.data_ptr(),
                                    params.data_ptr(), heu.data_ptr(), emb.data_ptr() if emb is not None else None,
                                    ws.data_ptr(), ws.numel())
        _lib.check(rc, "daco_gnn_forward")
        return (heu, emb) if return_embedding else heu
