"""CPU-side checks of the Net drop-in: module tree / state_dict keys equal the reference checkpoints',
graph construction (H1) equals the reference's utils on captured instances."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden


def names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def make_net(name):
    if "tsp_nls" in name:
        from deepaco_amd.tsp_nls.net import Net
    elif "cvrp" in name:
        from deepaco_amd.cvrp.net import Net
    else:
        from deepaco_amd.tsp.net import Net
    return Net()


def load_weights(net, g):
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w__")}
    missing, unexpected = net.load_state_dict(sd, strict=False)
    return missing, unexpected


@pytest.mark.parametrize("name", names("g5_net"))
def test_checkpoint_keys_load_unchanged(name):
    g = load_golden(name)
    net = make_net(name)
    missing, unexpected = load_weights(net, g)
    assert unexpected == []
    # the fixture drops only integer counters and the empty _dummy parameters
    assert all(k.endswith("num_batches_tracked") or k.endswith("_dummy") for k in missing), missing
    n_keys = len(net.state_dict())
    assert n_keys == (258 if name.startswith("g5_net_tsp_tsp") else 251)
    n_ckpt = sum(v.size for k, v in g.items() if k.startswith("w__") and "running_" not in k)
    assert sum(p.numel() for p in net.parameters()) == n_ckpt


@pytest.mark.parametrize("name", ["g5_net_tsp_tsp100", "g5_net_tsp_tsp20", "g5_net_tsp_nls_tsp100"])
def test_tsp_graph_construction(name):
    from deepaco_amd.tsp.utils import gen_pyg_data
    g = load_golden(name)
    start = 0 if "nls" in name else None
    pyg, dist = gen_pyg_data(torch.from_numpy(g["coords"]), int(g["k_sparse"]), start_node=start)
    assert np.array_equal(pyg.edge_index.numpy(), g["edge_index"])
    assert np.array_equal(pyg.edge_attr.numpy(), g["edge_attr"])
    assert np.array_equal(pyg.x.numpy(), g["x"])
    assert np.array_equal(dist.numpy(), g["distances"])


def test_cvrp_graph_construction():
    from deepaco_amd.cvrp.utils import gen_pyg_data, gen_instance
    g = load_golden("g5_net_cvrp_cvrp20")
    pyg = gen_pyg_data(torch.from_numpy(g["demand"]), torch.from_numpy(g["distances"]), "cpu")
    assert np.array_equal(pyg.edge_index.numpy(), g["edge_index"])
    assert np.array_equal(pyg.edge_attr.numpy(), g["edge_attr"])
    assert np.array_equal(pyg.x.numpy(), g["x"])
    torch.manual_seed(7)
    dem, dist = gen_instance(20, "cpu")
    assert np.array_equal(dem.numpy(), g["demand"]) and np.array_equal(dist.numpy(), g["distances"])


def test_net_refuses_cpu():
    from deepaco_amd import _lib
    g = load_golden("g5_net_tsp_tsp20")
    net = make_net("g5_net_tsp_tsp20").eval()
    from deepaco_amd.net import GraphData
    pyg = GraphData(x=torch.from_numpy(g["x"]), edge_index=torch.from_numpy(g["edge_index"]),
                    edge_attr=torch.from_numpy(g["edge_attr"]))
    with pytest.raises(_lib.DacoError):
        net(pyg)


def test_regular_knn_csr_shortcut_equals_derived_csr():
    """Net.forward_batch(k_sparse=...) writes the CSR arrays of the regular k-NN layout down directly; they must be what
    _csr_graph derives from the same merged edge list (sortedness check, bincount, cumsum)."""
    from deepaco_amd.net import _csr_graph, _merge_graphs
    from deepaco_amd import _lib
    g = torch.Generator().manual_seed(3)
    for B, n, k in ((1, 7, 3), (3, 20, 5), (4, 33, 1)):
        src = torch.arange(n).repeat_interleave(k)
        ei = torch.stack([torch.stack([src, torch.randint(0, n, (n * k,), generator=g)]) for _ in range(B)])   # [B, 2, n*k]
        x = torch.rand(B, n, 2, generator=g)
        ea = torch.rand(B, n * k, 1, generator=g)
        hinted = _merge_graphs(x, ei, ea, k_sparse=k)
        plain = _merge_graphs(x, ei, ea)
        assert not hasattr(plain, "_daco_graph")
        a = hinted._daco_graph
        b = _csr_graph(plain, B * n, x.device)
        assert b[3] is None and a[3] is None                       # already sorted by source: no permutation
        for u, v in zip(a[:3], b[:3]):
            assert u.dtype == torch.int32 and torch.equal(u, v)
        assert torch.equal(hinted.edge_index, plain.edge_index) and torch.equal(hinted.x, plain.x)
    with pytest.raises(_lib.DacoError):
        _merge_graphs(torch.rand(2, 5, 2), torch.zeros(2, 2, 12, dtype=torch.long), torch.rand(2, 12, 1), k_sparse=3)


def test_attached_merged_graph_is_taken_only_while_it_describes_the_tensor():
    """engine.tsp_knn_graph leaves the batch's merged int32 graph on the edge_index it returns (`_daco_csr`); _merge_graphs
    takes it only if it is of the same shape parameters and the tensor has not been written to since (version counter).
    Host logic: checked here with the attribute put on a CPU tensor by hand."""
    from deepaco_amd.net import _merge_graphs, _regular_rowptr
    g = torch.Generator().manual_seed(5)
    B, n, k = 3, 12, 4
    src = torch.arange(n).repeat_interleave(k)
    ei = torch.stack([torch.stack([src, torch.randint(0, n, (n * k,), generator=g)]) for _ in range(B)])
    x, ea = torch.rand(B, n, 2, generator=g), torch.rand(B, n * k, 1, generator=g)
    derived = _merge_graphs(x, ei, ea, k_sparse=k)
    off = (torch.arange(B) * n).view(B, 1)
    src32, dst32 = (ei[:, 0] + off).reshape(-1).int(), (ei[:, 1] + off).reshape(-1).int()
    ei._daco_csr = (src32, dst32, n, k, ei._version)
    fast = _merge_graphs(x, ei, ea, k_sparse=k)
    assert fast.edge_index is None and fast._daco_graph[0] is src32 and fast._daco_graph[1] is dst32
    for a, b in zip(fast._daco_graph[:3], derived._daco_graph[:3]):
        assert torch.equal(a, b)
    assert fast._daco_graph[2] is _regular_rowptr(B * n, k, x.device)           # (the row pointer is shared, read-only)
    assert _merge_graphs(x, ei, ea).edge_index is not None                      # no k_sparse promise: derived
    ei._daco_csr = (src32, dst32, n, k + 1, ei._version)                        # other shape parameters: derived
    assert _merge_graphs(x, ei, ea, k_sparse=k).edge_index is not None
    ei._daco_csr = (src32, dst32, n, k, ei._version)
    ei[0, 1, 0] = (ei[0, 1, 0] + 1) % n                                         # written to: derived (and correct)
    after = _merge_graphs(x, ei, ea, k_sparse=k)
    assert after.edge_index is not None and int(after._daco_graph[1][0]) == int(ei[0, 1, 0])
    assert getattr(ei[:2], "_daco_csr", None) is None                           # (a slice is a new tensor object: nothing rides on it)


def _tree_forward(net, x, ei, ea):
    return net.par_net_heu(net.emb_net(x, ei, ea))


def test_flatten_parameters_views_of_one_block():
    """Net.flatten_parameters (round 6): every parameter the kernels read becomes a view of ONE flat block in the training
    kernels' layout.  Held here on the CPU (the method is torch only): the values and the module tree's output do not move, the
    block equals what pack_params_train packs, an optimizer on the block alone updates the module's parameters -- AdamW is
    elementwise, so exactly as AdamW over the parameter list does --, a checkpoint loads INTO the views, and a cast afterwards is
    noticed."""
    import copy
    from deepaco_amd import _lib
    from deepaco_amd.tsp_nls.net import Net
    torch.manual_seed(5)
    net = Net()
    ref = copy.deepcopy(net)
    n, k = 12, 4
    gen = torch.Generator().manual_seed(2)
    x = torch.rand(n, 1, generator=gen)
    src = torch.repeat_interleave(torch.arange(n), k)
    dst = torch.randint(0, n, (n * k,), generator=gen)
    ei, ea = torch.stack([src, dst]), torch.rand(n * k, 1, generator=gen)
    before = _tree_forward(net.train(), x, ei, ea).detach().clone()
    packed = net.pack_params_train().detach().clone()
    sd_before = {k_: v.clone() for k_, v in net.state_dict().items()}
    block = net.flatten_parameters()
    assert block.is_leaf and block.requires_grad and torch.equal(block.detach(), packed)
    assert net.train_parameters() == [block] or (len(net.train_parameters()) == 1 and net.train_parameters()[0] is block)
    assert all(torch.equal(v, sd_before[k_]) for k_, v in net.state_dict().items())
    assert len(list(net.parameters())) == len(list(ref.parameters()))          # the block is not listed twice
    covered = sum(p.numel() for p, _, _, _ in net._flat_specs()[0])
    assert covered == block.numel()
    lo, hi = block.data_ptr(), block.data_ptr() + block.numel() * 4
    inside = [lo <= p.data_ptr() < hi for p in net.parameters() if p.numel()]
    assert sum(inside) == len(net._flat_specs()[0])                             # (par_net_phe: none in tsp_nls)
    torch.testing.assert_close(_tree_forward(net, x, ei, ea), before, rtol=0, atol=0)
    assert torch.equal(net.pack_params_train().detach(), block.detach())

    # one AdamW step on the block == one AdamW step on the parameter list, element for element
    coef = torch.randn(n * k, generator=gen)
    opt_ref = torch.optim.AdamW(ref.parameters(), lr=1e-2)
    torch.sum(_tree_forward(ref.train(), x, ei, ea) * coef).backward()
    opt_ref.step()
    opt = torch.optim.AdamW(net.train_parameters(), lr=1e-2)
    # (on the CPU the gradient of the block comes from the reference net's parameter gradients, packed the same way)
    gparts = copy.deepcopy(ref)
    for p, q in zip(gparts.parameters(), ref.parameters()):
        p.data = q.grad.clone() if q.grad is not None else torch.zeros_like(q)
    block.grad = gparts.pack_params_train().detach().clone()
    opt.step()
    for (k1, p), (k2, q) in zip(net.named_parameters(), ref.named_parameters()):
        if q.grad is not None:
            torch.testing.assert_close(p.detach(), q.detach(), rtol=1e-6, atol=1e-7, msg=k1)
    assert torch.equal(net.pack_params_train().detach(), block.detach())        # the views moved with the block

    # a checkpoint loads into the views; the block follows
    net.load_state_dict(sd_before)
    assert torch.equal(block.detach(), packed)
    assert net._flat_block() is block
    # frozen parameters / the variant without node updates cannot be flattened; a cast breaks the aliasing and is noticed
    net.double()
    with pytest.raises(_lib.DacoError):
        net._flat_block()
    frozen = Net()
    frozen.freeze_gnn()
    with pytest.raises(_lib.DacoError):
        frozen.flatten_parameters()


def test_trainer_has_no_cpu_path():
    """pipeline.TspNlsTrainer is a device object: on a host without a HIP device it says so (no silent CPU training loop)."""
    from deepaco_amd import _lib
    from deepaco_amd.pipeline import TspNlsTrainer
    from deepaco_amd.tsp_nls.net import Net
    if torch.cuda.is_available():
        pytest.skip("a HIP device is visible: the trainer runs (tests/test_gpu_07_net.py)")
    with pytest.raises(_lib.DacoError):
        TspNlsTrainer(Net(), 2, 20, 4, 5)
