"""GPU tests of the heuristic network: HIP inference path vs the reference's outputs with its shipped
checkpoints (fixtures g5_net_*), the torch-op training path, and an end-to-end training step."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, assert_close_mostly
from oracle import gnn as ognn
from test_net_host import make_net, load_weights, names

pytestmark = pytest.mark.gpu


def forward_as(net, pyg, backend):
    """backend 'hip': Net.forward (always the HIP kernels).  'torch': the module tree evaluated as torch ops with autograd (EmbNet /
    ParNet.forward: the op sequence of tsp/net.py:27-75) -- the cross-check these tests hold the kernels to; the product's
    forward() never takes it."""
    if backend == "hip":
        return net(pyg)
    return net.par_net_heu(net.emb_net(pyg.x, pyg.edge_index, pyg.edge_attr))

ATOL_HEU = 1e-5        # heu[E] eval-mode, abs tol 1e-5 (the HIP path: fixed arithmetic, same on every box)
ATOL_TORCH = 1e-4      # the torch-op path (rocBLAS GEMMs; kernel selection differs from box to box)


def dev():
    return torch.device("cuda:0")


def graph(g):
    from deepaco_amd.net import GraphData
    return GraphData(x=torch.from_numpy(g["x"]), edge_index=torch.from_numpy(g["edge_index"]),
                     edge_attr=torch.from_numpy(g["edge_attr"])).to(dev())


@pytest.mark.parametrize("name", names("g5_net"))
def test_net_eval_hip_matches_reference(name):
    g = load_golden(name)
    net = make_net(name)
    load_weights(net, g)
    net = net.to(dev()).eval()
    pyg = graph(g)
    with torch.no_grad():
        heu = net(pyg)                                   # HIP path (eval, no grad)
    np.testing.assert_allclose(heu.cpu().numpy(), g["heu_eval"], atol=ATOL_HEU, rtol=1e-4)
    heu2, emb = net.forward_hip(pyg, return_embedding=True)
    assert torch.equal(heu, heu2)
    np.testing.assert_allclose(emb.cpu().numpy(), g["emb_eval"], atol=3e-4, rtol=3e-4)
    # torch-op path (what training uses) agrees with the HIP path
    emb_t = net.emb_net(pyg.x, pyg.edge_index, pyg.edge_attr)
    heu_t = net.par_net_heu(emb_t)
    # checked against the reference's output at its own tolerance (rocBLAS picks box-dependent kernels, and after 12
    # residual layers one mid-sigmoid element can move by a few 1e-5), and only loosely against the HIP path
    np.testing.assert_allclose(heu_t.detach().cpu().numpy(), g["heu_eval"], atol=ATOL_TORCH, rtol=5e-4)
    np.testing.assert_allclose(heu_t.detach().cpu().numpy(), heu.cpu().numpy(), atol=ATOL_TORCH, rtol=5e-4)
    if "cvrp" not in name:
        mat = net.reshape(pyg, heu)
        np.testing.assert_allclose(mat.cpu().numpy(), g["heu_mat"], atol=ATOL_HEU, rtol=1e-4)
        assert (mat.cpu().numpy() == 0).sum() == (g["heu_mat"] == 0).sum()


@pytest.mark.parametrize("name", names("g5_net"))
def test_net_train_mode_matches_reference(name):
    g = load_golden(name)
    net = make_net(name)
    load_weights(net, g)
    net = net.to(dev()).train()
    for backend in ("hip", "torch"):                       # the HIP training kernels and the torch-op cross-check path
        with torch.no_grad():
            heu = forward_as(net, graph(g), backend)
        # the HIP kernels at the eval-mode tolerance (measured: largest error 2.9e-5 at |ref| ~ 1, no element beyond
        # 1e-5 + 1e-4 |ref|; tools/net_train_mode_error.py, profiles/r06_net_train_mode_error.txt); the torch-op path at its own
        if backend == "hip":
            np.testing.assert_allclose(heu.cpu().numpy(), g["heu_train"], atol=ATOL_HEU, rtol=1e-4, err_msg=backend)
        else:
            np.testing.assert_allclose(heu.cpu().numpy(), g["heu_train"], atol=ATOL_TORCH, rtol=5e-4, err_msg=backend)


def _zero_in_exact_arithmetic(key):
    """Biases added right in front of a BatchNorm (x1 -> bn_v; e_lins0, x3, x4 -> bn_e): the normalisation removes any
    constant shift, so their gradient is a sum that cancels to zero -- what is left is rounding noise, in the
    reference's autograd as much as in the kernels."""
    return key.endswith(".bias") and any(t in key for t in ("v_lins1.", "v_lins3.", "v_lins4.", "e_lins0."))


def _grad_close(got, ref, what, rel=3e-4, floor=1e-7):
    """max |got - ref| <= rel * max |ref| + floor.  `floor` absorbs gradients that are zero in exact arithmetic and pure
    rounding noise in both implementations (a bias in front of a BatchNorm: the normalisation removes the mean)."""
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max())
    assert err <= rel * scale + floor, (what, err, scale)


@pytest.mark.parametrize("gather", ["0", "1"])
@pytest.mark.parametrize("name", names("g7_netgrad"))
def test_net_training_step_hip_matches_reference(name, gather, monkeypatch):
    """(gather: the backward's node gradients as f32 atomics / as CSR row sums of per-edge contributions.)
    G7: train-mode forward (BatchNorm on the graph's own statistics), loss = sum(heu * coef), backward -- all in the
    HIP kernels (csrc/daco_gnn_train.hip) -- against what the reference network and torch autograd produced on the
    same weights and graph: heu, every parameter gradient, and the BatchNorm running statistics after the step."""
    monkeypatch.setenv("DACO_GNN_TRAIN_GATHER", gather)
    g7 = load_golden(name)
    g = load_golden(name.replace("g7_netgrad", "g5_net"))
    net = make_net(name)
    load_weights(net, g)
    net = net.to(dev()).train()
    heu = net(graph(g))
    np.testing.assert_allclose(heu.detach().cpu().numpy(), g7["heu_train"], atol=ATOL_TORCH, rtol=5e-4)
    loss = torch.sum(heu * torch.from_numpy(g7["coef"]).to(dev()))
    np.testing.assert_allclose(float(loss.detach()), float(g7["loss"]), rtol=1e-4, atol=1e-4)
    loss.backward()
    # the yardstick: the same step in float64 on the CPU (the module's torch ops).  The reference's f32 gradients are
    # themselves up to 7e-4 (relative to the tensor's largest entry) away from it on the tsp_nls network, whose
    # one-hot node feature makes layer 0's BatchNorm ill-conditioned; the kernels must be as close as the reference is.
    net64 = make_net(name)
    load_weights(net64, g)
    net64 = net64.double().train()
    h64 = net64.par_net_heu(net64.emb_net(torch.from_numpy(g["x"]).double(), torch.from_numpy(g["edge_index"]),
                                          torch.from_numpy(g["edge_attr"]).double()))
    torch.sum(h64 * torch.from_numpy(g7["coef"]).double()).backward()
    exact = {k: p.grad.numpy() for k, p in net64.named_parameters() if p.grad is not None}
    checked = 0
    for k, p in net.named_parameters():
        if "g__" + k in g7:
            assert p.grad is not None, k
            ref, got = g7["g__" + k].astype(np.float64), p.grad.cpu().numpy().astype(np.float64)
            if _zero_in_exact_arithmetic(k):       # noise against noise: both must be tiny next to the weight's gradient
                wmax = float(np.abs(g7["g__" + k[:-4] + "weight"]).max())
                assert float(np.abs(got).max()) <= 1e-3 * wmax and float(np.abs(ref).max()) <= 1e-3 * wmax, k
            else:
                scale = float(np.abs(exact[k]).max())
                err_ref = float(np.abs(ref - exact[k]).max()) / scale
                err_got = float(np.abs(got - exact[k]).max()) / scale
                assert err_got <= max(2.0 * err_ref, 2e-4), (k, err_got, err_ref)
            checked += 1
    assert checked == sum(1 for k in g7 if k.startswith("g__"))
    for k, v in net.state_dict().items():
        if "rs__" + k in g7:
            np.testing.assert_allclose(v.cpu().numpy(), g7["rs__" + k], rtol=2e-4, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("gather", ["0", "1"])
def test_net_training_hip_equals_torch_autograd_on_random_graph(gather, monkeypatch):
    """Random weights, unsorted edge list with uneven degrees and isolated sources: HIP training forward/backward
    against the same math as torch ops + autograd on the GPU (both forms of the backward's scatter)."""
    monkeypatch.setenv("DACO_GNN_TRAIN_GATHER", gather)
    from deepaco_amd.cvrp.net import Net
    from deepaco_amd.net import GraphData
    torch.manual_seed(0)
    net = Net().to(dev()).train()
    n, E = 41, 500
    gen = torch.Generator().manual_seed(1)
    src = torch.randint(0, n - 3, (E,), generator=gen)
    dst = torch.randint(0, n, (E,), generator=gen)
    pyg = GraphData(x=torch.rand(n, 1, generator=gen), edge_index=torch.stack([src, dst]),
                    edge_attr=torch.rand(E, 1, generator=gen)).to(dev())
    coef = torch.randn(E, generator=gen).to(dev())
    grads = {}
    for backend in ("hip", "torch"):
        net.zero_grad()
        heu = forward_as(net, pyg, backend)
        torch.sum(heu * coef).backward()
        grads[backend] = (heu.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    torch.testing.assert_close(grads["hip"][0], grads["torch"][0], atol=ATOL_TORCH, rtol=5e-4)
    gmax = max(float(v.abs().max()) for v in grads["torch"][1].values())
    for k in grads["hip"][1]:
        if k not in grads["torch"][1]:      # the last layer's node update feeds nothing: autograd leaves None, the kernels write 0
            assert float(grads["hip"][1][k].abs().max()) <= 2e-6 * gmax, k
            continue
        if _zero_in_exact_arithmetic(k):
            wmax = float(grads["torch"][1][k[:-4] + "weight"].abs().max())
            assert float(grads["hip"][1][k].abs().max()) <= 1e-3 * wmax, k
            continue
        _grad_close(grads["hip"][1][k].cpu().numpy(), grads["torch"][1][k].cpu().numpy(), k, rel=1e-3, floor=2e-6 * gmax)


def test_batched_training_forward_uses_per_graph_statistics():
    """B graphs side by side in one training pass == B separate training forwards (each graph normalised with its own
    BatchNorm statistics); gradients add up; running statistics advance as B successive forwards would move them."""
    import copy
    from deepaco_amd import engine
    from deepaco_amd.net import GraphData
    from deepaco_amd.tsp.net import Net
    torch.manual_seed(3)
    B, n, k = 3, 40, 8
    net = Net().to(dev()).train()
    ref = copy.deepcopy(net)
    coords = torch.rand(B, n, 2, device=dev())
    _, ei, ea = engine.tsp_knn_graph(coords, k, want_dist=False)
    coef = torch.randn(B, n * k, device=dev())
    heu = net.forward_batch_train(coords, ei, ea)
    torch.sum(heu * coef).backward()
    for b in range(B):
        one = ref(GraphData(x=coords[b], edge_index=ei[b], edge_attr=ea[b]))
        torch.testing.assert_close(heu[b].detach(), one.detach().view(-1), atol=2e-6, rtol=2e-5)
        torch.sum(one.view(-1) * coef[b]).backward()
    gmax = max(float(q.grad.abs().max()) for q in ref.parameters() if q.grad is not None)
    for (k1, p), (k2, q) in zip(net.named_parameters(), ref.named_parameters()):
        if p.grad is not None and not _zero_in_exact_arithmetic(k1):
            _grad_close(p.grad.cpu().numpy(), q.grad.cpu().numpy(), k1, rel=2e-4, floor=2e-6 * gmax)
    for (k1, v), (k2, w) in zip(net.state_dict().items(), ref.state_dict().items()):
        if "running_" in k1:
            torch.testing.assert_close(v, w, rtol=1e-5, atol=1e-7)


def test_random_graph_vs_oracle():
    """Random weights, unsorted edge list with uneven degrees (exercises perm / CSR path and tile tails)."""
    from deepaco_amd.cvrp.net import Net
    from deepaco_amd.net import GraphData
    torch.manual_seed(0)
    net = Net().to(dev())
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    net.eval()
    n, E = 37, 333
    gen = torch.Generator().manual_seed(1)
    src = torch.randint(0, n - 3, (E,), generator=gen)        # nodes n-3.. have no out-edges
    dst = torch.randint(0, n, (E,), generator=gen)
    pyg = GraphData(x=torch.rand(n, 1, generator=gen), edge_index=torch.stack([src, dst]),
                    edge_attr=torch.rand(E, 1, generator=gen)).to(dev())
    with torch.no_grad():
        heu = net(pyg)
    w = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items() if v.dtype.is_floating_point and v.numel()}
    ref = ognn.net_forward(w, pyg.x.cpu().numpy(), pyg.edge_index.cpu().numpy(), pyg.edge_attr.cpu().numpy())
    np.testing.assert_allclose(heu.cpu().numpy(), ref, atol=ATOL_HEU, rtol=1e-4)


def test_training_step_end_to_end():
    """tsp_nls/train.py:15-44 train_instance on a small instance: Net (torch ops, autograd) -> heuristic ->
    ACO.sample (HIP forward + HIP backward) -> sample_2opt (HIP) -> REINFORCE loss -> AdamW step."""
    from deepaco_amd.tsp_nls.net import Net
    from deepaco_amd.tsp_nls.aco import ACO
    from deepaco_amd.tsp_nls.utils import gen_pyg_data
    torch.manual_seed(1234)
    net = Net().to(dev())
    opt = torch.optim.AdamW(net.parameters(), lr=3e-4)
    coords = torch.rand(40, 2, device=dev())
    pyg, distances = gen_pyg_data(coords, k_sparse=8, start_node=0)
    before = [p.detach().clone() for p in net.parameters()]
    net.train()
    heu_vec = net(pyg)
    heu_mat = net.reshape(pyg, heu_vec) + 1e-10
    aco = ACO(n_ants=12, heuristic=heu_mat, distances=distances, device="cuda:0", local_search='nls')
    costs, log_probs, paths = aco.sample()
    costs_2opt, _ = aco.sample_2opt(paths)
    cost = (costs_2opt - costs_2opt.mean()) * 0.95 + (costs - costs.mean()) * 0.05
    loss = torch.sum(cost.detach() * log_probs.sum(dim=0)) / aco.n_ants
    opt.zero_grad()
    loss.backward()
    gn = torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=3.0, norm_type=2)
    assert torch.isfinite(gn) and float(gn) > 0
    opt.step()
    changed = sum(int(not torch.equal(a, b.detach())) for a, b in zip(before, net.parameters()))
    assert changed > 100
    assert bool((costs_2opt <= costs + 1e-4).all())


def test_batched_training_step_on_device():
    """pipeline.train_tsp_nls_batch: tsp_nls/train.py's step for B instances at once; parameters move, the loss is
    finite, local search only lowers costs.  A few steps at a fixed seed lower the mean sampled cost on held instances."""
    from deepaco_amd.pipeline import train_tsp_nls_batch
    from deepaco_amd.tsp_nls.net import Net
    torch.manual_seed(11)
    net = Net().to(dev())
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    before = [p.detach().clone() for p in net.parameters()]
    coords = torch.rand(6, 30, 2, device=dev())
    first = None
    for step in range(12):
        loss, c, c_ls = train_tsp_nls_batch(net, opt, coords, n_ants=16, k_sparse=8, seed=5, it=step)
        assert torch.isfinite(loss) and float(c_ls) <= float(c) + 1e-5
        first = float(c) if first is None else first
    changed = sum(int(not torch.equal(a, b.detach())) for a, b in zip(before, net.parameters()))
    assert changed > 100
    assert float(c) < first               # the policy improved on the instances it trains on


def test_flat_block_training_equals_the_parameter_list():
    """Net.flatten_parameters on the device: the training forward takes the block as it is, the backward's flat gradient is the
    block's .grad and every parameter's .grad a view of it -- the same heuristic and the same gradients as the network whose
    parameters are packed with torch.cat every step (bit for bit: the same kernels on the same values)."""
    import copy
    from deepaco_amd import engine
    from deepaco_amd.tsp_nls.net import Net
    torch.manual_seed(21)
    B, n, k = 3, 40, 8
    net = Net().to(dev()).train()
    ref = copy.deepcopy(net)
    block = net.flatten_parameters()
    coords = torch.rand(B, n, 2, device=dev())
    _, ei, ea = engine.tsp_knn_graph(coords, k, want_dist=False)
    x = torch.zeros(B, n, 1, device=dev())
    x[:, 0] = 1.0
    coef = torch.randn(B, n * k, device=dev())
    monkey_env = os.environ.get("DACO_GNN_TRAIN_GATHER")
    os.environ["DACO_GNN_TRAIN_GATHER"] = "1"                  # (CSR row sums: no f32 atomics, the backward is deterministic)
    try:
        heu = net.forward_batch_train(x, ei, ea, k_sparse=k)
        torch.sum(heu * coef).backward()
        heu_r = ref.forward_batch_train(x, ei, ea, k_sparse=k)
        torch.sum(heu_r * coef).backward()
    finally:
        if monkey_env is None:
            del os.environ["DACO_GNN_TRAIN_GATHER"]
        else:
            os.environ["DACO_GNN_TRAIN_GATHER"] = monkey_env
    assert torch.equal(heu.detach(), heu_r.detach())
    assert block.grad is not None and block.grad.shape == block.shape
    gmax = max(float(q.grad.abs().max()) for q in ref.parameters() if q.grad is not None)
    for (k1, p), (k2, q) in zip(net.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            continue
        assert p.grad is not None, k1
        assert block.grad.data_ptr() <= p.grad.data_ptr() < block.grad.data_ptr() + block.numel() * 4, k1
        # (the weight gradients are sums of MFMA tiles flushed with f32 atomics: equal up to the order of a few additions)
        torch.testing.assert_close(p.grad, q.grad, rtol=2e-5, atol=2e-6 * gmax, msg=k1)
    for (k1, v), (k2, w) in zip(net.state_dict().items(), ref.state_dict().items()):
        if "running_" in k1:
            assert torch.equal(v, w), k1


@pytest.mark.parametrize("graph", [False, True])
def test_trainer_draws_what_the_eager_step_draws(graph):
    """pipeline.TspNlsTrainer (the step as one captured HIP graph; graph=False: the same step eagerly on the flat block) against
    pipeline.train_tsp_nls_batch(..., seed, it = s) step by step.  With a learning rate of zero AdamW leaves the parameters where
    they are, so every step is a function of (parameters, coordinates, seed, s) alone: the captured step s -- whose Philox
    iteration counter lives in device memory and is advanced inside the graph -- must sample the tours of the eager step s:
    the same mean costs, the same loss."""
    import copy
    from deepaco_amd.pipeline import TspNlsTrainer, train_tsp_nls_batch
    from deepaco_amd.tsp_nls.net import Net
    torch.manual_seed(31)
    B, n, A, k = 4, 30, 12, 8
    net = Net().to(dev())
    eager_net = copy.deepcopy(net)
    opt = torch.optim.AdamW(eager_net.parameters(), lr=0.0)
    trainer = TspNlsTrainer(net, B, n, A, k, lr=0.0, seed=9, graph=graph)
    gen = torch.Generator().manual_seed(4)
    seen = []
    for s in range(6):
        coords = torch.rand(B, n, 2, generator=gen).to(dev())
        loss_g, c_g, cls_g = (float(v) for v in trainer.step(coords))
        loss_e, c_e, cls_e = (float(v) for v in train_tsp_nls_batch(eager_net, opt, coords, A, k, seed=9, it=s))
        assert abs(c_g - c_e) <= 1e-6 * abs(c_e) and abs(cls_g - cls_e) <= 1e-6 * abs(cls_e), (s, c_g, c_e, cls_g, cls_e)
        assert abs(loss_g - loss_e) <= 2e-4 * max(abs(loss_e), 1e-3), (s, loss_g, loss_e)
        seen.append(c_g)
    assert (trainer._graph is not None) == graph
    assert int(trainer.it_dev) == 6
    assert len(set(seen)) == 6                                         # (fresh instances and fresh draws every step)
    # the same coordinates twice: the counter advanced, the tours differ
    again = [float(trainer.step(coords)[1]) for _ in range(2)]
    assert again[0] != again[1]
    for p, q in zip(net.parameters(), eager_net.parameters()):         # lr = 0: nobody moved
        assert torch.equal(p.detach(), q.detach())


def test_trainer_captured_steps_train():
    """The captured step with a real learning rate: parameters move (through the block: the module tree's views follow), the
    running statistics advance, local search only lowers costs, and a dozen steps on the same instances lower their sampled
    cost -- what test_batched_training_step_on_device asks of the eager step."""
    from deepaco_amd.pipeline import TspNlsTrainer
    from deepaco_amd.tsp_nls.net import Net
    torch.manual_seed(11)
    net = Net().to(dev())
    before = [p.detach().clone() for p in net.parameters()]
    rm_before = net.emb_net.e_bns[3].module.running_mean.clone()
    trainer = TspNlsTrainer(net, 6, 30, 16, 8, lr=1e-3, seed=5, graph=True)
    coords = torch.rand(6, 30, 2, device=dev())
    first = None
    for step in range(12):
        loss, c, c_ls = trainer.step(coords)
        assert torch.isfinite(loss) and float(c_ls) <= float(c) + 1e-5
        first = float(c) if first is None else first
    assert trainer._graph is not None
    changed = sum(int(not torch.equal(a, b.detach())) for a, b in zip(before, net.parameters()))
    assert changed > 100
    assert not torch.equal(rm_before, net.emb_net.e_bns[3].module.running_mean)
    assert int(net.emb_net.e_bns[3].module.num_batches_tracked) == 12 * 6
    assert float(c) < first
    assert torch.equal(net.pack_params_train().detach(), trainer.block.detach())
    # eval-mode inference sees the trained values (the folded block is rebuilt from the views' version counter)
    net.eval()
    from deepaco_amd import engine
    _, ei, ea = engine.tsp_knn_graph(coords, 8, want_dist=False)
    x = torch.zeros(6, 30, 1, device=dev())
    x[:, 0] = 1.0
    h1 = net.forward_batch(x, ei, ea, k_sparse=8)
    assert torch.isfinite(h1).all()


@pytest.mark.parametrize("name", ["g5_net_tsp_tsp100", "g5_net_tsp_tsp20", "g5_net_tsp_nls_tsp100"])
def test_batched_graph_construction_matches_reference(name):
    """daco_tsp_knn_graph (one launch for a batch) == the reference's gen_pyg_data on captured instances."""
    from deepaco_amd.tsp.utils import gen_pyg_data_batch
    g = load_golden(name)
    k = int(g["k_sparse"])
    coords = torch.from_numpy(g["coords"]).to(dev())
    extra = torch.rand(3, coords.shape[0], 2, generator=torch.Generator().manual_seed(1)).to(dev())
    batch = torch.cat((coords[None], extra), 0)
    out = gen_pyg_data_batch(batch, k, start_node=0 if "nls" in name else None)
    pyg, dist = out[0]
    assert np.array_equal(pyg.edge_index.cpu().numpy(), g["edge_index"])
    np.testing.assert_allclose(pyg.edge_attr.cpu().numpy(), g["edge_attr"], rtol=1e-6)
    np.testing.assert_allclose(dist.cpu().numpy(), g["distances"], rtol=1e-6)
    assert np.array_equal(pyg.x.cpu().numpy(), g["x"])
    # the other instances agree with the per-instance torch construction
    from deepaco_amd.tsp.utils import gen_pyg_data
    for b in range(1, 4):
        ref, rd = gen_pyg_data(batch[b], k)
        assert torch.equal(out[b][0].edge_index, ref.edge_index)
        torch.testing.assert_close(out[b][0].edge_attr, ref.edge_attr, rtol=1e-6, atol=0)


@pytest.mark.parametrize("B,n,k", [(5, 60, 7), (3, 200, 100), (1, 130, 64), (2, 150, 65)])
def test_knn_graph_merged_int32_form_and_the_forward_that_uses_it(B, n, k):
    """engine.tsp_knn_graph also writes the batch as one block-diagonal int32 graph (daco_tsp_knn_graph_csr); Net.forward_batch
    takes it as it is.  Held against (a) torch.topk per instance (k > 64: the kernel stores its edges 64 rounds at a time),
    (b) the merged arrays derived from the int64 edge_index, (c) the forward on a copy of edge_index that carries nothing."""
    from deepaco_amd import engine
    from deepaco_amd.net import _merge_graphs
    from deepaco_amd.tsp.net import Net
    torch.manual_seed(B * 1000 + n)
    coords = torch.rand(B, n, 2, device=dev())
    dist, ei, ea = engine.tsp_knn_graph(coords, k)
    ref_d, ref_i = torch.topk(dist, k=k, dim=2, largest=False)
    assert torch.equal(ei[:, 1].reshape(B, n, k), ref_i)
    assert torch.equal(ea.reshape(B, n, k), ref_d)
    assert torch.equal(ei[:, 0].reshape(B, n, k), torch.arange(n, device=dev()).view(1, n, 1).expand(B, n, k))
    src32, dst32, n_, k_, version = ei._daco_csr
    off = (torch.arange(B, device=dev()) * n).view(B, 1)
    assert (n_, k_, version) == (n, k, ei._version)
    assert torch.equal(src32.long(), (ei[:, 0] + off).reshape(-1)) and torch.equal(dst32.long(), (ei[:, 1] + off).reshape(-1))
    fast = _merge_graphs(coords, ei, ea, k_sparse=k)
    assert fast.edge_index is None and fast._daco_graph[0] is src32
    plain = ei.clone()
    slow = _merge_graphs(coords, plain, ea, k_sparse=k)
    assert slow.edge_index is not None
    for a, b in zip(fast._daco_graph[:3], slow._daco_graph[:3]):
        assert torch.equal(a, b)
    torch.manual_seed(2)
    net = Net().to(dev()).eval()
    assert torch.equal(net.forward_batch(coords, ei, ea, k_sparse=k), net.forward_batch(coords, plain, ea, k_sparse=k))
    # a write to the tensor makes the attached arrays stale: the version counter sends the call down the derived path
    ei[0, 1, 0] = ei[0, 1, 1]
    assert _merge_graphs(coords, ei, ea, k_sparse=k).edge_index is not None


@pytest.mark.parametrize("B,n,k,eps", [(4, 60, 7, 1e-10), (1, 45, 5, 1e-10), (3, 131, 20, 0.0), (2, 33, 32, 1e-10)])
def test_reshape_batch_on_the_device_equals_the_indexed_assignment(B, n, k, eps):
    """Net.reshape_batch(..., eps) outside autograd = daco_heu_matrix: zeros / matrix[src, dst] = heu / + eps (tsp/net.py:94-102
    and the callers' `+ 1e-10`), bit for bit, including sizes whose element count is not a multiple of four."""
    from deepaco_amd import engine
    from deepaco_amd.net import Net
    torch.manual_seed(n)
    coords = torch.rand(B, n, 2, device=dev())
    _, ei, _ = engine.tsp_knn_graph(coords, k, want_dist=False)
    heu = torch.rand(B, n * k, device=dev())
    ref = torch.zeros((B, n, n), device=dev())
    bidx = torch.arange(B, device=dev()).view(B, 1).expand(B, n * k)
    ref[bidx, ei[:, 0], ei[:, 1]] = heu
    ref = ref + eps
    got = Net.reshape_batch(n, ei, heu, eps=eps)
    assert torch.equal(got, ref)
    # under autograd the torch ops run (the gradient flows back through the indexed assignment)
    h2 = heu.clone().requires_grad_(True)
    m2 = Net.reshape_batch(n, ei, h2, eps=eps)
    assert torch.equal(m2.detach(), ref)
    m2.sum().backward()
    assert torch.equal(h2.grad, torch.ones_like(heu))
    bad = ei.clone()
    bad[0, 1, 3] = n
    with pytest.raises(IndexError):
        engine.heu_matrix(n, bad, heu, check=True)
    skipped = engine.heu_matrix(n, bad, heu)                      # (unchecked: the edge is left out)
    assert float(skipped[0, int(ei[0, 0, 3]), int(ei[0, 1, 3])]) == 0.0
    # the class surface validates a foreign edge_index once and raises like the indexed assignment, negative ids too
    with pytest.raises(IndexError):
        Net.reshape_batch(n, bad, heu, eps=eps)
    neg = ei.clone()
    neg[B - 1, 0, 0] = -1
    with pytest.raises(IndexError):
        Net.reshape_batch(n, neg, heu, eps=eps)
    good = ei.clone()                                             # (no attached CSR: validated by aminmax, then cached)
    assert torch.equal(Net.reshape_batch(n, good, heu, eps=eps), ref) and good._daco_ids_ok == (n, good._version)


def test_batched_forward_equals_per_graph():
    """B graphs side by side in one pass == B separate forwards (eval mode), and the batched reshape."""
    from deepaco_amd import engine
    from deepaco_amd.net import GraphData
    from deepaco_amd.tsp.net import Net
    torch.manual_seed(3)
    dev = torch.device("cuda:0")
    net = Net().to(dev).eval()
    for m in net.modules():                     # non-trivial running statistics
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    B, n, k = 5, 60, 12
    coords = torch.rand(B, n, 2, device=dev)
    _, ei, ea = engine.tsp_knn_graph(coords, k, want_dist=False)
    heu = net.forward_batch(coords, ei, ea)
    assert heu.shape == (B, n * k)
    for b in range(B):
        with torch.no_grad():
            one = net(GraphData(x=coords[b], edge_index=ei[b], edge_attr=ea[b]))
        torch.testing.assert_close(heu[b], one.view(-1), rtol=1e-6, atol=2e-7)     # (tile position moves the last bit)
    mats = Net.reshape_batch(n, ei, heu)
    one = Net.reshape(GraphData(x=coords[2], edge_index=ei[2], edge_attr=ea[2]), heu[2])
    assert torch.equal(mats[2], one)      # same heu values scattered the same way


def test_batched_inference_pipeline():
    """coords -> kNN graph -> GNN (batched) -> colonies: equals building each colony from per-graph pieces."""
    from deepaco_amd import engine
    from deepaco_amd.net import GraphData
    from deepaco_amd.pipeline import infer_tsp_batch, EPS
    from deepaco_amd.tsp.net import Net
    torch.manual_seed(5)
    dev = torch.device("cuda:0")
    net = Net().to(dev).eval()
    B, n, k, A = 3, 40, 8, 16
    coords = torch.rand(B, n, 2, device=dev)
    best, colony = infer_tsp_batch(coords, A, [1, 4], k, net=net, seed=9)
    assert best.shape == (2, B) and bool((best[1] <= best[0]).all())
    dist, ei, ea = engine.tsp_knn_graph(coords, k)
    heu = torch.stack([Net.reshape(GraphData(x=coords[b], edge_index=ei[b], edge_attr=ea[b]),
                                   net(GraphData(x=coords[b], edge_index=ei[b], edge_attr=ea[b])).view(-1)) for b in range(B)])
    ref = engine.BatchedTSP(dist, n_ants=A, heuristic=heu + EPS, seed=9)
    torch.testing.assert_close(colony.heuristic, ref.heuristic, rtol=1e-6, atol=1e-12)
    vb, _ = infer_tsp_batch(coords, A, [3], k, net=None, seed=9)          # vanilla heuristic
    assert bool((vb > 0).all())


@pytest.mark.parametrize("npw", [4, 7, 11, 16])
def test_fused_layer_kernel_vs_oracle_on_ragged_sorted_graph(npw, monkeypatch):
    """The fused layer kernel (src-sorted edge lists; a wave owns nodes and their out-edges) on a graph with uneven
    degrees, nodes without out-edges at the front, in the middle and at the end, and a node count that is not a multiple
    of the nodes-per-wave: against the numpy restatement of the reference's forward (tsp/net.py:27-45)."""
    from deepaco_amd.cvrp.net import Net
    from deepaco_amd.net import GraphData
    monkeypatch.setenv("DACO_GNN_SPLIT_MIN_EDGES", "1")
    monkeypatch.setenv("DACO_GNN_FUSED_NPW", str(npw))
    torch.manual_seed(2)
    net = Net().to(dev())
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    net.eval()
    n = 203
    gen = torch.Generator().manual_seed(7)
    deg = torch.randint(0, 90, (n,), generator=gen)
    deg[:3] = 0; deg[70:75] = 0; deg[-2:] = 0; deg[100] = 200                 # empty runs and one node spanning several tiles
    src = torch.repeat_interleave(torch.arange(n), deg)
    E = int(src.numel())
    dst = torch.randint(0, n, (E,), generator=gen)
    pyg = GraphData(x=torch.rand(n, 1, generator=gen), edge_index=torch.stack([src, dst]),
                    edge_attr=torch.rand(E, 1, generator=gen)).to(dev())
    with torch.no_grad():
        heu = net(pyg)
    w = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items() if v.dtype.is_floating_point and v.numel()}
    ref = ognn.net_forward(w, pyg.x.cpu().numpy(), pyg.edge_index.cpu().numpy(), pyg.edge_attr.cpu().numpy())
    np.testing.assert_allclose(heu.cpu().numpy(), ref, atol=ATOL_HEU, rtol=1e-4)
    monkeypatch.setenv("DACO_GNN_FUSED_NPW", "0")                              # the split kernels on the same graph
    with torch.no_grad():
        heu_split = net(pyg)
    torch.testing.assert_close(heu, heu_split, rtol=1e-5, atol=2e-6)


def test_fused_layer_kernel_at_bench_size_equals_per_graph(monkeypatch):
    """9 graphs of TSP-500 (k = 50) side by side take the fused layer kernel (E >= 200 000); each graph alone takes the
    single-launch layer kernel: same heuristic up to the aggregation's summation order.  The output head inside the last
    layer's launch (round 6) against the separate head launch: the same bits."""
    from deepaco_amd import engine
    from deepaco_amd.net import GraphData
    from deepaco_amd.tsp.net import Net
    torch.manual_seed(4)
    d = torch.device("cuda:0")
    net = Net().to(d).eval()
    B, n, k = 9, 500, 50
    coords = torch.rand(B, n, 2, device=d)
    _, ei, ea = engine.tsp_knn_graph(coords, k, want_dist=False)
    heu = net.forward_batch(coords, ei, ea, k_sparse=k)
    assert torch.equal(heu, net.forward_batch(coords, ei, ea))          # the CSR shortcut describes the same graph
    monkeypatch.setenv("DACO_GNN_HEAD_FUSED", "0")
    assert torch.equal(heu, net.forward_batch(coords, ei, ea, k_sparse=k))      # head as its own launch over the stored edge state
    monkeypatch.delenv("DACO_GNN_HEAD_FUSED")
    for b in (0, 4, 8):
        one = net(GraphData(x=coords[b], edge_index=ei[b], edge_attr=ea[b]))
        torch.testing.assert_close(heu[b], one.view(-1), rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("name,wname,B", [("g5c_net_tsp_tsp500", "w_tsp_tsp500", 9), ("g5c_net_tsp_nls_tsp1000", "w_tsp_nls_tsp1000", 3)])
def test_net_at_bench_size_matches_the_reference(name, wname, B):
    """VERDICT r5 missing 3: the reference's own output at the sizes the bench runs the network (g5c fixtures: the imported
    reference with pretrained/tsp/tsp500.pt at n = 500 / k = 50, tsp/train.ipynb:268; pretrained/tsp_nls/tsp1000.pt at n = 1000 /
    k = 100) -- the per-graph kernels at 1e-5, and the FUSED layer kernel (B copies of the graph side by side, E >= 200 000: what
    bench.py's forward launches) at 1e-5 for every copy; train mode at the torch tolerance; the graph the device builds
    (engine.tsp_knn_graph) against the reference's edge list."""
    from deepaco_amd import engine
    from deepaco_amd.net import GraphData
    g = load_golden(name)
    net = make_net(name)
    load_weights(net, load_golden(wname))
    net = net.to(dev()).eval()
    ei = torch.from_numpy(g["edge_index"].astype(np.int64))
    pyg = GraphData(x=torch.from_numpy(g["x"]), edge_index=ei, edge_attr=torch.from_numpy(g["edge_attr"]).view(-1, 1)).to(dev())
    with torch.no_grad():
        heu = net(pyg)
    # (measured: n = 500 every element within 1e-5; n = 1000: 13 of 100 000 elements beyond it, the largest 6.7e-5 -- the float64
    # restatement itself has 2 beyond it, 3.9e-5, against the reference's float32 output)
    assert_close_mostly(heu.cpu().numpy(), g["heu_eval"], atol=ATOL_HEU, frac=5e-4)
    _, emb = net.forward_hip(pyg, return_embedding=True)
    np.testing.assert_allclose(emb.cpu().numpy()[g["emb_rows"]], g["emb_eval_rows"], atol=3e-4, rtol=3e-4)
    # the fused layer kernel: B copies of the graph in one pass
    n, E = g["x"].shape[0], ei.shape[1]
    xb = pyg.x.unsqueeze(0).expand(B, n, -1).contiguous()
    eib = pyg.edge_index.unsqueeze(0).expand(B, 2, E).contiguous()
    eab = pyg.edge_attr.view(1, E).expand(B, E).contiguous()
    assert B * E >= 200000
    hb = net.forward_batch(xb, eib, eab, k_sparse=int(g["k_sparse"]))
    for b in range(B):
        assert_close_mostly(hb[b].cpu().numpy(), g["heu_eval"], atol=ATOL_HEU, frac=5e-4, err_msg=f"copy {b}")
    # the device's own graph of these coordinates is the reference's (tsp/utils.py:16-36 / tsp_nls/utils.py:17-45)
    coords = torch.from_numpy(g["coords"]).to(dev())
    _, ei_dev, ea_dev = engine.tsp_knn_graph(coords[None], int(g["k_sparse"]), want_dist=False)
    assert torch.equal(ei_dev[0].cpu(), ei)
    np.testing.assert_allclose(ea_dev[0].cpu().numpy().reshape(-1), g["edge_attr"], rtol=0, atol=1e-7)
    net.train()
    with torch.no_grad():
        ht = net(pyg)
    assert_close_mostly(ht.cpu().numpy(), g["heu_train"], atol=ATOL_HEU, rtol=5e-4, cap=ATOL_TORCH)


# ------------------------------------------------------------------ per-directory nets (b-double-dagger: `from net import Net` everywhere)
@pytest.mark.parametrize("name", ["op", "pctsp", "sop", "smtwtp", "bpp", "mkp", "cvrp_nls"])
def test_sibling_nets_hip_equals_torch_ops(name):
    """The networks of the sibling directories (other feature widths; sop / smtwtp without the node update,
    sop/net.py:43) on the HIP kernels against the same module evaluated with torch ops: eval-mode forward, and
    training-mode forward + parameter gradients."""
    import importlib
    from deepaco_amd.net import GraphData
    Net = importlib.import_module(f"deepaco_amd.{name}.net").Net
    torch.manual_seed(5)
    net = Net().to(dev())
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    feats = net.emb_net.v_lin0.in_features
    n, k = 30, 6
    gen = torch.Generator().manual_seed(2)
    src = torch.repeat_interleave(torch.arange(n), k)
    dst = torch.randint(0, n, (n * k,), generator=gen)
    pyg = GraphData(x=torch.rand(n, feats, generator=gen), edge_index=torch.stack([src, dst]),
                    edge_attr=torch.rand(n * k, 1, generator=gen)).to(dev())
    net.eval()
    with torch.no_grad():
        hip = net(pyg)
        ref = net.par_net_heu(net.emb_net(pyg.x, pyg.edge_index, pyg.edge_attr))
    torch.testing.assert_close(hip, ref, atol=ATOL_HEU, rtol=1e-4)
    # training mode: HIP kernels vs torch ops + autograd
    coef = torch.randn(n * k, generator=gen).to(dev())
    net.train()
    grads, stats = {}, {}
    state = {k_: v.clone() for k_, v in net.state_dict().items()}
    for backend in ("hip", "torch"):
        net.load_state_dict(state)
        net.zero_grad()
        heu = forward_as(net, pyg, backend)
        torch.sum(heu * coef).backward()
        grads[backend] = (heu.detach().clone(), {k_: p.grad.clone() for k_, p in net.named_parameters() if p.grad is not None})
        stats[backend] = {k_: v.clone() for k_, v in net.state_dict().items() if "running_" in k_}
    torch.testing.assert_close(grads["hip"][0], grads["torch"][0], atol=ATOL_TORCH, rtol=5e-4)
    gmax = max(float(v.abs().max()) for v in grads["torch"][1].values())
    for k_ in grads["torch"][1]:
        if _zero_in_exact_arithmetic(k_):
            continue
        _grad_close(grads["hip"][1][k_].cpu().numpy(), grads["torch"][1][k_].cpu().numpy(), k_, rel=2e-3, floor=4e-6 * gmax)
    for k_ in stats["torch"]:
        torch.testing.assert_close(stats["hip"][k_], stats["torch"][k_], rtol=1e-4, atol=1e-6)
    if not net.emb_net.node_update:      # the unused node-update modules get NO gradient (None, as in the reference where they are
        for k_ in grads["hip"][1]:       # never called: AdamW's weight decay must not touch them), not a zero one
            assert not (".v_lins1." in k_ or ".v_lins2." in k_ or ".v_bns." in k_), k_


@pytest.mark.parametrize("name", ["g5b_net_sop_sop20", "g5b_net_op_op100", "g5b_net_mkp_mkp300"])
def test_sibling_nets_match_the_reference_checkpoints(name):
    """g5b (tests/golden/gen_g5b_sibling_nets.py): the reference's own sop / op / mkp Net.forward with the checkpoints it
    ships, on an instance built by its own utils.py.  The drop-in modules of those directories load the checkpoint keys
    unchanged and the HIP kernels reproduce the eval-mode heuristic at 1e-5 (feature widths 1 / 2 / 5; sop without the
    node update, sop/net.py:43), the embedding, the training-mode forward (both backends) and the reshaped matrix."""
    import importlib
    g = load_golden(name)
    Net = importlib.import_module(f"deepaco_amd.{name.split('_')[2]}.net").Net
    net = Net()
    missing, unexpected = load_weights(net, g)
    assert unexpected == [] and all(k.endswith("num_batches_tracked") or k.endswith("_dummy") for k in missing), (missing, unexpected)
    assert net.emb_net.node_update == ("sop" not in name)
    net = net.to(dev()).eval()
    pyg = graph(g)
    with torch.no_grad():
        heu = net(pyg)
    np.testing.assert_allclose(heu.cpu().numpy(), g["heu_eval"], atol=ATOL_HEU, rtol=1e-4)
    _, emb = net.forward_hip(pyg, return_embedding=True)
    np.testing.assert_allclose(emb.cpu().numpy(), g["emb_eval"], atol=3e-4, rtol=3e-4)
    mat = net.reshape(pyg, heu)
    np.testing.assert_allclose(mat.cpu().numpy(), g["heu_mat"], atol=ATOL_HEU, rtol=1e-4)
    net.train()
    for backend in ("hip", "torch"):
        with torch.no_grad():
            ht = forward_as(net, graph(g), backend)
        np.testing.assert_allclose(ht.cpu().numpy(), g["heu_train"], atol=ATOL_TORCH, rtol=5e-4, err_msg=backend)


def test_cvrp_nls_train_instance_on_the_drop_in():
    """The body of cvrp_nls/train.py:train_instance (:15-50) on the drop-in modules of deepaco_amd/cvrp_nls: instance and
    sparse graph from utils (float64 data, capacity 1), Net in training mode, ACO(swapstar=True, positions=...),
    sample_nls(), the REINFORCE loss on the improved costs, clipped AdamW step."""
    from deepaco_amd.cvrp_nls.net import Net
    from deepaco_amd.cvrp_nls.aco import ACO
    from deepaco_amd.cvrp_nls.utils import gen_instance, gen_pyg_data
    EPS = 1e-5
    torch.manual_seed(77)
    n = 30
    model = Net().to(dev())
    optimizer = torch.optim.AdamW(model.parameters(), lr=1e-3)
    data = []
    for _ in range(2):
        demands, distances, positions = gen_instance(n, "cuda:0", True)
        data.append((gen_pyg_data(demands, distances, "cuda:0", k_sparse=max(n // 5, 4)), demands, distances, positions))
    before = [p.detach().clone() for p in model.parameters()]
    model.train()
    sum_loss, count = 0.0, 0
    for pyg_data, demands, distances, positions in data:
        heu_vec = model(pyg_data)
        heu_mat = model.reshape(pyg_data, heu_vec) + EPS
        aco = ACO(n_ants=10, distances=distances, demand=demands, heuristic=heu_mat, device="cuda:0", swapstar=True,
                  positions=positions)
        costs_2opt, log_probs, costs_raw = aco.sample_nls()
        assert bool((costs_2opt <= costs_raw + 1e-4).all()) and float(costs_2opt.mean()) < float(costs_raw.mean())
        cost = costs_2opt - costs_2opt.mean()
        sum_loss = sum_loss + torch.sum(cost.detach() * log_probs.sum(dim=0)) / aco.n_ants
        count += 1
    sum_loss = sum_loss / count
    optimizer.zero_grad()
    sum_loss.backward()
    gn = torch.nn.utils.clip_grad_norm_(parameters=model.parameters(), max_norm=3.0, norm_type=2)
    assert torch.isfinite(gn) and float(gn) > 0
    optimizer.step()
    assert sum(int(not torch.equal(a, b.detach())) for a, b in zip(before, model.parameters())) > 100


@pytest.mark.parametrize("n,n_ants,k_sparse,wfix", [(20, 20, 10, "g5_net_tsp_tsp20"), (100, 20, 20, "g5_net_tsp_tsp100"),
                                                    (500, 50, 50, "w_tsp_tsp500")])
def test_notebook_validation_protocol_reproduces_the_published_costs(n, n_ants, k_sparse, wfix):
    """SURVEY 8(c)'s end-to-end check of the network path (the fixtures were generated through a torch_geometric
    stand-in): tsp/train.ipynb's `validation` on the reference's own validation set with the checkpoint it saved --
    infer_instance (cell 1): heuristic = Net(pyg) + 1e-10, ACO(n_ants).sample(), run(T = 5) -- prints, for the final
    epoch, (avg sample cost, best sample cost, best ACO cost) = g10 `notebook<n>`.  The drop-in classes on the same 100
    instances and weights must land on the same three numbers up to sampling noise (1.5 %; a wrong aggregation or
    normalisation in the network moves them by far more: an untrained net gives 64.9 at n = 500)."""
    from deepaco_amd.tsp.net import Net
    from deepaco_amd.tsp.aco import ACO
    from deepaco_amd.tsp.utils import gen_pyg_data
    g = np.load(os.path.join(GOLDEN, "g10_val_tsp.npz"))
    wz = np.load(os.path.join(GOLDEN, wfix + ".npz"))
    prefix = "w__"
    net = Net()
    net.load_state_dict({k[len(prefix):]: torch.from_numpy(wz[k]) for k in wz.files if k.startswith(prefix)}, strict=False)
    net = net.to(dev()).eval()
    torch.manual_seed(1234)
    sums = np.zeros(3)
    coords = torch.from_numpy(g[f"coords{n}"]).to(dev())
    with torch.no_grad():
        for inst in coords:
            pyg, distances = gen_pyg_data(inst, k_sparse=k_sparse)
            heu_mat = net.reshape(pyg, net(pyg)) + 1e-10
            aco = ACO(n_ants=n_ants, heuristic=heu_mat, distances=distances, device="cuda:0")
            costs, _ = aco.sample()
            aco.run(n_iterations=5)
            sums += (float(costs.mean()), float(costs.min()), float(aco.lowest_cost))
    got, want = sums / len(coords), g[f"notebook{n}"]
    np.testing.assert_allclose(got, want, rtol=0.015)



def _small_graph(n=45, E=520, seed=4):
    from deepaco_amd.net import GraphData
    gen = torch.Generator().manual_seed(seed)
    src = torch.sort(torch.randint(0, n, (E,), generator=gen)).values
    dst = torch.randint(0, n, (E,), generator=gen)
    pyg = GraphData(x=torch.rand(n, 2, generator=gen), edge_index=torch.stack([src, dst]),
                    edge_attr=torch.rand(E, 1, generator=gen)).to(dev())
    return pyg, torch.randn(E, generator=gen).to(dev())


def _grads(net):
    return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}


def test_eval_mode_under_autograd_runs_on_the_kernels():
    """A module in eval() whose output needs a gradient (this used to evaluate the module tree as torch
    ops): the training kernels with the running statistics as constants -- value = the inference kernels', gradient = torch
    autograd through the eval-mode module tree; the running statistics stay untouched."""
    from deepaco_amd.tsp.net import Net
    torch.manual_seed(5)
    net = Net().to(dev())
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.uniform_(-0.3, 0.3)
            m.running_var.uniform_(0.4, 1.6)
    net.eval()
    pyg, coef = _small_graph()
    before = {k: v.clone() for k, v in net.state_dict().items() if "running_" in k or "num_batches" in k}
    with torch.no_grad():
        inf = net(pyg)
    net.zero_grad()
    heu = net(pyg)
    assert heu.requires_grad
    torch.sum(heu * coef).backward()
    g_hip = _grads(net)
    net.zero_grad()
    ref = forward_as(net, pyg, "torch")
    torch.sum(ref * coef).backward()
    g_ref = _grads(net)
    torch.testing.assert_close(heu.detach(), inf, atol=ATOL_HEU, rtol=1e-4)
    torch.testing.assert_close(heu.detach(), ref.detach(), atol=ATOL_TORCH, rtol=5e-4)
    gmax = max(float(v.abs().max()) for v in g_ref.values())
    for k in g_ref:
        _grad_close(g_hip[k].cpu().numpy(), g_ref[k].cpu().numpy(), k, rel=2e-3, floor=4e-6 * gmax)
    for k, v in before.items():
        assert torch.equal(net.state_dict()[k], v), k


@pytest.mark.parametrize("config", ["momentum_none", "no_running_stats"])
def test_batchnorm_configurations_away_from_the_default(config):
    """BatchNorm1d(momentum=None) (cumulative average of the statistics) and track_running_stats=False (batch statistics in
    eval mode too): the kernels plus the host-side bookkeeping against the module tree as torch ops."""
    from deepaco_amd.tsp.net import Net
    torch.manual_seed(6)
    net = Net().to(dev())
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            if config == "momentum_none":
                m.momentum = None
            else:
                m.track_running_stats = False
                m.running_mean = None
                m.running_var = None
                m.num_batches_tracked = None
    pyg, coef = _small_graph(seed=8)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    out = {}
    for backend in ("hip", "torch"):
        net.load_state_dict(state)
        net.train()
        res = []
        for _ in range(3):                                # three training forwards: the cumulative average moves each time
            net.zero_grad()
            heu = forward_as(net, pyg, backend)
            torch.sum(heu * coef).backward()
            res.append(heu.detach().clone())
        net.eval()
        res.append(forward_as(net, pyg, backend).detach().clone())         # eval mode (under autograd)
        with torch.no_grad():                                              # eval mode without a gradient: the inference path (ADVICE r5:
            res.append(forward_as(net, pyg, backend).clone())              # without running statistics there is nothing to fold)
        out[backend] = (res, _grads(net), {k: v.clone() for k, v in net.state_dict().items() if "running_" in k})
    for a, b in zip(out["hip"][0], out["torch"][0]):
        torch.testing.assert_close(a, b, atol=ATOL_TORCH, rtol=5e-4)
    for k, v in out["torch"][2].items():
        torch.testing.assert_close(out["hip"][2][k], v, rtol=1e-4, atol=1e-6)
    gmax = max(float(v.abs().max()) for v in out["torch"][1].values())
    for k in out["torch"][1]:
        if not _zero_in_exact_arithmetic(k):
            _grad_close(out["hip"][1][k].cpu().numpy(), out["torch"][1][k].cpu().numpy(), k, rel=2e-3, floor=4e-6 * gmax)
