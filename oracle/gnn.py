"""numpy restatement of the reference's heuristic network forward.  TEST INFRASTRUCTURE ONLY.

EmbNet.forward tsp/net.py:27-45 (12 residual edge-GNN layers), MLP/ParNet.forward :59-66,74-75,
Net.forward :84-88, Net.reshape :94-102.  `weights` maps state_dict keys (emb_net.v_lins1.0.weight,
emb_net.v_bns.0.module.running_mean, par_net_heu.lins.2.bias, ...) to float32 arrays.
Third-party semantics restated (not vendored in the reference, README.md:20 pins PyG 2.0.4):
global_mean_pool = per-source mean of incoming rows; BatchNorm = torch BatchNorm1d
(eval: running statistics; train: biased batch variance), eps 1e-5.  Pinned against the
reference's own outputs with its shipped checkpoints (fixtures g5_net_*).
"""
import numpy as np

DEPTH = 12
BN_EPS = np.float32(1e-5)


def silu(x):
    return x / (1 + np.exp(-x))


def sigmoid(x):
    return 1 / (1 + np.exp(-x))


def lin(x, W, b):
    return x @ W.T + b


def bn(x, w, prefix, train):
    g, b = w[prefix + ".module.weight"], w[prefix + ".module.bias"]
    if train:
        m, v = x.mean(0), x.var(0)
    else:
        m, v = w[prefix + ".module.running_mean"], w[prefix + ".module.running_var"]
    return (x - m) / np.sqrt(v + BN_EPS) * g + b


def emb_forward(w, x, edge_index, edge_attr, train=False, dtype=np.float64, node_update=True):
    """node_update=False: sop/net.py:43 and smtwtp/net.py:43 have the node update commented out -- the node state stays
    silu(v_lin0(x)) through all layers, only the edge update runs."""
    w = {k: v.astype(dtype) for k, v in w.items()}
    x, e = x.astype(dtype), edge_attr.astype(dtype)
    src, dst = edge_index[0], edge_index[1]
    n = x.shape[0]
    x = silu(lin(x, w["emb_net.v_lin0.weight"], w["emb_net.v_lin0.bias"]))
    e = silu(lin(e, w["emb_net.e_lin0.weight"], w["emb_net.e_lin0.bias"]))
    deg = np.maximum(np.bincount(src, minlength=n), 1).astype(dtype)[:, None]
    for i in range(DEPTH):
        x0, w0 = x, e
        x1 = lin(x0, w[f"emb_net.v_lins1.{i}.weight"], w[f"emb_net.v_lins1.{i}.bias"])
        x2 = lin(x0, w[f"emb_net.v_lins2.{i}.weight"], w[f"emb_net.v_lins2.{i}.bias"])
        x3 = lin(x0, w[f"emb_net.v_lins3.{i}.weight"], w[f"emb_net.v_lins3.{i}.bias"])
        x4 = lin(x0, w[f"emb_net.v_lins4.{i}.weight"], w[f"emb_net.v_lins4.{i}.bias"])
        w1 = lin(w0, w[f"emb_net.e_lins0.{i}.weight"], w[f"emb_net.e_lins0.{i}.bias"])
        w2 = sigmoid(w0)
        agg = np.zeros((n, x.shape[1]), dtype)
        np.add.at(agg, src, w2 * x2[dst])
        agg = agg / deg
        if node_update:
            x = x0 + silu(bn(x1 + agg, w, f"emb_net.v_bns.{i}", train))
        e = w0 + silu(bn(w1 + x3[src] + x4[dst], w, f"emb_net.e_bns.{i}", train))
    return e


def net_forward(w, x, edge_index, edge_attr, train=False, dtype=np.float64, node_update=True):
    e = emb_forward(w, x, edge_index, edge_attr, train, dtype, node_update)
    wd = {k: v.astype(dtype) for k, v in w.items()}
    h = silu(lin(e, wd["par_net_heu.lins.0.weight"], wd["par_net_heu.lins.0.bias"]))
    h = silu(lin(h, wd["par_net_heu.lins.1.weight"], wd["par_net_heu.lins.1.bias"]))
    h = sigmoid(lin(h, wd["par_net_heu.lins.2.weight"], wd["par_net_heu.lins.2.bias"]))
    return h[:, 0]


def reshape(n, edge_index, vec):
    m = np.zeros((n, n), vec.dtype)
    m[edge_index[0], edge_index[1]] = vec
    return m


def weights_from_fixture(g):
    return {k[3:]: v for k, v in g.items() if k.startswith("w__")}
