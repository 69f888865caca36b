#!/usr/bin/env python3
"""g5b: Net.forward of three SIBLING directories with the reference's own checkpoints (this container only).

VERDICT r3 "missing 3": the sibling networks were only compared with the repo's own torch module tree.  Here the
reference's net.py is imported from /root/reference/{sop,op,mkp} (torch_geometric replaced by the attribute-bag shim),
loaded with the checkpoint it ships (pretrained/sop/sop20.pt -- the variant whose node update is commented out,
sop/net.py:43 --, pretrained/op/op100.pt, pretrained/mkp/mkp300.pt), and run on a small instance built by the
reference's utils.py: eval-mode heuristic vector, the edge embedding, the train-mode vector, the reshaped matrix.
Stored as numbers (graph, outputs, float weights), like g5.

Run:  python tests/golden/gen_g5b_sibling_nets.py   (writes tests/golden/g5b_net_<dir>_<ckpt>.npz)
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DEEPACO_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "shims"))


def load_ref(subdir, name):
    alias = f"ref_{subdir}_{name}"
    spec = importlib.util.spec_from_file_location(alias, os.path.join(REF, subdir, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[alias] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    for sub, ck in (("sop", "sop20"), ("op", "op100"), ("mkp", "mkp300")):
        net_mod, utils = load_ref(sub, "net"), load_ref(sub, "utils")
        sd = torch.load(os.path.join(REF, "pretrained", sub, ck + ".pt"), map_location="cpu")
        model = net_mod.Net()
        model.load_state_dict(sd)
        torch.manual_seed(11)
        np.random.seed(11)
        if sub == "sop":
            dist, adj, mask = utils.training_instance_gen(20, "cpu")
            pyg = utils.gen_pyg_data(dist, adj, "cpu")
        elif sub == "op":
            pyg, dist, prizes = utils.gen_pyg_data(torch.rand(40, 2), 8)
        else:
            prize, wm = utils.gen_instance(16, 5, "cpu")
            pyg = utils.gen_pyg_data(prize, wm)
        model.eval()
        with torch.no_grad():
            heu_eval = model(pyg)
            emb_eval = model.emb_net(pyg.x, pyg.edge_index, pyg.edge_attr)
            mat = net_mod.Net.reshape(pyg, heu_eval)
        model.train()
        with torch.no_grad():
            heu_train = model(pyg)
        weights = {"w__" + k: v.float().numpy() for k, v in sd.items() if v.numel() > 0 and v.dtype.is_floating_point}
        name = os.path.join(HERE, f"g5b_net_{sub}_{ck}.npz")
        np.savez_compressed(name, x=pyg.x.numpy(), edge_index=pyg.edge_index.numpy(), edge_attr=pyg.edge_attr.numpy(),
                            heu_eval=heu_eval.numpy(), emb_eval=emb_eval.numpy(), heu_train=heu_train.numpy(),
                            heu_mat=mat.numpy(), **weights)
        print(name, "nodes", pyg.x.shape, "edges", pyg.edge_index.shape[1], "heu range", float(heu_eval.min()), float(heu_eval.max()))


if __name__ == "__main__":
    main()
