"""Shim of torch_geometric.data.Data: a plain attribute bag (x, edge_index, edge_attr)."""


class Data:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)
