"""Shim of torch_geometric.nn: global_mean_pool and BatchNorm (PyG 2.0.4 semantics).

global_mean_pool(x, batch): mean of rows of x grouped by the int64 index `batch`
(size = batch.max()+1).  BatchNorm(c): a module holding `.module = BatchNorm1d(c)`
so checkpoint keys read `...v_bns.{i}.module.weight`.
"""
import torch


def global_mean_pool(x, batch, size=None):
    size = int(batch.max().item()) + 1 if size is None else size
    out = torch.zeros((size, x.shape[1]), dtype=x.dtype, device=x.device)
    out = out.index_add(0, batch, x)
    cnt = torch.zeros((size,), dtype=x.dtype, device=x.device)
    cnt = cnt.index_add(0, batch, torch.ones_like(batch, dtype=x.dtype))
    return out / cnt.clamp(min=1).unsqueeze(-1)


class BatchNorm(torch.nn.Module):
    def __init__(self, in_channels, eps=1e-5, momentum=0.1, affine=True,
                 track_running_stats=True):
        super().__init__()
        self.module = torch.nn.BatchNorm1d(in_channels, eps, momentum, affine,
                                           track_running_stats)

    def forward(self, x):
        return self.module(x)
