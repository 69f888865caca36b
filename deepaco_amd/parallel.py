"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm).

The rollout path shards two ways:

* instance-sharded (primary): colonies are independent, each rank owns B/N instances and its own
  pheromone; there is NO data-path collective -- only a final gather of the [B] best costs.
* ant-sharded: every rank runs A/N ants of the SAME instances with a replicated pheromone.  Two exchanges:
  - "delta": the ranks' deposits (delta-tau, [B, n, n] f32) are summed with ONE all-reduce and every rank
    applies tau <- decay*tau + delta.  The sum order across ranks differs from the single-GPU ant order,
    so pheromone agrees to ~1e-6 relative, not bitwise.
  - "tours" (exact): the ranks all-gather their tours (as int16: n*A_local*2 bytes per instance instead of
    4*n*n) and costs, and every rank applies the full deposit in ant order -- with the colony-wide ant ids
    (ant_gid_bstride) the N-GPU colony is bit-identical to the single-GPU one.

Everything here is host logic over torch.distributed and is exercised on CPU with the gloo
backend (tests/test_parallel_gloo.py); the kernels are injected as callables.
"""
import time

import torch
import torch.distributed as dist


def _host_staged(t):
    """gloo moves host tensors: a device tensor goes through the host (tests and CPU-only rendezvous); nccl (= RCCL)
    reduces device tensors in place over xGMI."""
    return t.is_cuda and dist.get_backend() != "nccl"


def all_gather_(out, x):
    if _host_staged(x):
        host = [o.cpu() for o in out]
        dist.all_gather(host, x.cpu())
        for o, h in zip(out, host):
            o.copy_(h)
    else:
        dist.all_gather(out, x)


def all_reduce_(x, op=None):
    op = dist.ReduceOp.SUM if op is None else op
    if _host_staged(x):
        h = x.cpu()
        dist.all_reduce(h, op=op)
        x.copy_(h)
    else:
        dist.all_reduce(x, op=op)


def shard_range(total, rank, world):
    """Contiguous, balanced [lo, hi) slice of `total` items for `rank` (first ranks get the extra)."""
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def barrier_max_time(fn, device, distributed):
    """Time fn() bracketed by barrier + device sync on both sides; return the MAX over ranks (s)."""
    def fence():
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        if distributed:
            dist.barrier()
        if device.type == "cuda":
            torch.cuda.synchronize(device)
    fence()
    t0 = time.perf_counter()
    fn()
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    if distributed:
        # gloo reduces host tensors, nccl (= RCCL) device tensors
        tdev = device if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.barrier()
    return dt


def gather_best(lowest_cost, total, rank, world):
    """Instance-sharded epilogue: all ranks' per-instance best costs -> one [total] tensor (all ranks)."""
    if world == 1:
        return lowest_cost
    q = -(-total // world)
    pad = torch.full((q,), float("inf"), dtype=lowest_cost.dtype, device=lowest_cost.device)
    pad[: lowest_cost.numel()] = lowest_cost
    out = [torch.empty_like(pad) for _ in range(world)]
    all_gather_(out, pad)
    parts = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        parts.append(out[r][: hi - lo])
    return torch.cat(parts)


class AntShardedColony:
    """Ant-sharded AS colony: replicated pheromone, local deposits summed by one all-reduce.

    sample_fn(tau, ant_gid0, n_local, it) -> paths [B, n, A_local]
    cost_fn(paths) -> costs [B, A_local]
    deposit_fn(zeros_like_tau, paths, costs) -> delta (deposit with decay = 1 onto zeros, in place)
    The callables are the engine's kernels on a GPU and plain torch on CPU in the gloo tests."""

    def __init__(self, tau, n_ants, decay, rank, world, sample_fn, cost_fn, deposit_fn, exchange="delta",
                 update_fn=None):
        """exchange="tours" needs update_fn(tau, paths [B,n,A], costs [B,A]) -> tau updated in place (the full
        evaporate + deposit of the single-GPU colony) instead of deposit_fn."""
        assert exchange in ("delta", "tours")
        assert exchange == "delta" or update_fn is not None
        self.tau = tau              # the colony OWNS this tensor from here on: step() updates it in place (pass a clone to keep yours)
        self.n_ants, self.decay, self.rank, self.world = n_ants, decay, rank, world
        self.lo, self.hi = shard_range(n_ants, rank, world)
        self.sample_fn, self.cost_fn, self.deposit_fn = sample_fn, cost_fn, deposit_fn
        self.exchange, self.update_fn = exchange, update_fn
        self.lowest_cost = torch.full((tau.shape[0],), float("inf"), device=tau.device)
        self.iteration = 0
        self._buffers = {}          # exchange buffers, allocated once (a [B,n,n] delta or B*n*A tours per iteration otherwise)

    def _buffer(self, tag, shape, dtype, device):
        key = (tag, tuple(shape), dtype)
        buf = self._buffers.get(key)
        if buf is None:
            buf = self._buffers[key] = torch.zeros(tuple(shape), dtype=dtype, device=device)
        return buf

    def _gather_ants(self, x, dim):
        """all-gather along the ant dimension (ranks may own one ant more or less: padded, then trimmed)."""
        if self.world == 1:
            return x
        q = -(-self.n_ants // self.world)
        shape = list(x.shape)
        shape[dim] = q
        pad = self._buffer("pad", shape, x.dtype, x.device)            # (rows past this rank's ants stay zero)
        pad.narrow(dim, 0, x.shape[dim]).copy_(x)
        out = [self._buffer(("out", r), shape, x.dtype, x.device) for r in range(self.world)]
        if pad.dtype == torch.int16:          # neither gloo nor RCCL moves int16: ship the same bytes as uint8
            all_gather_([o.view(torch.uint8) for o in out], pad.view(torch.uint8))
        else:
            all_gather_(out, pad)
        parts = []
        for r in range(self.world):
            lo, hi = shard_range(self.n_ants, r, self.world)
            parts.append(out[r].narrow(dim, 0, hi - lo))
        return torch.cat(parts, dim=dim)

    @torch.no_grad()
    def step(self):
        paths = self.sample_fn(self.tau, self.lo, self.hi - self.lo, self.iteration)
        costs = self.cost_fn(paths)
        if self.exchange == "tours":
            narrow = torch.int16 if paths.shape[1] <= 32767 else torch.int32
            all_paths = self._gather_ants(paths.to(narrow), 2).to(torch.int64)       # the one data-path collective
            all_costs = self._gather_ants(costs, 1)
            self.update_fn(self.tau, all_paths, all_costs)
            self.lowest_cost = torch.minimum(self.lowest_cost, all_costs.min(dim=1).values)
            self.iteration += 1
            return paths, costs
        buf = self._buffer("delta", self.tau.shape, self.tau.dtype, self.tau.device).zero_()
        delta = self.deposit_fn(buf, paths, costs)                # (deposit_fn adds into the buffer it is given and returns it; the
        if delta is not buf:                                      # buffer is zeroed again next step, so nobody may keep it)
            delta = buf.copy_(delta)
        best = costs.min(dim=1).values
        if self.world > 1:
            all_reduce_(delta, dist.ReduceOp.SUM)                 # the one data-path collective
            all_reduce_(best, dist.ReduceOp.MIN)
        self.tau.mul_(self.decay).add_(delta)                     # in place: tau * decay (rounded), then + delta, as before
        self.lowest_cost = torch.minimum(self.lowest_cost, best)
        self.iteration += 1
        return paths, costs
