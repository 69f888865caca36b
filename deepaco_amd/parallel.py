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
    import gc
    # the collector is off inside the region (a pause inside a dozen milliseconds would be measured as device time) -- and NOT run
    # here: a full collection is tens of milliseconds of host work during which the device idles and its clocks fall back, which
    # the first steps of the region then pay for (measured: 47.8 M instead of 51.2 M ant-tours/s with a gc.collect() at this point;
    # callers that want one run it before their warm-up steps)
    was_on = gc.isenabled()
    gc.disable()
    fence()
    t0 = time.perf_counter()
    try:
        fn()
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
    finally:
        if was_on:
            gc.enable()
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


def all_gather_into_(out, x):
    """out [world, *x.shape] <- every rank's x (one collective into a preallocated buffer)."""
    # (the concatenated form [world * x.shape[0], ...] of the output: the one both gloo and RCCL accept)
    flat = (out.shape[0] * x.shape[0],) + tuple(x.shape[1:])
    if _host_staged(x):
        host = torch.empty(flat, dtype=out.dtype)
        dist.all_gather_into_tensor(host, x.cpu().contiguous())
        out.copy_(host.view(out.shape))
    else:
        dist.all_gather_into_tensor(out.view(flat), x.contiguous())


class AntShardedColony:
    """Ant-sharded colony: replicated pheromone, every rank builds A / world ants of the same instances.
    AS, elitist and MMAS as tsp/aco.py:75-118 (cvrp/aco.py:67-130 with directed deposits): best-so-far cost AND tour are
    tracked colony-wide, the elitist deposit is the iteration-best ant's, the MMAS bounds follow the global best.

    sample_fn(tau, ant_lo, n_local, it) -> paths [B, L, A_local] int64
    cost_fn(paths) -> costs [B, A_local]
    exchange="delta":  deposit_fn(zeros_like_tau, paths, costs) -> the deposits of this rank's ants (decay 1, onto zeros);
                       one all-reduce(sum) of [B, n, n] per iteration (AS).  The best ant travels separately: all-gather of the
                       ranks' best costs [B], the winner's tour by an all-reduce of [B, L] int32 (zeros from the others).
                       elitist colonies need NO [B, n, n] exchange: every rank deposits the winner's tour itself
                       (update_fn on one ant).  Pheromone equals the single-GPU colony's to ~1e-6 (sum order), tours differ.
    exchange="tours":  update_fn(tau, paths [B, L, A], costs [B, A], elitist, cmin, cmax) -> tau updated in place by the full
                       evaporate + deposit (+ clamp) of the single-GPU colony; one all-gather of the tours (int16) and costs.
                       With colony-wide ant ids this is the single-GPU colony bit for bit, best tours included.
    The callables are the engine's kernels on a GPU and plain torch on CPU in the gloo tests."""

    def __init__(self, tau, n_ants, decay, rank, world, sample_fn, cost_fn, deposit_fn, exchange="delta",
                 update_fn=None, elitist=False, min_max=False, min=None, problem_size=None, floor=None):
        assert exchange in ("delta", "tours")
        assert exchange == "delta" or update_fn is not None
        assert not (exchange == "delta" and elitist) or update_fn is not None
        self.tau = tau              # the colony OWNS this tensor from here on: step() updates it in place (pass a clone to keep yours)
        self.n_ants, self.decay, self.rank, self.world = n_ants, decay, rank, world
        self.lo, self.hi = shard_range(n_ants, rank, world)
        self.sample_fn, self.cost_fn, self.deposit_fn = sample_fn, cost_fn, deposit_fn
        self.exchange, self.update_fn = exchange, update_fn
        self.elitist, self.min_max, self.floor = elitist, min_max, floor
        self.problem_size = tau.shape[-1] if problem_size is None else problem_size
        if min_max:
            self.min = 0.1 if min is None else min
            assert self.min > 1e-9
            self.max = None
            self.tau.mul_(self.min)                                        # tsp/aco.py:37-40 (tau starts at ones * min)
        self.lowest_cost = torch.full((tau.shape[0],), float("inf"), device=tau.device)
        self.shortest_path = None                                          # [B, L] int64 once a step has run
        self.iteration = 0
        self._buffers = {}          # exchange buffers, allocated once (a [B,n,n] delta or B*n*A tours per iteration otherwise)

    def _buffer(self, tag, shape, dtype, device):
        key = (tag, tuple(shape), dtype)
        buf = self._buffers.get(key)
        if buf is None:
            buf = self._buffers[key] = torch.zeros(tuple(shape), dtype=dtype, device=device)
        return buf

    def _gather_ants(self, x, dim):
        """all-gather along the ant dimension (ranks may own one ant more or less: padded, then trimmed): ONE
        all_gather_into_tensor into a preallocated [world, ...] buffer."""
        if self.world == 1:
            return x
        q = -(-self.n_ants // self.world)
        shape = list(x.shape)
        shape[dim] = q
        pad = self._buffer("pad", shape, x.dtype, x.device)            # (rows past this rank's ants stay zero)
        pad.narrow(dim, 0, x.shape[dim]).copy_(x)
        out = self._buffer("out", [self.world] + shape, x.dtype, x.device)
        if pad.dtype == torch.int16:          # neither gloo nor RCCL has a 16-bit integer type: the same bytes as uint8
            all_gather_into_(out.view(torch.uint8), pad.view(torch.uint8))
        else:
            all_gather_into_(out, pad)
        parts = []
        for r in range(self.world):
            lo, hi = shard_range(self.n_ants, r, self.world)
            parts.append(out[r].narrow(dim, 0, hi - lo))
        return torch.cat(parts, dim=dim)

    def _track(self, best_cost, best_path):
        """tsp/aco.py:78-88 with the colony-wide iteration best: (MMAS max | None)."""
        if self.shortest_path is None or self.shortest_path.shape[1] != best_path.shape[1]:
            self.shortest_path = torch.zeros_like(best_path)
        better = best_cost < self.lowest_cost
        self.shortest_path = torch.where(better.unsqueeze(1), best_path, self.shortest_path)
        self.lowest_cost = torch.where(better, best_cost, self.lowest_cost)
        if not self.min_max:
            return None
        new_max = (1 / self.lowest_cost) * self.problem_size          # the two roundings of daco_track_best
        if self.max is None:                                           # tsp/aco.py:86-87: first best rescales tau
            self.tau.mul_((new_max / self.tau.amax(dim=(1, 2))).view(-1, 1, 1))
        self.max = new_max
        return new_max

    def _iteration_best(self, paths, costs):
        """(cost [B], tour [B, L]) of the colony's best ant of this iteration (first minimum in ant order) on every rank."""
        lc, li = costs.min(dim=1)
        lp = torch.gather(paths, 2, li.view(-1, 1, 1).expand(-1, paths.shape[1], 1)).squeeze(2)
        if self.world == 1:
            return lc, lp
        allc = self._buffer("bestc", (self.world,) + tuple(lc.shape), lc.dtype, lc.device)
        all_gather_into_(allc, lc)
        gc, owner = allc.min(dim=0)                                   # first minimum = lowest rank = lowest ant id
        mine = (owner == self.rank).unsqueeze(1)
        contrib = torch.where(mine, lp, torch.zeros_like(lp)).to(torch.int32)
        all_reduce_(contrib, dist.ReduceOp.SUM)                       # the winner's tour; the other ranks add zeros
        return gc, contrib.to(torch.int64)

    @torch.no_grad()
    def step(self):
        paths = self.sample_fn(self.tau, self.lo, self.hi - self.lo, self.iteration)
        costs = self.cost_fn(paths)
        if self.exchange == "tours":
            narrow = torch.int16 if self.tau.shape[-1] <= 32767 else torch.int32
            all_paths = self._gather_ants(paths.to(narrow), 2).to(torch.int64)       # the one data-path collective
            all_costs = self._gather_ants(costs, 1)
            bc, bi = all_costs.min(dim=1)
            bp = torch.gather(all_paths, 2, bi.view(-1, 1, 1).expand(-1, all_paths.shape[1], 1)).squeeze(2)
            new_max = self._track(bc, bp)
            cmin = None if new_max is None else torch.full_like(new_max, self.min)
            self.update_fn(self.tau, all_paths.contiguous(), all_costs.contiguous(), self.elitist, cmin, new_max)
            self.iteration += 1
            return paths, costs
        bc, bp = self._iteration_best(paths, costs)
        new_max = self._track(bc, bp)
        cmin = None if new_max is None else torch.full_like(new_max, self.min)
        if self.elitist:
            # only the iteration-best ant deposits (tsp/aco.py:103-107): every rank has its tour, nothing else is exchanged
            self.update_fn(self.tau, bp.unsqueeze(2).contiguous(), bc.unsqueeze(1).contiguous(), False, cmin, new_max)
            self.iteration += 1
            return paths, costs
        buf = self._buffer("delta", self.tau.shape, self.tau.dtype, self.tau.device).zero_()
        delta = self.deposit_fn(buf, paths, costs)                # (deposit_fn adds into the buffer it is given and returns it; the
        if delta is not buf:                                      # buffer is zeroed again next step, so nobody may keep it)
            delta = buf.copy_(delta)
        if self.world > 1:
            all_reduce_(delta, dist.ReduceOp.SUM)                 # the one [B, n, n] collective
        self.tau.mul_(self.decay).add_(delta)                     # in place: tau * decay (rounded), then + delta
        if new_max is not None:                                   # tsp/aco.py:116-118
            torch.maximum(self.tau, cmin.view(-1, 1, 1), out=self.tau)
            torch.minimum(self.tau, new_max.view(-1, 1, 1), out=self.tau)
        if self.floor is not None:                                # cvrp/aco.py:130
            self.tau.clamp_(min=self.floor)
        self.iteration += 1
        return paths, costs
