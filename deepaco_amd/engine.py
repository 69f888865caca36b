"""Batched functional layer over the C ABI: B instances x A ants per call.

Every function takes torch tensors that live on a HIP device, enqueues the kernels on the
current torch stream and returns torch tensors; nothing here synchronises with the host.
Layouts follow the reference with a leading batch dimension:
paths [B, n, A] int64, log_probs [B, n-1, A] f32, costs [B, A] f32.
"""
import os

import ctypes as C

import torch

from . import _lib
from ._lib import RACE_NOISE, RACE_PHILOX, SCAN, SCAN_WAVE  # noqa: F401

# "scan": daco_tsp_sample / daco_cvrp_sample pack sixteen ants per wavefront for n <= 128, eight for n <= 256 and two for
# 256 < n <= 512 (TSP: 1024; measured crossovers, tools/sweep_layouts.py); "scan_wave" keeps the one-ant-per-wavefront draw for every n (what the step-wise
# service and the fused siblings use)
MODES = {"race_noise": RACE_NOISE, "race": RACE_PHILOX, "scan": SCAN, "scan_wave": SCAN_WAVE}

_workspaces = {}


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.DacoError(
                "deepaco_amd kernels run on a HIP device only (got a CPU tensor); there is no CPU fallback")


def stage_to_hip(t, like=None):
    """The reference's test scripts build the colony with device='cpu' and host tensors (tsp_nls/test.py:22-29,
    cvrp/test.py:20-27).  There is no CPU compute path here: host tensors handed to a colony are copied to the HIP
    device once (differentiably, so a heuristic keeps its autograd history) and everything runs -- and is returned --
    there.  `like`: a tensor whose device to use; default: the current HIP device."""
    if t is None or not torch.is_tensor(t) or t.is_cuda:
        return t
    if not torch.cuda.is_available():
        raise _lib.DacoError("deepaco_amd has no CPU path: no HIP device is visible for the host tensors passed in")
    dev = like.device if (like is not None and like.is_cuda) else torch.device("cuda", torch.cuda.current_device())
    return t.to(dev)


def _workspace(device, nbytes, tag):
    """Per (device, stream, tag) scratch buffer, grown on demand (owned by the caller side of the ABI)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, tag)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL_CTX = _NullCtx()


def _on(device):
    """`with torch.cuda.device(device)` only when that device is not the current one already: the context manager costs ~10 us per
    use, and at the reference's small sizes (TSP-20 / CVRP-100 with 20 ants: tools/host_overhead_small.py) an ACO iteration is three
    library calls whose host time IS the iteration time."""
    return _NULL_CTX if torch.cuda.current_device() == (device.index if device.index is not None else torch.cuda.current_device()) \
        else torch.cuda.device(device)


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()


def _bstride(t, n):
    """(tensor, element stride between instances) for a [n,n] (shared) or [B,n,n] matrix."""
    t = _f32c(t)
    return (t, 0) if t.dim() == 2 else (t, n * n)


def tsp_sample(tau, eta, n_ants, alpha=1.0, beta=1.0, mode="scan", norm_passes=1, start=None,
               fixed_start=-1, noise=None, seed=0, it=0, ant_gid0=0, require_prob=False, batch=None,
               events=None, dist=None, want_nbr=False, iter_dev=None, ant_gid_bstride=0, flags=None):
    """ACO.gen_path for a batch (tsp/aco.py:134-177, tsp_nls/aco.py:184-220).

    tau, eta: [B,n,n] or [n,n] (shared).  Returns (paths, log_probs|None, rowsum|None, flags).
    events: optional (begin, end) torch.cuda.Event pair (already recorded once, so the handles
    exist) re-recorded around the tour-construction kernel only.
    dist: if given, tour costs are fused into the kernel and returned; want_nbr: also return the
    neighbour table the pheromone update consumes.  With either, the return value is
    (paths, log_probs, rowsum, flags, costs|None, nbr|None)."""
    _require_gpu(tau, eta, start, noise)
    n = tau.shape[-1]
    B = batch or (tau.shape[0] if tau.dim() == 3 else (eta.shape[0] if eta.dim() == 3 else 1))
    dev = tau.device
    tau, tbs = _bstride(tau, n)
    eta, ebs = _bstride(eta, n)
    m = MODES[mode] if isinstance(mode, str) else int(mode)
    L = _lib.lib()
    with _on(dev):
        paths = torch.empty((B, n, n_ants), dtype=torch.int64, device=dev)
        logp = torch.empty((B, n - 1, n_ants), dtype=torch.float32, device=dev) if require_prob else None
        rowsum = torch.empty((B, n - 1, n_ants), dtype=torch.float32, device=dev) if require_prob else None
        if flags is None:                                    # (a caller's own flag words are OR-ed into: no fill launch per call)
            flags = torch.zeros((B,), dtype=torch.int32, device=dev)
        if start is not None:
            start = start.to(torch.int64).contiguous().view(B, n_ants)
        if noise is not None:      # race_noise: the reference's q tensors; scan modes: injected uniforms [B, n-1, A]
            noise = _f32c(noise).view(B, n - 1, n_ants, n) if m == RACE_NOISE else _f32c(noise).view(B, n - 1, n_ants)
        costs = nbr = None
        dbs = 0
        if dist is not None:
            _require_gpu(dist)
            dist, dbs = _bstride(dist, n)
            costs = torch.empty((B, n_ants), dtype=torch.float32, device=dev)
        if want_nbr:
            nbr = torch.empty((B, n, n_ants), dtype=torch.int32, device=dev)
        nbytes = L.daco_tsp_sample_workspace_bytes(B, n, m)
        ws = _workspace(dev, nbytes, "sample")
        rc = L.daco_tsp_sample(_stream(dev), B, n, n_ants, tau.data_ptr(), tbs, eta.data_ptr(), ebs,
                               float(alpha), float(beta), m, int(norm_passes),
                               start.data_ptr() if start is not None else None, int(fixed_start),
                               noise.data_ptr() if noise is not None else None,
                               int(seed) & (2 ** 64 - 1), int(it), iter_dev.data_ptr() if iter_dev is not None else None,
                               int(ant_gid0) & 0xFFFFFFFF, int(ant_gid_bstride),
                               paths.data_ptr(), logp.data_ptr() if require_prob else None,
                               rowsum.data_ptr() if require_prob else None, flags.data_ptr(),
                               dist.data_ptr() if dist is not None else None, dbs,
                               costs.data_ptr() if costs is not None else None,
                               nbr.data_ptr() if nbr is not None else None,
                               ws.data_ptr(), ws.numel(),
                               events[0].cuda_event if events else None,
                               events[1].cuda_event if events else None)
    _lib.check(rc, "daco_tsp_sample")
    if dist is not None or want_nbr:
        return paths, logp, rowsum, flags, costs, nbr
    return paths, logp, rowsum, flags


def sparse_head(weights, k, top=None):
    """Head table of scan_sparse for `weights` [B,n,n] or [n,n] (the colony passes its heuristic): per row the k (<= 127)
    largest entries, ids ascending (ties at the k-th value: the smaller id) -- [B,n,S] int16 holding uint16 ids with S = 64
    slots for k <= 63, else 128; the unused slots 0, the last slot = k (include/deepaco_hip.h daco_tsp_sample_sparse;
    oracle.sparse_head_ids is the same rule).
    top: the rows' largest values in descending order, at least k of them ([.., m >= k], e.g. what auto_head_k computed for the
    same matrix): the one torch.topk this function needs is then skipped."""
    _require_gpu(weights)
    w = weights if weights.dim() == 3 else weights.unsqueeze(0)
    B, n, _ = w.shape
    assert 1 <= k <= 127 and k <= n
    slots = 64 if k <= 63 else 128
    # the k largest by (value descending, id ascending), without sorting the rows: everything above the k-th value,
    # and of the entries equal to it the smallest ids.  The chosen ids ascending = the chosen columns in column order: their
    # running count is the slot (the others go to a spare slot that is cut off).
    v = w.detach().to(torch.float32)
    if top is not None and top.shape[-1] >= k:
        kth = (top if top.dim() == 3 else top.unsqueeze(0))[:, :, k - 1:k].to(torch.float32)     # ([1,n,1] broadcasts over B)
    else:
        kth = torch.topk(v, k, dim=2).values[:, :, k - 1:k]
    greater = v > kth
    eq = v == kth
    room = k - greater.sum(dim=2, keepdim=True)
    chosen = greater | (eq & (torch.cumsum(eq, dim=2, dtype=torch.int32) <= room))
    slot = torch.where(chosen, torch.cumsum(chosen, dim=2, dtype=torch.int32) - 1, slots).to(torch.int64)
    ids = torch.zeros((B, n, slots + 1), dtype=torch.int16, device=w.device)
    ids.scatter_(2, slot, torch.arange(n, device=w.device, dtype=torch.int16).view(1, 1, n).expand(B, n, n))
    ids = ids[:, :, :slots].contiguous()                 # (bit pattern of uint16: ids < 32768 here, n <= 1024)
    ids[:, :, slots - 1] = k
    return ids


SPARSE_MIN_N, SPARSE_MAX_N = 129, 1024        # sizes daco_tsp_sample_sparse / _race_head cover


def auto_head_k(heuristic, mass=0.98, want_top=False):
    """Head size for sampler='auto' on a heuristic nobody sparsified by hand: 63 or 127 if that many largest entries hold at
    least `mass` of (nearly) every row (the learned heuristic is k-sparse by construction: tsp/net.py:94-102 scatters k values per row
    into zeros, + 1e-10), else None.  One reduction and one host read per heuristic object.
    mass = 0.98: a step that has to leave the head re-reads the whole row for its ants (each such step stops four
    ants for a row walk); with a fifth of the mass in the tail (plain 1/d at n = 200: 0.85 in the best 127) the head rows lose.
    want_top: also return the rows' 127 largest values (descending), which sparse_head takes instead of its own torch.topk."""
    h = heuristic.detach()
    n = h.shape[-1]
    if not (SPARSE_MIN_N <= n <= SPARSE_MAX_N):
        return (None, None) if want_top else None
    h = h.to(torch.float32)
    top = torch.topk(h, min(127, n - 1), dim=-1).values
    tot = h.sum(dim=-1)
    # "every row" up to one row in twenty: a network output can leave single rows flat (all live entries ~1e-13 against the
    # 1e-10 floor); those rows cost a dense step when an ant stands on them, the colony still gains on the others
    ok63 = ((top[..., :63].sum(dim=-1) / tot) >= mass).float().mean()
    ok127 = ((top.sum(dim=-1) / tot) >= mass).float().mean()
    # Round 6: a head that leaves a slot free AND fits a CU's LDS next to four ants (n x ceil((k + 1) / 4) lanes x 24 bytes: k <= 51 at
    # n = 500, <= 62 below n = 410) lets a launch of few ants -- one instance, the reference's own call pattern, tsp/test.ipynb:66-68 --
    # keep its head rows in LDS; taken when it holds practically ALL of the mass (1 - 1e-4: the network's output has
    # exactly the graph's k live entries per row, so k = 50 -> a head of 51), since whatever it left out would be walked as tail
    lanes = min(16, (160 * 1024 - 128 - 4 * 528 - 4 * 514 * 2 - 32) // (n * 24))
    k_lds = min(62, 4 * lanes - 1, top.shape[-1])
    ok_lds = ((top[..., :k_lds].sum(dim=-1) / tot) >= 1.0 - 1e-4).float().mean() if k_lds >= 8 else torch.zeros((), device=h.device)
    ok63, ok127, ok_lds = (float(x) for x in torch.stack((ok63, ok127, ok_lds)).tolist())          # (one host read)
    k = k_lds if ok_lds >= 0.95 else (63 if ok63 >= 0.95 else (127 if ok127 >= 0.95 else None))
    return (k, top if k else None) if want_top else k


_warned_sparse_range = False


def resolve_sampler(sampler, n, head_k, heuristic, cache):
    """('scan' | 'scan_wave' | 'race' | 'scan_sparse', head size) for a colony of n nodes.
    'auto': head / tail rows (scan_sparse: the same categorical as 'scan', 384 / 768 bytes per step instead of a row) whenever
    they apply -- after sparsify(k), or when auto_head_k finds the heuristic concentrated -- and 129 <= n <= 1024; else the dense
    scan.  An explicit 'scan_sparse' outside that range falls back to 'scan' with one warning.
    cache: a dict of the colony (the concentration test is repeated only when the heuristic object changes)."""
    global _warned_sparse_range
    in_range = SPARSE_MIN_N <= n <= SPARSE_MAX_N
    if sampler == "scan_sparse" and not in_range:
        if not _warned_sparse_range:
            _warned_sparse_range = True
            import warnings
            warnings.warn(f"sampler='scan_sparse' covers {SPARSE_MIN_N} <= n <= {SPARSE_MAX_N}; n = {n} runs the dense scan "
                          f"(the same distribution)", RuntimeWarning, stacklevel=3)
        return "scan", None
    if sampler != "auto":
        return sampler, head_k
    if not in_range:
        return "scan", None
    if head_k is not None:
        return "scan_sparse", head_k
    hit = cache.get("auto_head")
    if hit is None or hit[0] is not heuristic:
        k, top = auto_head_k(heuristic, want_top=True)
        hit = cache["auto_head"] = (heuristic, k)
        cache["auto_top"] = (heuristic, top)            # (taken -- and dropped -- by the colony's next head table)
    return ("scan_sparse", hit[1]) if hit[1] else ("scan", None)


def take_auto_top(cache, heuristic):
    """The sorted top values resolve_sampler left in a colony's cache for this heuristic object (None if there are none); they
    are handed over once: 16 MB per 64 x 500 rows that nothing needs after the head table is built."""
    hit = cache.pop("auto_top", None) if cache else None
    return hit[1] if hit is not None and hit[0] is heuristic else None


def sparse_workspace(device, B, n, n_ants, unit_exponents=True):
    """A scratch tensor of the head-row samplers that a colony KEEPS (daco_tsp_sparse_workspace_bytes: the iteration's head rows
    and, for n > 512, the tours as they are built): pheromone_update_(heads=...) writes the next iteration's head rows into it and
    tsp_sample_sparse(heads_ready=True) reads them, so it must not be the per-stream scratch other colonies share."""
    L = _lib.lib()
    nbytes = (L.daco_tsp_sparse_workspace_bytes if unit_exponents else L.daco_tsp_sparse_workspace_bytes_general)(B, n, n_ants)
    if nbytes == 0:
        raise ValueError(f"scan_sparse serves 129 <= n <= 1024 (n = {n})")
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def sparse_tours16(workspace, B, n, n_ants):
    """The tours of the last tsp_sample_sparse call on `workspace` in their compact form: an int16-typed view [B, A, ld] of the
    workspace (ld = 512 for n <= 512, else 1024; entry t of ant a = the node visited at step t, entries >= n undefined; node ids are
    below 32768, so the sign bit is never set).  Written by calls with n > 512 and by calls with want_paths=False
    (include/deepaco_hip.h daco_tsp_sparse_tours_offset)."""
    off = int(_lib.lib().daco_tsp_sparse_tours_offset(B, n, n_ants))
    ld = 512 if n <= 512 else 1024
    return workspace[off:off + B * n_ants * ld * 2].view(torch.int16).view(B, n_ants, ld)


def tsp_sample_sparse(tau, eta, n_ants, head, alpha=1.0, beta=1.0, start=None, fixed_start=-1, seed=0, it=0, ant_gid0=0,
                      batch=None, events=None, dist=None, want_nbr=False, iter_dev=None, ant_gid_bstride=0, want_stats=False,
                      want_paths=True, race=False, workspace=None, heads_ready=False, head_live_max=0, nbr_grouped=False, flags=None):
    """ACO.gen_path on head / tail rows (sampler "scan_sparse", include/deepaco_hip.h daco_tsp_sample_sparse): the
    distribution of tsp_sample(mode="scan"), 384 / 768 bytes per step instead of a row while the head has a live candidate.
    head: sparse_head(heuristic, k).  Returns (paths, flags, costs|None, nbr|None[, stats]).
    race=True: daco_tsp_sample_race_head -- the exponential race of mode="race" on the head rows, with the tours of the dense
    race kernel (same seed), one variate per head slot and step instead of n.
    workspace: a sparse_workspace() tensor the caller keeps (default: the per-stream scratch); heads_ready=True: it already holds
    this iteration's head rows (pheromone_update_(heads=...) wrote them for these very tensors) and the pass over tau is skipped
    (include/deepaco_hip.h daco_tsp_sample_heads) -- the same tours.
    head_live_max: the k the head table was built with (sparse_head(.., k)), or 0: lets a launch of few ants (one instance, a few
    hundred ants) keep the head rows in LDS -- the same tours, about half the time at TSP-500 x 512 ants x 1 instance.
    nbr_grouped: the update's table as [B, ceil(A/8), n, 8] (written in 1 KB pieces; pheromone_update_(heads={..., "nbr_grouped": True})
    takes it) instead of [B, n, A]; the returned tensor keeps the shape [B, n, A] either way (same bytes, opaque to the caller)."""
    _require_gpu(tau, eta, start, head, workspace)
    assert not heads_ready or workspace is not None
    n = tau.shape[-1]
    B = batch or (tau.shape[0] if tau.dim() == 3 else (eta.shape[0] if eta.dim() == 3 else 1))
    dev = tau.device
    tau, tbs = _bstride(tau, n)
    eta, ebs = _bstride(eta, n)
    assert head.dtype == torch.int16 and head.is_contiguous() and tuple(head.shape) in ((B, n, 64), (B, n, 128))
    L = _lib.lib()
    with _on(dev):
        paths = torch.empty((B, n, n_ants), dtype=torch.int64, device=dev) if want_paths else None
        if flags is None:                                    # (a caller that keeps its flag words -- they are OR-ed into -- saves a fill launch per call)
            flags = torch.zeros((B,), dtype=torch.int32, device=dev)
        if start is not None:
            start = start.to(torch.int64).contiguous().view(B, n_ants)
        costs = nbr = None
        dbs = 0
        if dist is not None:
            _require_gpu(dist)
            dist, dbs = _bstride(dist, n)
            costs = torch.empty((B, n_ants), dtype=torch.float32, device=dev)
        if want_nbr:
            assert not nbr_grouped or n_ants % 8 == 0, "the grouped table wants a multiple of eight ants"
            nbr = torch.empty((B, n, n_ants), dtype=torch.int32, device=dev)
        stats = torch.zeros(3, dtype=torch.int64, device=dev) if want_stats else None
        unit = float(alpha) == 1.0 and float(beta) == 1.0       # (other exponents: tau^alpha, eta^beta are formed in the workspace first)
        nbytes = (L.daco_tsp_sparse_workspace_bytes if unit else L.daco_tsp_sparse_workspace_bytes_general)(B, n, n_ants)
        if nbytes == 0:
            raise ValueError(f"scan_sparse serves 129 <= n <= 1024 (n = {n})")
        ws = workspace if workspace is not None else _workspace(dev, nbytes, "sample_sparse")
        assert ws.numel() >= nbytes
        rc = L.daco_tsp_sample_heads(_stream(dev), int(bool(race)), int(bool(heads_ready)), int(head_live_max), int(bool(nbr_grouped)), B, n, n_ants, tau.data_ptr(), tbs, eta.data_ptr(), ebs, float(alpha),
                                      float(beta), head.data_ptr(), int(head.shape[2]), start.data_ptr() if start is not None else None,
                                      int(fixed_start), int(seed) & (2 ** 64 - 1), int(it),
                                      iter_dev.data_ptr() if iter_dev is not None else None, int(ant_gid0) & 0xFFFFFFFF,
                                      int(ant_gid_bstride), paths.data_ptr() if paths is not None else None, flags.data_ptr(),
                                      dist.data_ptr() if dist is not None else None, dbs,
                                      costs.data_ptr() if costs is not None else None,
                                      nbr.data_ptr() if nbr is not None else None,
                                      stats.data_ptr() if stats is not None else None, ws.data_ptr(), ws.numel(),
                                      events[0].cuda_event if events else None, events[1].cuda_event if events else None)
    _lib.check(rc, "daco_tsp_sample_heads")
    out = (paths, flags, costs, nbr)
    return out + (stats,) if want_stats else out


def cvrp_sample(tau, eta, demand, capacity, n_ants, alpha=1.0, beta=1.0, mode="scan", noise=None, seed=0,
                it=0, ant_gid0=0, require_prob=False, Lmax=None, batch=None, dist=None, want_table=False,
                iter_dev=None, events=None, ant_gid_bstride=0, flags=None):
    """CVRP ACO.gen_path for a batch (cvrp/aco.py:138-205).  tau, eta [B,n,n] or [n,n]; demand [B,n]
    or [n] (demand[0] = 0).  Returns (paths [B,Lmax,A], log_probs|None, rowsum|None, lens [B,A], flags [B]);
    the reference's result is paths[:, :lens.max()].
    dist: if given, route costs are fused into the kernel; want_table: also return the successor table
    the directed pheromone update consumes.  With either, (..., costs|None, table|None) is appended.
    events: as in tsp_sample (a pair of recorded torch.cuda.Event re-recorded around the construction kernel).
    A float64 `demand` (cvrp_nls/ keeps its instance data in double) selects the float64 load bookkeeping
    (cvrp_nls/aco.py:254-272: used + demand, demand > capacity - used in double), see include/deepaco_hip.h."""
    _require_gpu(tau, eta, demand, noise)
    n = tau.shape[-1]
    B = batch or (tau.shape[0] if tau.dim() == 3 else (eta.shape[0] if eta.dim() == 3 else 1))
    dev = tau.device
    tau, tbs = _bstride(tau, n)
    eta, ebs = _bstride(eta, n)
    demand64 = None
    if demand.dtype == torch.float64:
        demand64 = demand.detach().contiguous()
        if demand64.dim() == 1:
            demand64 = demand64.unsqueeze(0).expand(B, n).contiguous()
    demand = _f32c(demand)
    if demand.dim() == 1:
        demand = demand.unsqueeze(0).expand(B, n).contiguous()
    m = MODES[mode] if isinstance(mode, str) else int(mode)
    Lmax = Lmax or 2 * n + 1
    L = _lib.lib()
    with _on(dev):
        paths = torch.empty((B, Lmax, n_ants), dtype=torch.int64, device=dev)
        logp = torch.empty((B, Lmax - 1, n_ants), dtype=torch.float32, device=dev) if require_prob else None
        rowsum = torch.ones((B, Lmax - 1, n_ants), dtype=torch.float32, device=dev) if require_prob else None
        lens = torch.empty((B, n_ants), dtype=torch.int32, device=dev)
        if flags is None:                                    # (a caller that keeps its flag words -- they are OR-ed into -- saves a fill launch per call)
            flags = torch.zeros((B,), dtype=torch.int32, device=dev)
        steps = 0
        if noise is not None:
            noise = _f32c(noise)
            steps = noise.shape[-3]
            noise = noise.view(B, steps, n_ants, n)
        costs = table = None
        dbs = 0
        if dist is not None:
            _require_gpu(dist)
            dist, dbs = _bstride(dist, n)
            costs = torch.empty((B, n_ants), dtype=torch.float32, device=dev)
        if want_table:
            table = torch.empty((L.daco_directed_table_bytes(B, n, n_ants),), dtype=torch.uint8, device=dev)
        nbytes = L.daco_tsp_sample_workspace_bytes(B, n, m)
        ws = _workspace(dev, nbytes, "sample")
        rc = L.daco_cvrp_sample(_stream(dev), B, n, n_ants, tau.data_ptr(), tbs, eta.data_ptr(), ebs, float(alpha),
                                float(beta), demand.data_ptr(), float(capacity), m,
                                noise.data_ptr() if noise is not None else None, steps,
                                int(seed) & (2 ** 64 - 1), int(it), iter_dev.data_ptr() if iter_dev is not None else None,
                                int(ant_gid0) & 0xFFFFFFFF, int(ant_gid_bstride), Lmax,
                                paths.data_ptr(), logp.data_ptr() if require_prob else None,
                                rowsum.data_ptr() if require_prob else None, lens.data_ptr(),
                                flags.data_ptr(), dist.data_ptr() if dist is not None else None, dbs,
                                costs.data_ptr() if costs is not None else None,
                                table.data_ptr() if table is not None else None, ws.data_ptr(), ws.numel(),
                                demand64.data_ptr() if demand64 is not None else None, float(capacity),
                                events[0].cuda_event if events else None, events[1].cuda_event if events else None)
    _lib.check(rc, "daco_cvrp_sample")
    if dist is not None or want_table:
        return paths, logp, rowsum, lens, flags, costs, table
    return paths, logp, rowsum, lens, flags


def sample_backward(tau, eta, alpha, beta, paths, rowsum, grad_logp, lens=None, demand=None, capacity=0.0):
    """Gradient of sum(grad_logp * log_probs) w.r.t. eta -> [B,n,n] (autograd through
    Categorical.log_prob in tsp/aco.py:174-176 / cvrp/aco.py:171-173).  CVRP: pass lens, demand, capacity."""
    _require_gpu(tau, eta, paths, rowsum, grad_logp)
    n = tau.shape[-1]
    B, rows, A = paths.shape
    tau, tbs = _bstride(tau, n)
    eta, ebs = _bstride(eta, n)
    paths = paths.contiguous()
    rowsum, grad_logp = _f32c(rowsum), _f32c(grad_logp)
    dev = paths.device
    demand64 = None
    if demand is not None:
        if demand.dtype == torch.float64:                  # replay the capacity rule in double, as the sampler applied it
            demand64 = demand.detach().contiguous()
            if demand64.dim() == 1:
                demand64 = demand64.unsqueeze(0).expand(B, n).contiguous()
        demand = _f32c(demand)
        if demand.dim() == 1:
            demand = demand.unsqueeze(0).expand(B, n).contiguous()
        lens = lens.contiguous()
    with _on(dev):
        grad = torch.zeros((B, n, n), dtype=torch.float32, device=dev)
        rc = _lib.lib().daco_sample_backward(_stream(dev), B, n, A, rows, tau.data_ptr(), tbs, eta.data_ptr(), ebs,
                                             float(alpha), float(beta), paths.data_ptr(), rowsum.data_ptr(),
                                             grad_logp.data_ptr(), lens.data_ptr() if demand is not None else None,
                                             demand.data_ptr() if demand is not None else None, float(capacity),
                                             grad.data_ptr(), demand64.data_ptr() if demand64 is not None else None,
                                             float(capacity))
    _lib.check(rc, "daco_sample_backward")
    return grad


SIB_KINDS = {"sop": 3, "pctsp": 4, "op": 5, "mkp": 6}


def sibling_sample(kind, tau, eta, n_ants, alpha=1.0, beta=1.0, aux_vec=None, aux_mat=None, scalar0=0.0,
                   item_weights=None, mode="scan", start=None, noise=None, seed=0, it=0, ant_gid0=0,
                   require_prob=False, Lmax=None):
    """Fused solution construction for sop / pctsp / op / mkp, one instance batch B = leading dim of tau
    (or 1).  See include/deepaco_hip.h daco_sibling_sample for the meaning of aux_vec / aux_mat / scalar0.
    Returns (paths [B,rows,A], log_probs|None, rowsum|None, lens [B,A]|None, flags [B])."""
    _require_gpu(tau, eta, aux_vec, aux_mat, item_weights, start, noise)
    n = tau.shape[-1]
    B = tau.shape[0] if tau.dim() == 3 else 1
    dev = tau.device
    tau, tbs = _bstride(tau, n)
    eta, ebs = _bstride(eta, n)
    k = SIB_KINDS[kind]
    varlen = kind != "sop"
    rows = (Lmax or 2 * n + 1) if varlen else n
    m = MODES[mode] if isinstance(mode, str) else int(mode)
    if aux_vec is not None:
        aux_vec = _f32c(aux_vec).reshape(-1, n)
        if aux_vec.shape[0] != B:
            aux_vec = aux_vec.expand(B, n).contiguous()
    abs_ = 0
    if aux_mat is not None:
        aux_mat, abs_ = _bstride(aux_mat, n)
    mdim = 0
    if item_weights is not None:
        item_weights = _f32c(item_weights)
        mdim = item_weights.shape[-1]
        if item_weights.dim() == 2:
            item_weights = item_weights.unsqueeze(0).expand(B, n, mdim).contiguous()
    L = _lib.lib()
    with _on(dev):
        paths = torch.empty((B, rows, n_ants), dtype=torch.int64, device=dev)
        logp = torch.empty((B, rows - 1, n_ants), dtype=torch.float32, device=dev) if require_prob else None
        rowsum = torch.ones((B, rows - 1, n_ants), dtype=torch.float32, device=dev) if require_prob else None
        lens = torch.empty((B, n_ants), dtype=torch.int32, device=dev) if varlen else None
        flags = torch.zeros((B,), dtype=torch.int32, device=dev)
        steps = 0
        if noise is not None:
            noise = _f32c(noise)
            steps = noise.shape[-3]
            noise = noise.view(B, steps, n_ants, n)
        if start is not None:
            start = start.to(torch.int64).contiguous().view(B, n_ants)
        nbytes = L.daco_sibling_workspace_bytes(B, n, m)
        ws = _workspace(dev, nbytes, "sibling")
        rc = L.daco_sibling_sample(_stream(dev), k, B, n, n_ants, tau.data_ptr(), tbs, eta.data_ptr(), ebs, float(alpha),
                                   float(beta), aux_vec.data_ptr() if aux_vec is not None else None,
                                   aux_mat.data_ptr() if aux_mat is not None else None, abs_, float(scalar0),
                                   item_weights.data_ptr() if item_weights is not None else None, mdim, m,
                                   start.data_ptr() if start is not None else None,
                                   noise.data_ptr() if noise is not None else None, steps,
                                   int(seed) & (2 ** 64 - 1), int(it), int(ant_gid0) & 0xFFFFFFFF, rows,
                                   paths.data_ptr(), logp.data_ptr() if require_prob else None,
                                   rowsum.data_ptr() if require_prob else None,
                                   lens.data_ptr() if lens is not None else None, flags.data_ptr(), ws.data_ptr(),
                                   ws.numel())
    _lib.check(rc, "daco_sibling_sample")
    return paths, logp, rowsum, lens, flags


def sibling_backward(kind, tau, eta, alpha, beta, paths, rowsum, grad_logp, lens=None, aux_vec=None, aux_mat=None,
                     scalar0=0.0, item_weights=None):
    """Gradient of sum(grad_logp * log_probs) w.r.t. eta for a fused sibling construction -> [B,n,n]."""
    _require_gpu(tau, eta, paths, rowsum, grad_logp, aux_vec, aux_mat, item_weights)
    n = tau.shape[-1]
    B, rows, A = paths.shape
    tau, tbs = _bstride(tau, n)
    eta, ebs = _bstride(eta, n)
    paths = paths.contiguous()
    rowsum, grad_logp = _f32c(rowsum), _f32c(grad_logp)
    if aux_vec is not None:
        aux_vec = _f32c(aux_vec).reshape(-1, n)
        if aux_vec.shape[0] != B:
            aux_vec = aux_vec.expand(B, n).contiguous()
    abs_ = 0
    if aux_mat is not None:
        aux_mat, abs_ = _bstride(aux_mat, n)
    mdim = 0
    if item_weights is not None:
        item_weights = _f32c(item_weights)
        mdim = item_weights.shape[-1]
        if item_weights.dim() == 2:
            item_weights = item_weights.unsqueeze(0).expand(B, n, mdim).contiguous()
    dev = paths.device
    with _on(dev):
        grad = torch.zeros((B, n, n), dtype=torch.float32, device=dev)
        rc = _lib.lib().daco_sibling_backward(
            _stream(dev), SIB_KINDS[kind], B, n, A, rows, tau.data_ptr(), tbs, eta.data_ptr(), ebs, float(alpha),
            float(beta), aux_vec.data_ptr() if aux_vec is not None else None,
            aux_mat.data_ptr() if aux_mat is not None else None, abs_, float(scalar0),
            item_weights.data_ptr() if item_weights is not None else None, mdim, paths.data_ptr(), rowsum.data_ptr(),
            grad_logp.data_ptr(), lens.contiguous().data_ptr() if lens is not None else None, grad.data_ptr())
    _lib.check(rc, "daco_sibling_backward")
    return grad


class PickService:
    """ACO.pick_move as a service for the sibling problems (op, pctsp, sop, smtwtp, bpp, mkp):
    build the fused transition matrix once per construction, then draw one action per ant per call
    from a caller-maintained mask (include/deepaco_hip.h: daco_prob_matrix + daco_pick_move)."""

    def __init__(self, tau, eta, n_ants, alpha=1.0, beta=1.0, mode="scan", seed=0, it=0, ant_gid0=0):
        _require_gpu(tau, eta)
        self.n = tau.shape[-1]
        self.B = tau.shape[0] if tau.dim() == 3 else (eta.shape[0] if eta.dim() == 3 else 1)
        self.A, self.mode = n_ants, (MODES[mode] if isinstance(mode, str) else int(mode))
        self.seed, self.it, self.gid0, self.dev = int(seed) & (2 ** 64 - 1), int(it), int(ant_gid0), tau.device
        tau, tbs = _bstride(tau, self.n)
        eta, ebs = _bstride(eta, self.n)
        L = _lib.lib()
        with torch.cuda.device(self.dev):
            nbytes = L.daco_tsp_sample_workspace_bytes(self.B, self.n, self.mode)
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)     # owned: lives across steps
            rc = L.daco_prob_matrix(_stream(self.dev), self.B, self.n, tau.data_ptr(), tbs, eta.data_ptr(), ebs,
                                    float(alpha), float(beta), self.mode, self.ws.data_ptr(), self.ws.numel())
        _lib.check(rc, "daco_prob_matrix")
        self.flags = torch.zeros((self.B,), dtype=torch.int32, device=self.dev)

    def pick(self, prev, mask, step, require_prob=False, noise=None):
        """prev [B,A] (or [A]) int64, mask [B,A,n] (or [A,n]) float -> (actions, log_probs|None, rowsum|None)
        with the leading batch dimension of the inputs."""
        squeeze = prev.dim() == 1
        prev = prev.reshape(self.B, self.A).to(torch.int64).contiguous()
        mask = _f32c(mask).reshape(self.B, self.A, self.n)
        m = RACE_NOISE if noise is not None else self.mode
        if noise is not None:
            noise = _f32c(noise).reshape(self.B, self.A, self.n)
        with torch.cuda.device(self.dev):
            actions = torch.empty((self.B, self.A), dtype=torch.int64, device=self.dev)
            logp = torch.empty((self.B, self.A), dtype=torch.float32, device=self.dev) if require_prob else None
            rowsum = torch.empty((self.B, self.A), dtype=torch.float32, device=self.dev) if require_prob else None
            rc = _lib.lib().daco_pick_move(_stream(self.dev), self.B, self.n, self.A, self.ws.data_ptr(), self.ws.numel(),
                                           m, prev.data_ptr(), mask.data_ptr(),
                                           noise.data_ptr() if noise is not None else None, self.seed, self.it,
                                           self.gid0, int(step), actions.data_ptr(),
                                           logp.data_ptr() if require_prob else None,
                                           rowsum.data_ptr() if require_prob else None, self.flags.data_ptr())
        _lib.check(rc, "daco_pick_move")
        if squeeze:
            return actions[0], (logp[0] if require_prob else None), (rowsum[0] if require_prob else None)
        return actions, logp, rowsum


def tsp_knn_graph(coords, k_sparse, diag=1e9, want_dist=True):
    """Batched gen_pyg_data (tsp/utils.py:16-36): coords [B,n,2] -> (dist [B,n,n] | None, edge_index [B,2,n*k],
    edge_attr [B,n*k,1]) in one launch.  The same launch writes the batch as one block-diagonal int32 graph (the form the
    network's kernels take); it rides on the returned edge_index as `_daco_csr`, and Net.forward_batch(..., k_sparse=k) uses it
    instead of deriving it again (valid as long as that tensor object is not written to: its version counter is checked)."""
    _require_gpu(coords)
    coords = _f32c(coords)
    B, n, _ = coords.shape
    dev = coords.device
    E = n * int(k_sparse)
    with _on(dev):
        dist = torch.empty((B, n, n), dtype=torch.float32, device=dev) if want_dist else None
        ei = torch.empty((B, 2, E), dtype=torch.int64, device=dev)
        ea = torch.empty((B, E, 1), dtype=torch.float32, device=dev)
        src32 = torch.empty((B * E,), dtype=torch.int32, device=dev)
        dst32 = torch.empty((B * E,), dtype=torch.int32, device=dev)
        # src / dst rows of one instance are the two halves of its [2, E] block
        src = ei[:, 0]
        dst = ei[:, 1]
        if B > 1:                       # the kernel writes [B][E] arrays: use separate contiguous buffers
            src_c = torch.empty((B, E), dtype=torch.int64, device=dev)
            dst_c = torch.empty((B, E), dtype=torch.int64, device=dev)
        else:
            src_c, dst_c = src, dst
        rc = _lib.lib().daco_tsp_knn_graph_csr(_stream(dev), B, n, int(k_sparse), coords.data_ptr(), float(diag),
                                               dist.data_ptr() if want_dist else None, src_c.data_ptr(), dst_c.data_ptr(),
                                               ea.data_ptr(), src32.data_ptr(), dst32.data_ptr())
        if B > 1:
            ei[:, 0], ei[:, 1] = src_c, dst_c
    _lib.check(rc, "daco_tsp_knn_graph_csr")
    ei._daco_csr = (src32, dst32, n, int(k_sparse), ei._version)
    return dist, ei, ea


def heu_matrix(n, edge_index, heu, fill=0.0, add=0.0, check=False):
    """Net.reshape for a batch (tsp/net.py:94-102) with its callers' `+ eps`: edge_index [B,2,E] int64 (graph-local ids),
    heu [B,E] f32 -> [B,n,n] f32 = `fill` everywhere, heu + add at [src, dst].  check=True: raise if an id lies outside
    [0, n) (the reference's indexed assignment would; this syncs) -- otherwise such edges are skipped."""
    _require_gpu(edge_index, heu)
    B, E = heu.shape
    assert edge_index.shape == (B, 2, E) and edge_index.dtype == torch.int64
    ei = edge_index.contiguous()
    heu = _f32c(heu.detach())
    dev = heu.device
    with _on(dev):
        out = torch.empty((B, n, n), dtype=torch.float32, device=dev)
        bad = torch.zeros((1,), dtype=torch.int32, device=dev) if check else None
        rc = _lib.lib().daco_heu_matrix(_stream(dev), B, n, E, ei.data_ptr(), heu.data_ptr(), float(fill), float(add),
                                        out.data_ptr(), bad.data_ptr() if check else None)
    _lib.check(rc, "daco_heu_matrix")
    if check and int(bad.item()):
        raise IndexError(f"heu_matrix: {int(bad.item())} edge(s) with a node id outside [0, {n})")
    return out


def tour_costs(dist, paths, closed=True):
    """ACO.gen_path_costs for a batch (tsp/aco.py:121-132; closed=False: cvrp/aco.py:133-136)."""
    _require_gpu(dist, paths)
    n = dist.shape[-1]
    B, length, A = paths.shape
    dist, dbs = _bstride(dist, n)
    paths = paths.contiguous()
    dev = paths.device
    with _on(dev):
        costs = torch.empty((B, A), dtype=torch.float32, device=dev)
        rc = _lib.lib().daco_tour_costs(_stream(dev), B, n, length, A, dist.data_ptr(), dbs, paths.data_ptr(),
                                        int(closed), costs.data_ptr())
    _lib.check(rc, "daco_tour_costs")
    return costs


def track_best_(costs, paths, lowest, shortest=None, mmas_scale=None, tours16=None):
    """Best-so-far bookkeeping of ACO.run on the device (tsp/aco.py:78-88): updates lowest [B] and shortest
    [B,len] in place where this iteration's first-minimum cost beats the record.  mmas_scale (= problem size):
    also returns the MMAS upper bound n / lowest_cost [B] (computed like the reference's rtruediv).
    tours16 (with paths=None): the tours as sparse_tours16() rows [B, A, ld] instead of int64 paths [B, len, A]; len = shortest's."""
    _require_gpu(costs, paths, lowest, shortest, tours16)
    assert costs.dtype == torch.float32 and costs.is_contiguous()
    B, A = costs.shape
    assert lowest.dtype == torch.float32 and lowest.is_contiguous() and lowest.numel() == B
    dev = costs.device
    with _on(dev):
        mx = torch.empty((B,), dtype=torch.float32, device=dev) if mmas_scale is not None else None
        if paths is None:
            assert tours16 is not None and tours16.dtype == torch.int16 and tours16.is_contiguous() and shortest is not None
            assert tuple(tours16.shape[:2]) == (B, A)
            rc = _lib.lib().daco_track_best_tours16(_stream(dev), B, int(shortest.shape[1]), A, int(tours16.shape[2]), costs.data_ptr(),
                                                    tours16.data_ptr(), lowest.data_ptr(), shortest.data_ptr(), None,
                                                    mx.data_ptr() if mx is not None else None,
                                                    float(mmas_scale) if mmas_scale is not None else 0.0)
            _lib.check(rc, "daco_track_best_tours16")
            return mx
        length = paths.shape[1]
        assert paths.is_contiguous() and tuple(paths.shape) == (B, length, A)
        rc = _lib.lib().daco_track_best(_stream(dev), B, length, A, costs.data_ptr(), paths.data_ptr(), lowest.data_ptr(),
                                        shortest.data_ptr() if shortest is not None else None, None,
                                        mx.data_ptr() if mx is not None else None,
                                        float(mmas_scale) if mmas_scale is not None else 0.0)
    _lib.check(rc, "daco_track_best")
    return mx


def pheromone_update_(tau, paths, costs, decay, elitist=False, symmetric=True, clamp_min=None,
                      clamp_max=None, floor=0.0, nbr=None, weights=None, hub=0, heads=None):
    """In-place ACO.update_pheronome for a batch (tsp/aco.py:95-118, cvrp/aco.py:107-130).

    tau [B,n,n] f32 contiguous (modified in place); clamp_min/clamp_max: [B] f32 tensors or None.
    weights [B,A]: explicit deposit per ant (default 1/cost); hub: see include/deepaco_hip.h.
    heads (symmetric only): dict(eta, alpha, beta, head, race, workspace) of a colony whose next construction is
    tsp_sample_sparse(..., workspace=workspace, heads_ready=True): the update also writes that call's head rows
    (daco_pheromone_update_heads: tau is read once per iteration instead of twice)."""
    _require_gpu(tau, paths, costs, clamp_min, clamp_max)
    assert tau.dim() == 3 and tau.dtype == torch.float32 and tau.is_contiguous()
    B, n, _ = tau.shape
    if paths is None:                                        # (the table is all the deposit reads)
        assert nbr is not None and symmetric
        length, A = n, costs.shape[-1]
    else:
        _, length, A = paths.shape
        paths = paths.contiguous()
    pptr = paths.data_ptr() if paths is not None else None
    costs = _f32c(costs)
    if weights is not None:
        weights = _f32c(weights)
    dev = tau.device
    L = _lib.lib()
    with _on(dev):
        nbytes = L.daco_pheromone_update_workspace_bytes(B, n, length, A)
        ws = _workspace(dev, nbytes, "update")
        if heads is not None:
            assert symmetric and length == n
            eta, ebs = _bstride(heads["eta"], n)
            head, sws = heads["head"], heads["workspace"]
            _require_gpu(eta, head, sws)
            rc = L.daco_pheromone_update_heads(_stream(dev), B, n, A, tau.data_ptr(), pptr, costs.data_ptr(), float(decay),
                                               int(bool(elitist)), clamp_min.data_ptr() if clamp_min is not None else None,
                                               clamp_max.data_ptr() if clamp_max is not None else None, float(floor),
                                               nbr.data_ptr() if nbr is not None else None,
                                               weights.data_ptr() if weights is not None else None, ws.data_ptr(), ws.numel(),
                                               eta.data_ptr(), ebs, float(heads["alpha"]), float(heads["beta"]), head.data_ptr(),
                                               int(head.shape[2]), int(bool(heads.get("race", False))),
                                               int(bool(heads.get("nbr_grouped", False))) if nbr is not None else 0, sws.data_ptr(), sws.numel())
            _lib.check(rc, "daco_pheromone_update_heads")
            return tau
        rc = L.daco_pheromone_update(_stream(dev), B, n, length, A, tau.data_ptr(), pptr,
                                     costs.data_ptr(), float(decay), int(bool(elitist)), int(bool(symmetric)),
                                     clamp_min.data_ptr() if clamp_min is not None else None,
                                     clamp_max.data_ptr() if clamp_max is not None else None,
                                     float(floor), nbr.data_ptr() if nbr is not None else None,
                                     weights.data_ptr() if weights is not None else None, int(hub),
                                     ws.data_ptr(), ws.numel())
    _lib.check(rc, "daco_pheromone_update")
    return tau


class TwoOptTables:
    """Sorted neighbour lists + tolerance ranks of a batch of matrices for the candidate-list 2-opt kernel
    (daco_two_opt_prepare).  Built once per matrix; `tables_t` are the tables of the transposed matrices (the same
    object for symmetric matrices)."""

    def __init__(self, dist, dist_t=None):
        _require_gpu(dist)
        n = dist.shape[-1]
        if n > 1024:
            raise _lib.DacoError(f"two_opt tables: n = {n} above 1024")
        if dist_t is None:
            dist_t = transposed_for_two_opt(dist)
        self.n = n
        self.B = 1 if dist.dim() == 2 else dist.shape[0]      # instances the tables were built for
        self.dist_t = dist_t                       # what two_opt_'s dense kernel wants as well
        self.tables = self._build(dist)
        self.tables_t = self.tables if isinstance(dist_t, str) or dist_t is dist else self._build(dist_t)

    @staticmethod
    def _build(m):
        n = m.shape[-1]
        m, dbs = _bstride(m, n)
        B = 1 if m.dim() == 2 else m.shape[0]
        L = _lib.lib()
        dev = m.device
        with _on(dev):
            nbytes = L.daco_two_opt_tables_bytes(B, n)
            buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            rc = L.daco_two_opt_prepare(_stream(dev), B, n, m.data_ptr(), dbs, buf.data_ptr(), nbytes)
        _lib.check(rc, "daco_two_opt_prepare")
        return buf


def two_opt_(dist, tours, max_iterations=1000, want_sweeps=False, dist_t=None, tables=None, kernel="auto"):
    """In-place batched 2-opt (tsp_nls/two_opt.py:41-49).  dist [B,n,n] or [n,n];
    tours [B,T,n] or [T,n] int16/uint16 storage (values < 65536), one ROW per tour.
    dist_t: the transposed matrices (same shape as dist), "symmetric" if dist equals its transpose, or None: only
    changes how the kernel reads the matrix (see include/deepaco_hip.h), never the result.
    tables: a TwoOptTables of `dist` -> the candidate-list kernel takes over whenever a tour's candidate count is small
    (same moves, same result; far less work per sweep on tours near a local optimum, more on tours with many long edges:
    kernel="auto" switches per tour between it and the dense kernel, kernel="nbr" forces the candidate lists)."""
    _require_gpu(dist, tours)
    n = dist.shape[-1]
    if tables is not None:
        assert tours.dtype in (torch.int16, torch.uint16) and tours.is_contiguous() and tables.n == n
        t3 = tours if tours.dim() == 3 else tours.unsqueeze(0)
        shape = t3.shape[:2]
        dist, dbs = _bstride(dist, n)
        if dist.dim() == 2:                         # one matrix (and one table set) for every tour
            t3 = t3.view(1, -1, n)
        B, T, _ = t3.shape
        assert tables.B == B, f"two_opt_: tables built for {tables.B} instances, launch has {B}"
        dev = tours.device
        with _on(dev):
            if kernel == "cached":                  # one launch of the NLS kernel without rounds: dirty-list sweeps
                sweeps = torch.empty(tuple(shape), dtype=torch.int32, device=dev) if want_sweeps else None
                rc = _lib.lib().daco_tsp_nls(_stream(dev), B, T, n, dist.data_ptr(), dbs, tables.tables.data_ptr(),
                                             tables.tables_t.data_ptr(), None, 0, None, None, t3.data_ptr(),
                                             int(max_iterations), 0, 0, sweeps.data_ptr() if want_sweeps else None,
                                             None, None)
            elif kernel == "nbr":
                sweeps = torch.empty(tuple(shape), dtype=torch.int32, device=dev) if want_sweeps else None
                rc = _lib.lib().daco_two_opt_nbr(_stream(dev), B, T, n, dist.data_ptr(), dbs, tables.tables.data_ptr(),
                                                 tables.tables_t.data_ptr(), t3.data_ptr(), int(max_iterations),
                                                 sweeps.data_ptr() if want_sweeps else None)
            else:
                assert kernel == "auto"
                dt = tables.dist_t
                dt = dist if isinstance(dt, str) else (None if dt is None else _bstride(dt, n)[0])
                sweeps = torch.empty(tuple(shape), dtype=torch.int32, device=dev)
                rc = _lib.lib().daco_two_opt_auto(_stream(dev), B, T, n, dist.data_ptr(),
                                                  dt.data_ptr() if dt is not None else None, dbs,
                                                  tables.tables.data_ptr(), tables.tables_t.data_ptr(), t3.data_ptr(),
                                                  int(max_iterations), sweeps.data_ptr())
        _lib.check(rc, "daco_two_opt_" + kernel)
        return (tours, sweeps) if want_sweeps else tours
    if isinstance(dist_t, str):
        assert dist_t == "symmetric"
        dist_t = dist
    assert tours.dtype in (torch.int16, torch.uint16) and tours.is_contiguous()
    t3 = tours if tours.dim() == 3 else tours.unsqueeze(0)
    B, T, _ = t3.shape
    same = dist_t is dist
    dist, dbs = _bstride(dist, n)
    if dist_t is not None:
        dist_t = dist if same else _bstride(dist_t, n)[0]
        assert dist_t.shape == dist.shape
    dev = tours.device
    with _on(dev):
        sweeps = torch.empty((B, T), dtype=torch.int32, device=dev) if want_sweeps else None
        rc = _lib.lib().daco_two_opt(_stream(dev), B, T, n, dist.data_ptr(),
                                     dist_t.data_ptr() if dist_t is not None else None, dbs, t3.data_ptr(),
                                     int(max_iterations), sweeps.data_ptr() if want_sweeps else None)
    _lib.check(rc, "daco_two_opt")
    return (tours, sweeps) if want_sweeps else tours


def cvrp_local_search_(dist, demand, capacity, paths, max_moves, want_stats=False):
    """In-place local search on CVRP solutions (cvrp_nls/aco.py:114-126): dist [B,n,n] or [n,n], demand [B,n] or [n],
    paths [B,Lmax,A] or [Lmax,A] int64 (route sequences as gen_path returns them).  Best improvement over HGS's move
    families (relocate 1 / 2 / 2 reversed, swap 1-1 / 2-1 / 2-2, 2-opt, 2-opt* both ways; SWAP* when none of them
    improves), hard capacity, at most max_moves moves per solution (csrc/daco_cvrp_ls.hip has the specification).
    Returns paths (and lens, moves [B,A])."""
    _require_gpu(dist, demand, paths)
    n = dist.shape[-1]
    assert paths.dtype == torch.int64
    p3 = paths if paths.dim() == 3 else paths.unsqueeze(0)
    assert p3.is_contiguous()
    B, Lmax, A = p3.shape
    dist, dbs = _bstride(dist, n)
    demand = _f32c(demand)
    if demand.dim() == 1:
        demand = demand.unsqueeze(0).expand(B, n).contiguous()
    dev = paths.device
    with _on(dev):
        lens = torch.empty((B, A), dtype=torch.int32, device=dev) if want_stats else None
        moves = torch.empty((B, A), dtype=torch.int32, device=dev) if want_stats else None
        rc = _lib.lib().daco_cvrp_local_search(_stream(dev), B, n, A, Lmax, dist.data_ptr(), dbs, demand.data_ptr(),
                                               float(capacity), p3.data_ptr(), int(max_moves),
                                               lens.data_ptr() if want_stats else None,
                                               moves.data_ptr() if want_stats else None)
    _lib.check(rc, "daco_cvrp_local_search")
    return (paths, lens, moves) if want_stats else paths


class HgsTables:
    """What HGS's Params derives from a matrix (Params.cpp:77-103, LocalSearch.cpp:9): per instance the largest entry, the
    correlated vertices (nb_granular nearest, symmetric) and the shuffled node order -- daco_hgs_prepare's output, built once
    per matrix and shared by every ant (and iteration, for the distance matrix)."""

    def __init__(self, matrix, nb_granular=20):
        _require_gpu(matrix)
        m = matrix if matrix.dim() == 3 else matrix.unsqueeze(0)
        self.matrix = m if (m.dtype == torch.float64 and m.is_contiguous()) else m.contiguous().double()
        self.B, self.n = self.matrix.shape[0], self.matrix.shape[-1]
        # the transposed copy the search reads "column" entries from (None for a symmetric matrix: one device comparison)
        mt = self.matrix.transpose(-1, -2)
        self.matrix_t = None if bool(torch.equal(self.matrix, mt)) else mt.contiguous()
        self.nb_granular = int(nb_granular)
        L = _lib.lib()
        self.table_bytes = L.daco_hgs_table_bytes(self.n, self.nb_granular)
        dev = self.matrix.device
        with _on(dev):
            self.tables = torch.empty(self.B * self.table_bytes, dtype=torch.uint8, device=dev)
            rc = L.daco_hgs_prepare(_stream(dev), self.B, self.n, self.matrix.data_ptr(), self.n * self.n, self.nb_granular,
                                    self.tables.data_ptr())
        _lib.check(rc, "daco_hgs_prepare")


def hgs_local_search_(paths, stages, demand, capacity=1000.001, demand_scale=1000.0, want_stats=False):
    """The reference's CVRP local search on every column of `paths`, route for route (csrc/daco_hgs_ls.hip; cvrp_nls/aco.py:
    114-126 -> swapstar.py:324-346 -> HGS LocalSearch::run as the reference runs it: moves 1-9, granular, no SWAP*).
    paths [B,Lmax,A] or [Lmax,A] int64, rewritten in place in merge_subroutes' layout; stages: up to three
    (HgsTables, count) pairs run one after the other on each solution (neural_swapstar: (dist, limit), (heuristic_dist, 10),
    (dist, limit)); demand [B,n] or [n] as the colony holds it (scaled by demand_scale = 1000 as swapstar.py:335 does).
    Returns paths (and status [B,A], stats [B,A,4] = moves, loops, evaluation rounds, watchdog)."""
    _require_gpu(paths)
    assert paths.dtype == torch.int64 and 1 <= len(stages) <= 3
    p3 = paths if paths.dim() == 3 else paths.unsqueeze(0)
    assert p3.is_contiguous()
    B, Lmax, A = p3.shape
    t0 = stages[0][0]
    n, g = t0.n, t0.nb_granular
    dev = paths.device
    dem = demand.to(dev).double()
    if dem.dim() == 1:
        dem = dem.unsqueeze(0).expand(B, n)
    dem = (dem * demand_scale).contiguous()
    S = len(stages)
    mats = (C.c_void_p * S)(*[st[0].matrix.data_ptr() for st in stages])
    mats_t = (C.c_void_p * S)(*[(st[0].matrix_t.data_ptr() if st[0].matrix_t is not None else None) for st in stages])
    strides = (C.c_long * S)(*[(0 if st[0].B == 1 and B > 1 else n * n) for st in stages])
    tabs = (C.c_void_p * S)(*[st[0].tables.data_ptr() for st in stages])
    counts = (C.c_int * S)(*[int(st[1]) for st in stages])
    for st in stages:
        assert st[0].n == n and st[0].nb_granular == g and st[0].B in (1, B)
        if st[0].B == 1 and B > 1:
            raise ValueError("hgs_local_search_: one table set per instance is needed (B tables)")
    L = _lib.lib()
    with _on(dev):
        wsb = L.daco_hgs_workspace_bytes(B, n, A, Lmax, g)
        ws = _workspace(dev, wsb, "hgs_ls")
        status = torch.empty((B, A), dtype=torch.int32, device=dev)
        stats = torch.empty((B, A, 4), dtype=torch.int32, device=dev) if want_stats else None
        rc = L.daco_hgs_local_search(_stream(dev), B, n, A, Lmax, S, mats, mats_t, strides, tabs, counts, dem.data_ptr(), float(capacity), g,
                                     p3.data_ptr(), status.data_ptr(), stats.data_ptr() if want_stats else None,
                                     ws.data_ptr(), ws.numel())
    _lib.check(rc, "daco_hgs_local_search")
    return (paths, status, stats) if want_stats else paths


@torch.no_grad()
def transposed_for_two_opt(m):
    """What two_opt_'s dist_t wants for matrix m: "symmetric" if m equals its transpose (one device comparison),
    else a transposed contiguous copy."""
    mt = m.transpose(-1, -2)
    return "symmetric" if bool(torch.equal(m, mt)) else mt.contiguous()


def two_opt_tables(dist, dist_t=None):
    """TwoOptTables(dist) where the candidate-list kernel applies (n <= 1024), else None (two_opt_ then runs the dense kernel)."""
    return TwoOptTables(dist, dist_t) if dist.shape[-1] <= 1024 else None


def nls_(dist, heuristic_dist, tours, maxt, T_nls=10, T_p=20, dist_t=None, heuristic_dist_t=None, tables=None,
         heuristic_tables=None, fused=None, want_costs=False, counters=None):
    """Batched NLS driver (tsp_nls/aco.py:241-258) fully on the device.
    dist, heuristic_dist [B,n,n]; tours [B,T,n] int16 (one row per tour).  Returns improved tours (and, with want_costs,
    their f32 lengths as daco_tour_costs computes them).
    dist_t / heuristic_dist_t, tables / heuristic_tables: see two_opt_ (callers that run many iterations pass them once;
    the tables are built here otherwise -- one sort of every matrix row -- since the 21 passes of one NLS amortise them).
    fused (default: whenever the tables exist, i.e. n <= 1024; DACO_NLS_FUSED=0 turns it off): the whole search of a tour
    in one launch of daco_tsp_nls; otherwise 2 T_nls + 1 two_opt_ passes driven from here (the same tours either way)."""
    B, T, n = tours.shape
    if dist_t is None:
        dist_t = transposed_for_two_opt(dist) if tables is None else tables.dist_t
    if heuristic_dist_t is None:
        heuristic_dist_t = transposed_for_two_opt(heuristic_dist) if heuristic_tables is None else heuristic_tables.dist_t
    if tables is None:
        tables = two_opt_tables(dist, dist_t)
    if heuristic_tables is None:
        heuristic_tables = two_opt_tables(heuristic_dist, heuristic_dist_t)
    if fused is None:
        fused = os.environ.get("DACO_NLS_FUSED", "1") != "0"

    def lengths(t):
        return tour_costs(dist, t.permute(0, 2, 1).to(torch.int64).contiguous())

    if fused and tables is not None and heuristic_tables is not None:
        _require_gpu(dist, heuristic_dist, tours)
        assert tours.dtype in (torch.int16, torch.uint16)
        assert tables.B == B and heuristic_tables.B == B and dist.dim() == 3 and heuristic_dist.dim() == 3
        best = tours.clone().contiguous()
        d, dbs = _bstride(dist, n)
        h, hbs = _bstride(heuristic_dist, n)
        dev = tours.device
        with _on(dev):
            costs = torch.empty((B, T), dtype=torch.float32, device=dev) if want_costs else None
            rc = _lib.lib().daco_tsp_nls(_stream(dev), B, T, n, d.data_ptr(), dbs, tables.tables.data_ptr(),
                                         tables.tables_t.data_ptr(), h.data_ptr(), hbs,
                                         heuristic_tables.tables.data_ptr(), heuristic_tables.tables_t.data_ptr(),
                                         best.data_ptr(), int(maxt), int(T_nls), int(T_p), None,
                                         costs.data_ptr() if want_costs else None,
                                         counters.data_ptr() if counters is not None else None)
        _lib.check(rc, "daco_tsp_nls")
        return (best, costs) if want_costs else best

    best = tours.clone().contiguous()
    two_opt_(dist, best, maxt, dist_t=dist_t, tables=tables)
    best_costs = lengths(best)
    new = best
    for _ in range(T_nls):
        pert = new.clone()
        two_opt_(heuristic_dist, pert, T_p, dist_t=heuristic_dist_t, tables=heuristic_tables)
        two_opt_(dist, pert, maxt, dist_t=dist_t, tables=tables)
        new = pert
        new_costs = lengths(new)
        improved = new_costs < best_costs
        best = torch.where(improved.unsqueeze(2), new, best)
        best_costs = torch.where(improved, new_costs, best_costs)
    return (best, best_costs) if want_costs else best


class BatchedTSP:
    """B independent TSP colonies advanced in lock-step on one GPU (the throughput path).

    Semantics per instance are those of tsp/aco.py ACO.run (AS / elitist / MMAS); best-so-far
    tracking is done on the device, so an iteration never synchronises with the host."""

    def __init__(self, distances, n_ants=20, decay=0.9, alpha=1, beta=1, elitist=False, min_max=False,
                 pheromone=None, heuristic=None, min=None, sampler="auto", seed=None, ant_gid0=0,
                 fixed_start=-1, local_search=None, inference=False):
        """sampler: 'auto' (default: head / tail rows where they apply -- after sparsify(k) or on a concentrated heuristic,
        129 <= n <= 1024 -- else the dense scan; resolve_sampler), 'scan', 'scan_wave', 'race', 'scan_sparse'."""
        _require_gpu(distances)
        assert distances.dim() == 3
        self.distances = _f32c(distances)
        self.B, self.n = distances.shape[0], distances.shape[1]
        self.n_ants, self.decay, self.alpha, self.beta = n_ants, decay, alpha, beta
        self.elitist, self.min_max = elitist, min_max
        dev = distances.device
        if min_max:
            self.min = 0.1 if min is None else min
            assert self.min > 1e-9
            self.max = None
        self.pheromone = torch.ones_like(self.distances) if pheromone is None else _f32c(pheromone).clone()
        if min_max and pheromone is None:
            self.pheromone = self.pheromone * self.min
        self.heuristic = (1 / self.distances) if heuristic is None else heuristic
        self.lowest_cost = torch.full((self.B,), float("inf"), device=dev)
        self.shortest_path = torch.zeros((self.B, self.n), dtype=torch.int64, device=dev)
        self.sampler = sampler
        self.seed = torch.initial_seed() if seed is None else seed
        self.iteration = 0
        self.ant_gid0 = ant_gid0
        self.fixed_start = fixed_start
        assert local_search in (None, "2opt", "nls")
        self.local_search = local_search          # tsp_nls/aco.py: applied to the tours before costing
        self.inference = inference                # tsp_nls/aco.py:235,242: 2-opt sweeps n//4 (training) or 10000 (inference)
        self._hdist = None
        self._hdist_t = None
        self._dist_t = None
        self._tables = self._htables = None
        self._cmin = None
        self.nls_counters = None                  # optional int64[2] on the device: sweeps, list entries walked (bench)
        # sampler="scan_sparse": head / tail rows (tsp_sample_sparse).  The head of a row = the k largest heuristic entries:
        # k from sparsify(k), else `head_k` (default n // 10, at most 127); rebuilt when the heuristic object changes.
        self.head_k = None
        self._head = None
        # the head-row samplers' workspace is the colony's own, and the pheromone update leaves the NEXT iteration's head rows in
        # it (pheromone_update_(heads=...)); _heads_for names the state those rows were formed from -- the next step takes them
        # only if pheromone (object and version counter), heuristic, head table and exponents are still those
        self._sparse_ws = None
        self._heads_for = None
        self._flags = torch.zeros((self.B,), dtype=torch.int32, device=dev)      # sticky: bit 0 a draw without a candidate, bit 2 see daco_tsp_sample_heads
        self.fuse_head_rows = os.environ.get("DACO_FUSE_HEAD_ROWS", "1") != "0"      # (knob: 0 = a pre-pass every iteration, as until round 5)

    def check_feasible(self):
        """Raise like the reference's Categorical if any head-row draw so far had no candidate (bit 0), or the head table holds more
        live entries than the colony said (bit 2: daco_tsp_sample_heads); syncs.  The flag words are sticky."""
        fl = int(self._flags.max())
        if fl & 1:
            raise ValueError("BatchedTSP: a transition row had no feasible candidate")
        if fl & 4:
            raise RuntimeError("BatchedTSP: the head table has more live entries per row than head_live_max")

    def _heads_state(self, head, race):
        return (self.pheromone, self.pheromone._version, self.heuristic, head, bool(race), float(self.alpha), float(self.beta))

    def _heuristic_dist(self):
        if self._hdist is None:
            h = self.heuristic.detach().float()
            self._hdist = (1 / (h / h.amax(dim=-1, keepdim=True) + 1e-5)).contiguous()
        return self._hdist

    @torch.no_grad()
    def sparsify(self, k_sparse):
        """tsp/aco.py:52-67 for a batch: 1/dist on each node's k nearest edges, 1e-10 elsewhere."""
        _, idx = torch.topk(self.distances, k=k_sparse, dim=2, largest=False)
        sparse = torch.full_like(self.distances, 1e10)
        sparse.scatter_(2, idx, torch.gather(self.distances, 2, idx))
        self.heuristic = 1 / sparse
        self.head_k = min(int(k_sparse), 127)
        self._head = None

    def resolved_sampler(self):
        """(kernel family this colony's next step runs, head size): see resolve_sampler."""
        if getattr(self, "_auto", None) is None:
            self._auto = {}
        return resolve_sampler(self.sampler, self.n, self.head_k, self.heuristic, self._auto)

    def _head_table(self, k=None):
        """(heuristic object it was built from, [B,n,64] head ids) for sampler='scan_sparse'."""
        if self._head is None or self._head[0] is not self.heuristic or (k is not None and self._head[2] != k):
            k = k if k is not None else (self.head_k if self.head_k is not None else max(1, min(127, self.n // 10)))
            h = self.heuristic.detach()
            h = h if h.dim() == 3 else h.unsqueeze(0).expand(self.B, self.n, self.n)
            self._head = (self.heuristic, sparse_head(_f32c(h), k, top=take_auto_top(getattr(self, "_auto", None), self.heuristic)), k)
        return self._head[1]

    @torch.no_grad()
    def step(self, events=None, _iter_dev=None, ls_events=None, want_paths=True):
        # want_paths=False (head-row samplers without local search; what run() passes): the iteration keeps its tours to itself, as
        # ACO.run does (tsp/aco.py:75-92: `paths` is a local of the loop) -- the construction kernel writes them as compact u16 rows
        # in its workspace (32 MB instead of 131 MB of int64 [B, n, A] per iteration at TSP-500 x 512 x 64), the best one is copied
        # from there, the deposit takes the table: returns (None, costs); sparse_tours16(self._sparse_ws, ...) holds the tours.
        # (_iter_dev: device-side iteration counter of a captured graph; self.iteration then stays frozen)
        # events: torch.cuda.Event pair re-recorded around the construction kernel; ls_events: a pair recorded (on the
        # current stream, which is the stream the library launches on) right before / after the local-search launches
        # sampler="race" after sparsify(k): the same tours from the head rows (daco_tsp_sample_race_head), an eighth of the noise
        sampler, hk = self.resolved_sampler()
        race_head = sampler == "race" and self.head_k is not None and 128 < self.n <= 1024
        heads = None
        if sampler == "scan_sparse" or race_head:
            head = self._head_table(hk)
            unit = float(self.alpha) == 1.0 and float(self.beta) == 1.0
            if self._sparse_ws is None:
                self._sparse_ws = sparse_workspace(self.distances.device, self.B, self.n, self.n_ants, unit_exponents=unit)
            fused = self.fuse_head_rows and unit                 # (the update forms the rows of tau itself: unit exponents only)
            grouped = fused and self.n_ants % 8 == 0 and self.local_search is None
            st = self._heads_state(head, race_head)
            ready = self._heads_for is not None and len(st) == len(self._heads_for) and all(
                (a is b) if torch.is_tensor(a) else (a == b) for a, b in zip(st, self._heads_for))
            compact = not want_paths and self.local_search is None
            paths, _, costs, nbr = tsp_sample_sparse(self.pheromone, self.heuristic, self.n_ants, head, self.alpha,
                                                     self.beta, seed=self.seed, it=self.iteration, ant_gid0=self.ant_gid0,
                                                     fixed_start=self.fixed_start, batch=self.B, events=events,
                                                     dist=self.distances, want_nbr=True, iter_dev=_iter_dev, race=race_head,
                                                     workspace=self._sparse_ws, heads_ready=ready, head_live_max=self._head[2],
                                                     nbr_grouped=grouped, flags=self._flags, want_paths=not compact)
            tours16 = sparse_tours16(self._sparse_ws, self.B, self.n, self.n_ants) if compact else None
            if fused:
                heads = {"eta": self.heuristic, "alpha": self.alpha, "beta": self.beta, "head": head, "race": race_head,
                         "workspace": self._sparse_ws, "nbr_grouped": grouped}
        else:
            tours16 = None
            paths, _, _, _, costs, nbr = tsp_sample(self.pheromone, self.heuristic, self.n_ants, self.alpha,
                                                    self.beta, mode=sampler, seed=self.seed, it=self.iteration,
                                                    ant_gid0=self.ant_gid0, fixed_start=self.fixed_start,
                                                    batch=self.B, events=events, dist=self.distances, want_nbr=True,
                                                    iter_dev=_iter_dev)
        if _iter_dev is None:
            self.iteration += 1
        if self.local_search is not None:
            ls_costs = None
            tours = paths.permute(0, 2, 1).to(torch.int16).contiguous()
            if ls_events:
                ls_events[0].record()
            maxt = 10000 if self.inference else self.n // 4
            if self._dist_t is None:
                self._dist_t = transposed_for_two_opt(self.distances)
                self._tables = two_opt_tables(self.distances, self._dist_t)
            if self.local_search == "2opt":
                two_opt_(self.distances, tours, maxt, dist_t=self._dist_t, tables=self._tables)
            else:
                hd = self._heuristic_dist()
                if self._hdist_t is None:
                    self._hdist_t = transposed_for_two_opt(hd)
                    self._htables = two_opt_tables(hd, self._hdist_t)
                # (costs: the fused search sums the tour lengths in daco_tour_costs' order, bit for bit)
                tours, ls_costs = nls_(self.distances, hd, tours, maxt, dist_t=self._dist_t,
                                       heuristic_dist_t=self._hdist_t, tables=self._tables,
                                       heuristic_tables=self._htables, want_costs=True, counters=self.nls_counters)
            if ls_events:
                ls_events[1].record()
            paths = tours.permute(0, 2, 1).to(torch.int64).contiguous()
            costs, nbr = (tour_costs(self.distances, paths) if ls_costs is None else ls_costs), None
        # in place: the best-so-far state lives at fixed addresses (a captured graph replays these very writes)
        new_max = track_best_(costs, paths, self.lowest_cost, self.shortest_path,
                              mmas_scale=self.n if self.min_max else None, tours16=tours16)
        cmin = cmax = None
        if self.min_max:
            if self.max is None:
                self.pheromone *= (new_max / self.pheromone.amax(dim=(1, 2))).view(self.B, 1, 1)
            self.max = new_max
            if self._cmin is None:
                self._cmin = torch.full_like(new_max, self.min)
            cmin, cmax = self._cmin, new_max
        pheromone_update_(self.pheromone, paths, costs, self.decay, self.elitist, True, cmin, cmax, nbr=nbr, heads=heads)
        self._heads_for = self._heads_state(heads["head"], heads["race"]) if heads is not None else None
        return paths, costs

    @torch.no_grad()
    def run(self, n_iterations, graph=False):
        """graph=True: the iteration is captured once into a HIP graph and replayed (small colonies are bound by
        launch latency: ~10 launches of a few microseconds of work each).  Same tours and pheromone as the eager
        loop: the Philox iteration counter of the captured sampler lives in device memory and is advanced inside
        the graph."""
        if not graph or n_iterations < 3:
            for _ in range(n_iterations):
                self.step(want_paths=False)
            return self.lowest_cost
        self.step(want_paths=False)                        # eager: workspaces, first-iteration MMAS rescale
        dev = self.distances.device
        it_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                      # one un-captured pass on the capture stream (allocator warm-up)
            self.step(_iter_dev=it_dev, want_paths=False)
            it_dev += 1
        torch.cuda.current_stream(dev).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            self.step(_iter_dev=it_dev, want_paths=False)
            it_dev += 1
        for _ in range(n_iterations - 2):
            g.replay()
        self.iteration += n_iterations - 1
        self._graph = g                                    # keeps the captured buffers alive with the colony
        return self.lowest_cost


class StreamedTSP:
    """The B colonies of a BatchedTSP as `parts` BatchedTSP colonies of B / parts instances, each on its own HIP stream.

    Instances are independent (no data-path exchange between them), so the parts need not advance in lock-step:
    the small kernels of one part's iteration -- deposit, transition matrix, best-so-far tracking: a tenth of an iteration --
    can run under another part's construction kernel.  Measured at TSP-500 x 512 ants x 64 instances
    (profiles/r04_streams.txt): a sustained loop gains 4 % with two parts (24.7 -> 25.8 M ant-tours/s) and nothing with four
    or eight (the parts' construction kernels then share the CUs and each runs longer); bench.py's default stays one colony.
    Ant ids keep the colony-wide numbering, so tours, costs, pheromone and best tours are those of ONE BatchedTSP over all
    instances, bit for bit.

    step() launches one iteration of every part and returns nothing (the parts' outputs stay per part: cols[p].step's
    returns are discarded); join() makes the current stream wait for the parts; lowest_cost / shortest_path / pheromone gather
    the parts (and join).  The constructor's arguments are BatchedTSP's; per-instance tensors are split along dim 0."""

    def __init__(self, distances, parts=4, ant_gid0=0, pheromone=None, heuristic=None, **kw):
        _require_gpu(distances)
        B = distances.shape[0]
        parts = max(1, min(int(parts), B))
        dev = distances.device
        n_ants = kw.get("n_ants", 20)
        bounds = [(p * B) // parts for p in range(parts + 1)]
        self.B, self.n, self.n_ants = B, distances.shape[1], n_ants
        self.cols, self.streams, self.bounds = [], [], bounds
        cur = torch.cuda.current_stream(dev)
        for p in range(parts):
            lo, hi = bounds[p], bounds[p + 1]

            def part(t):
                return None if t is None else (t[lo:hi] if t.dim() == 3 else t)
            self.cols.append(BatchedTSP(distances[lo:hi], ant_gid0=ant_gid0 + lo * n_ants, pheromone=part(pheromone),
                                        heuristic=part(heuristic), **kw))
            st = torch.cuda.Stream(device=dev)
            st.wait_stream(cur)
            self.streams.append(st)
        self._dev = dev

    def _fork(self):
        """The part streams wait for what the current stream has queued (a gather of the parts' state that is still
        pending there must not be overtaken by a step that rewrites that state in place -- ADVICE r4)."""
        cur = torch.cuda.current_stream(self._dev)
        for st in self.streams:
            st.wait_stream(cur)

    def sparsify(self, k_sparse):
        self._fork()
        for col, st in zip(self.cols, self.streams):
            with torch.cuda.stream(st):
                col.sparsify(k_sparse)
                col.heuristic = col.heuristic.contiguous()

    @property
    def iteration(self):
        return self.cols[0].iteration

    def step(self, events=None, want_paths=True):
        """events: one (begin, end) torch.cuda.Event pair PER PART, re-recorded around that part's construction kernel.
        want_paths=False: the parts keep their tours compact (BatchedTSP.step; the parts' paths are discarded here either way)."""
        self._fork()
        for p, (col, st) in enumerate(zip(self.cols, self.streams)):
            with torch.cuda.stream(st):
                col.step(events=events[p] if events is not None else None, want_paths=want_paths)

    def join(self):
        cur = torch.cuda.current_stream(self._dev)
        for st in self.streams:
            cur.wait_stream(st)

    def run(self, n_iterations):
        for _ in range(n_iterations):
            self.step(want_paths=False)
        return self.lowest_cost

    def _gather(self, name):
        self.join()
        cur = torch.cuda.current_stream(self._dev)
        src = [getattr(c, name) for c in self.cols]
        for t in src:                       # allocated on a part's stream, read on this one
            t.record_stream(cur)
        return torch.cat(src, dim=0)

    @property
    def lowest_cost(self):
        return self._gather("lowest_cost")

    @property
    def shortest_path(self):
        return self._gather("shortest_path")

    @property
    def pheromone(self):
        return self._gather("pheromone")

    @property
    def distances(self):
        return self._gather("distances")

    @property
    def heuristic(self):
        return self._gather("heuristic")


class BatchedCVRP:
    """B independent CVRP colonies in lock-step (cvrp/aco.py ACO.run semantics per instance, AS /
    elitist / MMAS); an iteration is sampler (+ fused costs and successor table) -> deposit, no host sync."""

    def __init__(self, distances, demand, n_ants=20, decay=0.9, alpha=1, beta=1, elitist=False, min_max=False,
                 pheromone=None, heuristic=None, min=None, capacity=50, sampler="scan", seed=None, ant_gid0=0,
                 local_search=None, ls_ants=8, inference=False):
        """local_search="hgs": the colony iteration of cvrp_nls/aco.py:134-171 (swapstar=True) for every instance -- the `ls_ants`
        cheapest ants of each instance go through neural_swapstar (cvrp_nls/aco.py:143-146, 443-448; daco_hgs_local_search, the
        reference's routes) before best tracking and the deposit.  The local search reads `distances` and the heuristic in the
        dtype they are passed in (float64 instance data as cvrp_nls/utils.py builds it) and the demands as demand / capacity."""
        _require_gpu(distances, demand)
        assert local_search in (None, "hgs")
        self.local_search, self.ls_ants, self.inference = local_search, int(ls_ants), inference
        self._ls_src = (distances.detach(), demand.detach())
        self._hgs = None
        self.distances = _f32c(distances)
        # float64 demands (cvrp_nls/utils.py:12-26) keep the load bookkeeping of the construction in double, as cvrp_sample does
        self.demand = demand.contiguous() if demand.dtype == torch.float64 else _f32c(demand)
        self.B, self.n = distances.shape[0], distances.shape[1]
        self.n_ants, self.decay, self.alpha, self.beta, self.capacity = n_ants, decay, alpha, beta, capacity
        self.elitist, self.min_max = elitist, min_max
        if min_max:
            self.min = 0.1 if min is None else min
            self.max = None
        self.pheromone = torch.ones_like(self.distances) if pheromone is None else _f32c(pheromone).clone()
        if min_max and pheromone is None:
            self.pheromone = self.pheromone * self.min
        self.heuristic = (1 / self.distances) if heuristic is None else heuristic
        self._own_heuristic = self.heuristic if heuristic is None else None        # (derived here from the f32 distances)
        self.lowest_cost = torch.full((self.B,), float("inf"), device=distances.device)
        self.shortest_path = None               # [B, Lmax] int64 once a step has run (zero-padded routes)
        self.sampler, self.iteration, self.ant_gid0 = sampler, 0, ant_gid0
        self.seed = torch.initial_seed() if seed is None else seed

    @torch.no_grad()
    def step(self, Lmax=None, trim=False, events=None):
        """One colony iteration without a host round trip: the sampler also produces the route costs and the
        successor table the directed deposit consumes.  Returns (paths [B, Lmax, A], costs [B, A]); rows past
        an ant's route are 0 (self.last_lens holds the used rows; trim=True cuts to their maximum, which syncs).
        events: see cvrp_sample."""
        paths, _, _, lens, flags, costs, table = cvrp_sample(
            self.pheromone, self.heuristic, self.demand, self.capacity, self.n_ants, self.alpha, self.beta,
            mode=self.sampler, seed=self.seed, it=self.iteration, ant_gid0=self.ant_gid0, Lmax=Lmax, batch=self.B,
            dist=self.distances, want_table=True, events=events)
        self.iteration += 1
        self.last_lens, self.last_flags = lens, flags
        if self.local_search == "hgs":
            if self._hgs is None or self._hgs[3] is not self.heuristic:
                # the heuristic the colony holds NOW (cvrp_nls/aco.py:128-132 reads self.heuristic at first use; an assignment
                # after construction counts, and a later one rebuilds the perturbation tables -- ADVICE r5).  The colony's own
                # 1 / distances is taken from the distances in the dtype they were passed in (float64 instance data).
                d_src, dem_src = self._ls_src
                heu = (1 / d_src) if self.heuristic is self._own_heuristic else self.heuristic.detach()
                hd = 1 / (heu / heu.max(-1, keepdim=True).values + 1e-5)
                td = self._hgs[0] if self._hgs is not None else HgsTables(d_src)
                self._hgs = (td, HgsTables(hd), dem_src.double() / float(self.capacity), self.heuristic)
            td, th, dem_n, _ = self._hgs
            k = min(self.ls_ants, self.n_ants)
            idx = costs.topk(k, dim=1, largest=False).indices                               # [B, k]
            gi = idx.unsqueeze(1).expand(self.B, paths.shape[1], k)
            work = paths.gather(2, gi).contiguous()
            limit = 100000 if self.inference else max(self.n, 50)
            hgs_local_search_(work, [(td, limit), (th, 10), (td, limit)], dem_n)
            paths.scatter_(2, gi, work)
            costs.scatter_(1, idx, tour_costs(self.distances, work, closed=False))
            # the rewritten columns' used rows: up to and including the depot after their last client
            rows = torch.arange(1, work.shape[1] + 1, device=work.device, dtype=lens.dtype).view(1, -1, 1)
            lens.scatter_(1, idx, ((work != 0) * rows).amax(dim=1) + 1)
            table = None                                                                    # (the deposit rebuilds it from the paths)
        if self.shortest_path is None or self.shortest_path.shape[1] != paths.shape[1]:
            self.shortest_path = torch.zeros((self.B, paths.shape[1]), dtype=torch.int64, device=paths.device)
        new_max = track_best_(costs, paths, self.lowest_cost, self.shortest_path,
                              mmas_scale=self.n if self.min_max else None)
        cmin = cmax = None
        if self.min_max:
            if self.max is None:
                self.pheromone *= (new_max / self.pheromone.amax(dim=(1, 2))).view(self.B, 1, 1)
            self.max = new_max
            cmin, cmax = torch.full_like(new_max, self.min), new_max.contiguous()
        pheromone_update_(self.pheromone, paths, costs, self.decay, self.elitist, False, cmin, cmax, floor=1e-10,
                          nbr=table)
        if trim:
            paths = paths[:, :int(lens.max())].contiguous()
        return paths, costs

    def check_feasible(self):
        """Raise like the reference's Categorical if any draw of the last step had no feasible candidate (syncs)."""
        fl = int(self.last_flags.max())
        if fl & 1:
            raise ValueError("BatchedCVRP: a transition row had no feasible candidate")
        if fl & 2:
            raise RuntimeError("BatchedCVRP: route buffer too short")

    @torch.no_grad()
    def run(self, n_iterations):
        for _ in range(n_iterations):
            self.step()
        return self.lowest_cost


def ant_sharded_tsp(distances, n_ants, rank, world, decay=0.9, alpha=1.0, beta=1.0, heuristic=None, sampler="scan",
                    seed=0, exchange="delta", elitist=False, min_max=False, min=None, head_k=None, local_search=None,
                    inference=False, fixed_start=-1):
    """Ant-sharded TSP colony on this rank's GPU: A/world ants of every instance here, pheromone
    replicated, one collective per iteration (RCCL over xGMI when the process group is nccl):
    exchange="delta": all-reduce of the deposits [B,n,n]; exchange="tours": all-gather of the tours (int16) and
    costs, every rank applies the full deposit in ant order -- bit-identical to BatchedTSP with the same seed, for AS,
    elitist and MMAS colonies, best tours (`shortest_path`) included.
    sampler: as BatchedTSP ("auto" / "scan_sparse": head / tail rows, head_k entries per row as after sparsify(head_k));
    local_search "2opt" | "nls" (tsp_nls/aco.py:105-129): every rank improves ITS ants' tours before the exchange (the search
    of a tour depends on that tour only, so the colony stays the single-GPU one).
    Returns a parallel.AntShardedColony whose kernels are the HIP ones."""
    from .parallel import AntShardedColony
    _require_gpu(distances)
    dist_ = _f32c(distances)
    B, n, _ = dist_.shape
    eta = (1 / dist_) if heuristic is None else heuristic
    state, cache = {}, {}
    exact = exchange == "tours"
    assert local_search in (None, "2opt", "nls")

    def sample_fn(tau, lo, n_local, it):
        # colony-wide ant ids in both modes (ant lo + a of instance b is ant b*A + lo + a of the single-GPU colony):
        # rank-local ids would overlap when n_ants % world != 0 (shard_range gives the first ranks one ant more)
        want_nbr = not exact and not elitist and local_search is None
        mode, hk = resolve_sampler(sampler, n, head_k, eta, cache)
        if mode == "scan_sparse":
            if cache.get("head") is None or cache["head"][0] != hk:
                h = eta.detach()
                h = h if h.dim() == 3 else h.unsqueeze(0).expand(B, n, n)
                cache["head"] = (hk, sparse_head(_f32c(h), hk, top=take_auto_top(cache, eta)))     # (the top values are handed over once)
            paths, _, costs, nbr = tsp_sample_sparse(tau, eta, n_local, cache["head"][1], alpha, beta, seed=seed, it=it, ant_gid0=lo,
                                                     ant_gid_bstride=n_ants, fixed_start=fixed_start, batch=B, dist=dist_,
                                                     want_nbr=want_nbr)
        else:
            paths, _, _, _, costs, nbr = tsp_sample(tau, eta, n_local, alpha, beta, mode=mode, seed=seed, it=it,
                                                    ant_gid0=lo, ant_gid_bstride=n_ants, fixed_start=fixed_start, batch=B,
                                                    dist=dist_, want_nbr=want_nbr)
        if local_search is not None:
            tours = paths.permute(0, 2, 1).to(torch.int16).contiguous()
            maxt = 10000 if inference else n // 4
            if "dist_t" not in cache:
                cache["dist_t"] = transposed_for_two_opt(dist_)
                cache["tables"] = two_opt_tables(dist_, cache["dist_t"])
            if local_search == "2opt":
                two_opt_(dist_, tours, maxt, dist_t=cache["dist_t"], tables=cache["tables"])
                costs = None
            else:
                if "hd" not in cache:
                    h = eta.detach().float()
                    cache["hd"] = (1 / (h / h.amax(dim=-1, keepdim=True) + 1e-5)).contiguous()
                    cache["hd_t"] = transposed_for_two_opt(cache["hd"])
                    cache["htables"] = two_opt_tables(cache["hd"], cache["hd_t"])
                tours, costs = nls_(dist_, cache["hd"], tours, maxt, dist_t=cache["dist_t"], heuristic_dist_t=cache["hd_t"],
                                    tables=cache["tables"], heuristic_tables=cache["htables"], want_costs=True)
            paths = tours.permute(0, 2, 1).to(torch.int64).contiguous()
            if costs is None:
                costs = tour_costs(dist_, paths)
            nbr = None
        state["costs"], state["nbr"] = costs, nbr
        return paths

    def cost_fn(paths):
        return state["costs"]                     # fused into the sampler (or the local search)

    def deposit_fn(zero, paths, costs):
        return pheromone_update_(zero, paths, costs, 1.0, nbr=state["nbr"])

    def update_fn(tau, paths, costs, elit, cmin, cmax):
        return pheromone_update_(tau, paths, costs, decay, elit, True, cmin, cmax)

    return AntShardedColony(torch.ones_like(dist_), n_ants, decay, rank, world, sample_fn, cost_fn, deposit_fn,
                            exchange=exchange, update_fn=update_fn, elitist=elitist, min_max=min_max, min=min, problem_size=n)


def ant_sharded_cvrp(distances, demand, n_ants, rank, world, capacity=50, decay=0.9, alpha=1.0, beta=1.0, heuristic=None,
                     sampler="scan", seed=0, exchange="tours", elitist=False, min_max=False, min=None):
    """Ant-sharded CVRP colony (cvrp/aco.py:67-130 per instance: directed deposits tau[path[:-1], path[1:]] += 1/cost, floor
    1e-10): A/world ants of every instance on this rank, the route sequences ([B, 2n+1, A_local], zero padded) exchanged as
    int16 (exchange="tours": the single-GPU BatchedCVRP bit for bit) or the directed deposits summed (exchange="delta")."""
    from .parallel import AntShardedColony
    _require_gpu(distances, demand)
    dist_ = _f32c(distances)
    B, n, _ = dist_.shape
    eta = (1 / dist_) if heuristic is None else heuristic
    state = {}
    exact = exchange == "tours"

    def sample_fn(tau, lo, n_local, it):
        paths, _, _, lens, flags, costs, table = cvrp_sample(tau, eta, demand, capacity, n_local, alpha, beta, mode=sampler,
                                                             seed=seed, it=it, ant_gid0=lo, ant_gid_bstride=n_ants, batch=B,
                                                             dist=dist_, want_table=not exact and not elitist)
        state["costs"], state["table"], state["flags"] = costs, table, flags
        return paths

    def cost_fn(paths):
        return state["costs"]

    def deposit_fn(zero, paths, costs):
        return pheromone_update_(zero, paths, costs, 1.0, False, False, nbr=state["table"])

    def update_fn(tau, paths, costs, elit, cmin, cmax):
        return pheromone_update_(tau, paths, costs, decay, elit, False, cmin, cmax, floor=1e-10)

    return AntShardedColony(torch.ones_like(dist_), n_ants, decay, rank, world, sample_fn, cost_fn, deposit_fn,
                            exchange=exchange, update_fn=update_fn, elitist=elitist, min_max=min_max, min=min, problem_size=n,
                            floor=1e-10)
