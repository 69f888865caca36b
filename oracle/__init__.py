"""ctypes front-end of the CPU oracle (oracle/daco_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product package (deepaco_amd) never imports it and has no CPU fallback.

All functions take / return numpy arrays in the reference's layouts
(paths: (n, A) int64; log_probs: (n-1, A) float32; costs: (A,) float32).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ORC_INFEASIBLE = 1


def build(force=False):
    so = os.path.join(_HERE, "libdaco_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("daco_oracle.c", "hgs_ls.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libdaco_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_u01.restype = C.c_float
        _LIB.orc_neg_log2_1m.restype = C.c_float
        _LIB.orc_neg_log2_1m.argtypes = [C.c_float]
        _LIB.orc_two_opt_once.restype = C.c_float
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def vec_for_n(n):
    return lib().orc_vec_for_n(int(n))


def ld_for_n(n):
    return lib().orc_ld_for_n(int(n))


def philox4x32_10(ctr, key):
    ctr = np.ascontiguousarray(ctr, dtype=np.uint32)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_p(ctr), _p(key), _p(out))
    return out


def u01(x):
    return float(lib().orc_u01(C.c_uint32(int(x))))


def neg_log2_1m(w):
    return float(lib().orc_neg_log2_1m(C.c_float(float(w))))


def prob_matrix(tau, eta, alpha=1.0, beta=1.0):
    tau, eta = _f32(tau), _f32(eta)
    n = tau.shape[0]
    P = np.empty((n, n), dtype=np.float32)
    lib().orc_prob_matrix(n, _p(tau), _p(eta), C.c_float(alpha), C.c_float(beta), _p(P))
    return P


def tsp_sample_noise(P, start, noise, norm_passes=1, require_prob=True):
    P, noise, start = _f32(P), _f32(noise), _i64(start)
    n, A = P.shape[0], start.shape[0]
    assert noise.shape == (n - 1, A, n)
    paths = np.zeros((n, A), dtype=np.int64)
    logp = np.zeros((n - 1, A), dtype=np.float32) if require_prob else None
    rc = lib().orc_tsp_sample_noise(n, A, _p(P), _p(start), _p(noise), int(norm_passes), _p(paths),
                                    _p(logp) if require_prob else None)
    return paths, logp, rc


def _sample_rng(fn, P, A, seed, it, ant_gid0, fixed_start, require_prob):
    P = _f32(P)
    n = P.shape[0]
    paths = np.zeros((n, A), dtype=np.int64)
    logp = np.zeros((n - 1, A), dtype=np.float32) if require_prob else None
    rc = fn(n, A, _p(P), C.c_uint64(seed), C.c_uint64(it), C.c_uint32(ant_gid0), int(fixed_start),
            _p(paths), _p(logp) if require_prob else None)
    return paths, logp, rc


def tsp_sample_race(P, A, seed, it=0, ant_gid0=0, fixed_start=-1, require_prob=False):
    return _sample_rng(lib().orc_tsp_sample_race, P, A, seed, it, ant_gid0, fixed_start, require_prob)


def tsp_sample_scan(P, A, seed, it=0, ant_gid0=0, fixed_start=-1, require_prob=False, wave=False):
    """DACO_SCAN (four / two ants per wavefront for n <= 256 / <= 1024) or, with wave=True, DACO_SCAN_WAVE."""
    fn = lib().orc_tsp_sample_scan_wave if wave else lib().orc_tsp_sample_scan
    return _sample_rng(fn, P, A, seed, it, ant_gid0, fixed_start, require_prob)


def sparse_head_ids(weights, k):
    """Head of every row for scan_sparse: the k (<= 127) largest entries of `weights` [n][n] (the colony passes its heuristic),
    ids ascending; returns (head_id [n][kh] uint16, head_cnt [n] uint8) with kh = 64 slots for k <= 63, else 128.  Ties at the
    k-th value: the smaller id."""
    w = np.asarray(weights, dtype=np.float64)
    n = w.shape[0]
    assert 1 <= k <= 127 and k <= n
    kh = 64 if k <= 63 else 128
    order = np.lexsort((np.arange(n)[None, :].repeat(n, 0), -w), axis=1)[:, :k]          # by value descending, then id
    ids = np.zeros((n, kh), dtype=np.uint16)
    ids[:, :k] = np.sort(order, axis=1)
    return ids, np.full(n, k, dtype=np.uint8)


def sparse_head_values(P, head_id, head_cnt):
    """head_val [n][kh] f32: P at the head ids, +0 in the empty slots, the tail total T_i in slot kh - 1."""
    P = _f32(P)
    n = P.shape[0]
    hid, cnt = np.ascontiguousarray(head_id, dtype=np.uint16), np.ascontiguousarray(head_cnt, dtype=np.uint8)
    kh = hid.shape[1]
    assert kh in (64, 128) and hid.shape == (n, kh) and cnt.shape == (n,) and int(cnt.max()) <= kh - 1
    out = np.zeros((n, kh), dtype=np.float32)
    lib().orc_sparse_head_values(n, _p(P), _p(hid), _p(cnt), _p(out), kh)
    return out


def tsp_sample_scan_sparse(P, head_id, head_cnt, A, seed, it=0, ant_gid0=0, fixed_start=-1):
    """scan_sparse (head / tail rows, daco_oracle.c).  Returns (paths [n][A], rc, stats = [dense steps, tail walks, rejections])."""
    P = _f32(P)
    n = P.shape[0]
    assert n > 128, "scan_sparse: the tail walk and the dense step use the vec-4 layout (n > 128)"
    hid, cnt = np.ascontiguousarray(head_id, dtype=np.uint16), np.ascontiguousarray(head_cnt, dtype=np.uint8)
    hval = sparse_head_values(P, hid, cnt)
    paths = np.zeros((n, A), dtype=np.int64)
    stats = np.zeros(3, dtype=np.int64)
    rc = lib().orc_tsp_sample_scan_sparse(n, A, _p(P), _p(hid), _p(cnt), _p(hval), int(hid.shape[1]), C.c_uint64(seed),
                                          C.c_uint64(it), C.c_uint32(ant_gid0), int(fixed_start), _p(paths), _p(stats))
    return paths, rc, stats


def sparse_rounding_picks():
    """Head draws of tsp_sample_scan_sparse so far (this process) that took the rounding branch."""
    fn = lib().orc_sparse_rounding_picks
    fn.restype = C.c_long
    return int(fn())


def tsp_sample_scan_injected(P, uniforms, fixed_start=0, wave=False, require_prob=False):
    """The scan draw with caller-supplied uniforms [n-1][A] (f32) instead of Philox."""
    P, u = _f32(P), _f32(uniforms)
    n, A = P.shape[0], u.shape[1]
    assert u.shape == (n - 1, A)
    paths = np.zeros((n, A), dtype=np.int64)
    logp = np.zeros((n - 1, A), dtype=np.float32) if require_prob else None
    rc = lib().orc_tsp_sample_scan_injected(n, A, _p(P), _p(u), int(wave), int(fixed_start), None, _p(paths),
                                            _p(logp) if require_prob else None)
    return paths, logp, rc


def tour_costs(dist, paths, closed=True):
    dist, paths = _f32(dist), _i64(paths)
    n, (length, A) = dist.shape[0], paths.shape
    costs = np.zeros(A, dtype=np.float32)
    lib().orc_tour_costs(n, length, A, _p(dist), _p(paths), int(closed), _p(costs))
    return costs


def pheromone_update_tsp(tau, paths, costs, decay, elitist=False, clamp_min=0.0, clamp_max=0.0):
    tau = _f32(tau).copy()
    paths, costs = _i64(paths), _f32(costs)
    n, A = paths.shape
    lib().orc_pheromone_update_tsp(n, A, _p(tau), _p(paths), _p(costs), C.c_float(decay), int(elitist),
                                   C.c_float(clamp_min), C.c_float(clamp_max))
    return tau


def pheromone_update_cvrp(tau, paths, costs, decay, elitist=False, clamp_min=0.0, clamp_max=0.0):
    tau = _f32(tau).copy()
    paths, costs = _i64(paths), _f32(costs)
    n = tau.shape[0]
    length, A = paths.shape
    lib().orc_pheromone_update_cvrp(n, length, A, _p(tau), _p(paths), _p(costs), C.c_float(decay),
                                    int(elitist), C.c_float(clamp_min), C.c_float(clamp_max))
    return tau


def two_opt_once(dist, tour):
    dist = _f32(dist)
    t = np.ascontiguousarray(tour, dtype=np.uint16).copy()
    delta = lib().orc_two_opt_once(dist.shape[0], _p(dist), _p(t))
    return t, float(delta)


def two_opt_batch(dist, tours, max_iterations=1000):
    dist = _f32(dist)
    t = np.ascontiguousarray(tours, dtype=np.uint16).copy()
    T, n = t.shape
    sweeps = np.zeros(T, dtype=np.int32)
    lib().orc_two_opt_batch(n, T, _p(dist), _p(t), C.c_long(int(max_iterations)), _p(sweeps))
    return t, sweeps


def nls_batch(dist, heuristic_dist, tours, maxt, T_nls=10, T_p=20):
    """tsp_nls/aco.py:241-258 (ACO.nls) over orc_two_opt_batch: 2-opt, then T_nls x {T_p sweeps on the perturbation
    matrix, 2-opt, keep the tour if shorter}.  Lengths summed like tour_costs.  Returns (tours, total sweeps)."""
    def lengths(t):
        return tour_costs(dist, np.ascontiguousarray(t.T.astype(np.int64)), closed=True)
    best, sw = two_opt_batch(dist, tours, maxt)
    total = int(sw.sum())
    best_costs = lengths(best)
    new = best
    for _ in range(T_nls):
        pert, s1 = two_opt_batch(heuristic_dist, new, T_p)
        new, s2 = two_opt_batch(dist, pert, maxt)
        total += int(s1.sum()) + int(s2.sum())
        c = lengths(new)
        better = c < best_costs
        best = np.where(better[:, None], new, best)
        best_costs = np.where(better, c, best_costs)
    return best, total


def roulette_route(probmat, uniforms, start=0):
    probmat = _f32(probmat)
    uniforms = np.ascontiguousarray(uniforms, dtype=np.float64)
    n = probmat.shape[0]
    route = np.zeros(n, dtype=np.uint16)
    lib().orc_roulette_route(n, _p(probmat), _p(uniforms), int(start), _p(route))
    return route


def cvrp_sample_noise(P, demand, capacity, noise, Lmax=None, require_prob=True):
    """recorded-noise draw (the reference's torch.multinomial arithmetic).  A float64 `demand` decides the capacity mask in
    double, as cvrp_nls/aco.py:254-272 does on that directory's float64 data."""
    f64 = np.asarray(demand).dtype == np.float64
    demand64 = np.ascontiguousarray(demand, dtype=np.float64) if f64 else None
    P, demand, noise = _f32(P), _f32(demand), _f32(noise)
    n1 = P.shape[0]
    steps, A = noise.shape[0], noise.shape[1]
    Lmax = Lmax or 2 * n1 + 1
    paths = np.zeros((Lmax, A), dtype=np.int64)
    logp = np.zeros((Lmax - 1, A), dtype=np.float32) if require_prob else None
    if f64:
        L = lib().orc_cvrp_sample64(0, n1, A, _p(P), _p(demand), _p(demand64), C.c_double(capacity), _p(noise), steps,
                                    C.c_uint64(0), C.c_uint64(0), C.c_uint32(0), Lmax, _p(paths), _p(logp) if require_prob else None)
    else:
        L = lib().orc_cvrp_sample_noise(n1, A, _p(P), _p(demand), C.c_float(capacity), _p(noise), steps,
                                        Lmax, _p(paths), _p(logp) if require_prob else None)
    if L < 0:
        return None, None, L
    return paths[:L], (logp[:L - 1] if require_prob else None), L


def cvrp_sample_rng(P, demand, capacity, A, mode, seed, it=0, ant_gid0=0, Lmax=None, require_prob=False):
    """mode: 'race' or 'scan' (Philox); returns (paths[:L], logp[:L-1] | None, L).  A float64 `demand` (cvrp_nls/ keeps its
    data in double) decides the capacity mask in double, cvrp_nls/aco.py:254-272."""
    f64 = np.asarray(demand).dtype == np.float64
    demand64 = np.ascontiguousarray(demand, dtype=np.float64) if f64 else None
    P, demand = _f32(P), _f32(demand)
    n1 = P.shape[0]
    Lmax = Lmax or 2 * n1 + 1
    paths = np.zeros((Lmax, A), dtype=np.int64)
    logp = np.zeros((Lmax - 1, A), dtype=np.float32) if require_prob else None
    m = {"race": 1, "scan": 2, "scan_wave": 3}[mode]
    if f64:
        L = lib().orc_cvrp_sample64(m, n1, A, _p(P), _p(demand), _p(demand64), C.c_double(capacity), None, 0, C.c_uint64(seed),
                                    C.c_uint64(it), C.c_uint32(ant_gid0), Lmax, _p(paths),
                                    _p(logp) if require_prob else None)
    else:
        L = lib().orc_cvrp_sample(m, n1, A, _p(P), _p(demand), C.c_float(capacity), None, 0, C.c_uint64(seed),
                                  C.c_uint64(it), C.c_uint32(ant_gid0), Lmax, _p(paths),
                                  _p(logp) if require_prob else None)
    if L < 0:
        return None, None, L
    return paths[:L], (logp[:L - 1] if require_prob else None), L


def pheromone_update_directed(tau, paths, costs, decay, weights=None, elitist=False, clamp_min=0.0, clamp_max=0.0,
                              floor=0.0):
    tau = _f32(tau).copy()
    paths, costs = _i64(paths), _f32(costs)
    n = tau.shape[0]
    length, A = paths.shape
    w = _f32(weights) if weights is not None else None
    lib().orc_pheromone_update_directed(n, length, A, _p(tau), _p(paths), _p(costs), _p(w) if w is not None else None,
                                        C.c_float(decay), int(elitist), C.c_float(clamp_min), C.c_float(clamp_max),
                                        C.c_float(floor))
    return tau


def pick_move(P, prev, mask, mode="scan", noise=None, seed=0, it=0, ant_gid0=0, step=1, require_prob=True):
    P, mask, prev = _f32(P), _f32(mask), _i64(prev)
    n, A = P.shape[0], prev.shape[0]
    actions = np.zeros(A, dtype=np.int64)
    logp = np.zeros(A, dtype=np.float32) if require_prob else None
    m = 0 if noise is not None else {"race": 1, "scan": 2}[mode]
    nz = _f32(noise) if noise is not None else None
    rc = lib().orc_pick_move(m, n, A, _p(P), _p(prev), _p(mask), _p(nz) if nz is not None else None, C.c_uint64(seed),
                             C.c_uint64(it), C.c_uint32(ant_gid0), int(step), _p(actions),
                             _p(logp) if require_prob else None)
    return actions, logp, rc


# ---------------------------------------------------------------------------------------------- HGS local search
def hgs_local_search(positions, matrix, demands, seq, count, capacity=1000.001, demand_scale=1000.0, nb_granular=20, seed=1,
                     use_swap_star=False, out_len=None, want_stats=False):
    """One call of the reference's CVRP local search on one solution (oracle/hgs_ls.c: the restatement of
    HGS-CVRP-main/Program/LocalSearch.cpp as cvrp_nls/swapstar.py:324-346 drives it: demands * 1000, capacity 1000.001).
    positions [n,2], matrix [n,n], demands [n] float64 (depot first); seq: zero-separated node sequence (a column of
    `paths`).  Returns (sequence [out_len] int64 as merge_subroutes lays it out, status) -- status 1: HGS would have thrown and
    the reference keeps its input -- and, with want_stats, (moves, loops, rng draws, pairs evaluated)."""
    pos = np.ascontiguousarray(positions, dtype=np.float64)
    m = np.ascontiguousarray(matrix, dtype=np.float64)
    n = m.shape[0]
    xs, ys = np.ascontiguousarray(pos[:, 0]), np.ascontiguousarray(pos[:, 1])
    dem = np.ascontiguousarray(np.asarray(demands, dtype=np.float64) * demand_scale)
    s = np.ascontiguousarray(seq, dtype=np.int32)
    out_len = len(s) + 2 if out_len is None else out_len
    out = np.zeros(out_len, dtype=np.int32)
    st = np.zeros(4, dtype=np.int64)
    f = lib().hgs_local_search
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_int, C.c_int,
                  C.c_uint32, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    rc = f(n, _p(xs), _p(ys), _p(m), _p(dem), float(capacity), _p(s), len(s), int(count), int(nb_granular), int(seed),
           int(bool(use_swap_star)), _p(out), out_len, _p(st))
    if rc < 0:
        raise ValueError("hgs_local_search: bad argument")
    return (out.astype(np.int64), rc, st.tolist()) if want_stats else (out.astype(np.int64), rc)


def hgs_neural_swapstar(positions, distances, heuristic_dist, demands, seq, limit, disturb=10, **kw):
    """cvrp_nls/aco.py:443-448: search on the distances, `disturb` loops on the heuristic-derived matrix, search again."""
    L = len(seq) + 2
    s1, _ = hgs_local_search(positions, distances, demands, seq, limit, out_len=L, **kw)
    s2, _ = hgs_local_search(positions, heuristic_dist, demands, s1, disturb, out_len=L, **kw)
    s3, _ = hgs_local_search(positions, distances, demands, s2, limit, out_len=L, **kw)
    return s3


def hgs_correlated(matrix, nb_granular=20):
    """Params.cpp:80-103: (lists [n, n-1] int32 row i = ascending neighbours of client i, lens [n])."""
    m = np.ascontiguousarray(matrix, dtype=np.float64)
    n = m.shape[0]
    out = np.zeros((n, max(1, n - 1)), dtype=np.int32)
    ln = np.zeros(n, dtype=np.int32)
    f = lib().hgs_correlated
    f.restype = None
    f.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    f(n, _p(m), int(nb_granular), _p(out), _p(ln))
    return out, ln


def hgs_shuffle(v, seed=1, skip_draws=0):
    """std::shuffle(v, std::minstd_rand(seed) advanced by skip_draws) as libstdc++ does it; returns (shuffled, draws)."""
    a = np.ascontiguousarray(v, dtype=np.int32).copy()
    f = lib().hgs_shuffle
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int]
    d = f(_p(a), len(a), int(seed), int(skip_draws))
    return a, d
