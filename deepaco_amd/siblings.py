"""ACO for DeepACO's sibling problems (S1-S6 of SURVEY.md section 8) on MI355X.

Each class keeps the constructor, method names and return layouts of the reference's
<problem>/aco.py.  What runs on the GPU:

  * every draw -- the reference's `pick_move` / `pick_node` / `pick_item`
    (Categorical(tau[prev]^a * eta[prev]^b * masks).sample()/log_prob) -- is one launch of
    daco_pick_move over all ants (op/aco.py:186-193, pctsp/aco.py:157-164, sop/aco.py:156-169,
    smtwtp/aco.py:139-151, mkp/aco.py:147-154); BPP uses the fused CVRP kernel directly
    (bpp/aco.py:130-196 has cvrp's visit + capacity masks);
  * the pheromone update is the directed deposit kernel with the problem's own amount per ant
    (daco_pheromone_update with `weights` / `hub`).

The feasibility rules themselves stay on the caller's side of that boundary, as in the
reference, but vectorised over ants as torch ops on the device (the reference loops over ants
in Python for OP and MKP, op/aco.py:208-215, mkp/aco.py:174-181).  `sample()` returns
log-probabilities that carry gradient to the heuristic (one autograd.Function around the fused
construction; around each draw on the step-wise path).
Pass `_noise=[q_1, q_2, ...]` (the reference's recorded Exp(1) tensors) to reproduce the
reference's solutions bit for bit.
"""
import torch

from . import engine


class _PickFn(torch.autograd.Function):
    """One draw with gradient: d log p / d eta[prev][k] = b*([k=j]/eta - p_k m_k/(eta S))."""

    @staticmethod
    def forward(ctx, heuristic, svc, tau, prev, mask, step, noise, alpha, beta):
        act, logp, rowsum = svc.pick(prev, mask, step, require_prob=True, noise=noise)
        ctx.save_for_backward(heuristic.detach(), tau, prev, mask, act, rowsum, logp)
        ctx.ab = (alpha, beta)
        ctx.mark_non_differentiable(act)
        return act, logp

    @staticmethod
    def backward(ctx, _ga, glogp):
        eta, tau, prev, mask, act, S, logp = ctx.saved_tensors
        a, b = ctx.ab
        A = prev.shape[0]
        e_rows = eta[prev]
        p = (tau[prev] ** a) * (e_rows ** b) * mask
        eps = torch.finfo(torch.float32).eps
        inside = (logp > torch.log(torch.tensor(eps, device=logp.device))) & \
                 (logp < torch.log(torch.tensor(1 - eps, device=logp.device)))
        g = (glogp * inside).unsqueeze(1)
        # d p_k / d eta_k = b tau^a eta^(b-1): b p/eta where eta != 0; at eta == 0 the reference's autograd gives
        # tau^a (b = 1) or 0 (b > 1), not 0/0
        zero = e_rows == 0
        dp = torch.where(zero, (tau[prev] ** a) * mask if b == 1 else torch.zeros_like(p),
                         b * p / torch.where(zero, torch.ones_like(e_rows), e_rows))
        contrib = -g * dp / S.unsqueeze(1)
        contrib[torch.arange(A, device=prev.device), act] += (g.squeeze(1) * b) / e_rows[torch.arange(A), act]
        grad = torch.zeros_like(eta).index_add_(0, prev, contrib)
        return (grad,) + (None,) * 8


class _Base:
    """Shared plumbing: the draw service for one construction, MMAS clamp, directed deposit."""

    sampler = "scan"

    def _setup_common(self, n_ants, decay, alpha, beta, elitist, min_max, min, sampler, seed):
        self.n_ants, self.decay, self.alpha, self.beta = n_ants, decay, alpha, beta
        self.elitist, self.min_max = elitist, min_max
        if min_max:
            if min is not None:
                assert min > 1e-9
            else:
                min = 0.1
            self.min = min
            self.max = None
        self.sampler = sampler
        self.seed = torch.initial_seed() if seed is None else seed
        self._calls = 0

    def _require_gpu(self, *ts):
        for t in ts:
            if torch.is_tensor(t) and not t.is_cuda:
                raise engine._lib.DacoError(f"{type(self).__module__}.ACO needs tensors on a HIP device; there is no CPU path")

    def _begin(self):
        self._svc = engine.PickService(self.pheromone.detach().float(), self.heuristic.detach().float(), self.n_ants,
                                       self.alpha, self.beta, mode=self.sampler, seed=self.seed, it=self._calls)
        self._calls += 1
        self._grad = torch.is_grad_enabled() and self.heuristic.requires_grad

    def _pick(self, prev, mask, step, require_prob, noise):
        q = None if noise is None else noise[step - 1]
        if require_prob and self._grad:
            return _PickFn.apply(self.heuristic, self._svc, self.pheromone.detach().float(), prev, mask.clone(), step, q,
                                 self.alpha, self.beta)
        act, logp, _ = self._svc.pick(prev, mask, step, require_prob=require_prob, noise=q)
        return act, logp

    def _fused(self, kind, require_prob, noise, **kw):
        """One-launch construction (daco_sibling_sample).  With a heuristic that requires grad the log-probs come
        from autograd.SiblingSampleFn (backward = daco_sibling_backward, a replay of the routes; n <= 1024)."""
        it = self._calls
        self._calls += 1
        mode = "race_noise" if noise is not None else self.sampler
        nz = None if noise is None else torch.stack(list(noise)).unsqueeze(0)
        if self._wants_grad(require_prob):
            from .autograd import SiblingSampleFn
            paths, logp, lens, flags = SiblingSampleFn.apply(self.heuristic, self.pheromone, kind, self.n_ants, self.alpha,
                                                             self.beta, mode, nz, self.seed, it, kw)
            lens = lens if lens.numel() else None
        else:
            paths, logp, _, lens, flags = engine.sibling_sample(
                kind, self.pheromone.detach().float(), self.heuristic.detach().float(), self.n_ants, self.alpha, self.beta,
                mode=mode, noise=nz, seed=self.seed, it=it, require_prob=require_prob, **kw)
        fl = int(flags[0])
        if fl & 1:
            raise ValueError("a transition row had no feasible candidate")
        if fl & 2:
            raise RuntimeError("solution buffer / noise tensor too short")
        L = int(lens.max()) if lens is not None else paths.shape[1]
        if require_prob:
            return paths[0, :L], logp[0, :L - 1]
        return paths[0, :L]

    def _wants_grad(self, require_prob):
        return require_prob and torch.is_grad_enabled() and self.heuristic.requires_grad

    def check_feasible(self):
        if bool(self._svc.flags.any()):
            raise ValueError("a transition row had no feasible candidate")

    def _deposit(self, sols_LA, select_costs, weights, hub, floor=0.0):
        """sols_LA [L, A]; select_costs picks the elitist ant (first minimum); weights [A] deposit."""
        tau = self.pheromone.detach().to(torch.float32).clone().contiguous().unsqueeze(0)
        cmin = cmax = None
        if self.min_max:
            cmin = torch.full((1,), float(self.min), device=tau.device)
            cmax = torch.as_tensor(self.max, dtype=torch.float32, device=tau.device).reshape(1).contiguous()
        engine.pheromone_update_(tau, sols_LA.contiguous().unsqueeze(0), select_costs.float().unsqueeze(0), self.decay,
                                 self.elitist, False, cmin, cmax, floor=floor,
                                 weights=weights.float().contiguous().unsqueeze(0), hub=hub)
        self.pheromone = tau[0]


# =============================================================================== S4 SMTWTP
class SMTWTP(_Base):
    """smtwtp/aco.py:5-153: n jobs + dummy start node 0; cost = total weighted tardiness."""

    def __init__(self, due_time, weights, processing_time, n_ants=20, decay=0.9, alpha=1, beta=1, elitist=False,
                 min_max=False, pheromone=None, heuristic=None, min=None, device='cpu', *, sampler='scan', seed=None):
        self._require_gpu(due_time, weights, processing_time)
        self.n = len(due_time)
        self.due_time, self.weights, self.processing_time = due_time, weights, processing_time
        self._setup_common(n_ants, decay, alpha, beta, elitist, min_max, min, sampler, seed)
        dev = due_time.device
        if min_max:
            self.max = 1
        if pheromone is None:
            self.pheromone = torch.ones(size=(self.n + 1, self.n + 1), device=dev)
            if min_max:
                self.pheromone = self.pheromone * self.min
        else:
            self.pheromone = pheromone
        # jobs with an earlier due time are preferred (smtwtp/aco.py:50-52)
        self.heuristic = ((1 / torch.cat([torch.ones(1, device=dev), self.due_time])).repeat(self.n + 1, 1)
                          if heuristic is None else heuristic)
        self.best_sol, self.lowest_cost, self.device = None, float('inf'), dev

    def sample(self):
        paths, log_probs = self.gen_path(require_prob=True)
        return self.gen_path_costs(paths), log_probs

    @torch.no_grad()
    def run(self, n_iterations):
        for _ in range(n_iterations):
            paths = self.gen_path(require_prob=False)
            costs = self.gen_path_costs(paths)
            best_cost, best_idx = costs.min(dim=0)
            if best_cost < self.lowest_cost:
                self.best_sol, self.lowest_cost = paths[:, best_idx], best_cost
            self.update_pheronome(paths, costs)
        return self.lowest_cost

    @torch.no_grad()
    def update_pheronome(self, paths, costs):
        self._deposit(paths, costs, 1.0 / (costs + 1), hub=-1)          # smtwtp/aco.py:84-95

    @torch.no_grad()
    def gen_path_costs(self, paths):
        jobs = (paths - 1).T                                             # [A, n]
        finish = torch.cumsum(self.processing_time[jobs], dim=1)
        late = (finish - self.due_time[jobs]).clamp(min=0)
        return (self.weights[jobs] * late).sum(dim=1)

    def gen_path(self, require_prob=False, *, _noise=None, _stepwise=False):
        """A permutation of the jobs drawn after the dummy start node: exactly the fused TSP kernel with
        every ant starting at node 0 (one launch); `_stepwise=True` keeps the draw-by-draw service path
        (same Philox counters, hence the same sequences) for cross-checking."""
        if _stepwise:
            self._begin()
            A, dev = self.n_ants, self.device
            rows = torch.arange(A, device=dev)
            prev = torch.zeros(A, dtype=torch.long, device=dev)
            mask = torch.ones(A, self.n + 1, device=dev)
            mask[:, 0] = 0
            seq, lps = [], []
            for t in range(1, self.n + 1):
                act, lp = self._pick(prev, mask, t, require_prob, _noise)
                seq.append(act)
                lps.append(lp)
                mask[rows, act] = 0
                prev = act
            return (torch.stack(seq), torch.stack(lps)) if require_prob else torch.stack(seq)
        it = self._calls
        self._calls += 1
        mode = "race_noise" if _noise is not None else self.sampler
        if mode == "scan":
            mode = "scan_wave"          # the draw of the step-wise service: fused == step-wise for every n
        noise = None if _noise is None else torch.stack(list(_noise)).unsqueeze(0)
        tau = self.pheromone.detach().float()
        if require_prob and torch.is_grad_enabled() and self.heuristic.requires_grad:
            from .autograd import TspSampleFn
            paths, logp, _ = TspSampleFn.apply(self.heuristic, tau, self.n_ants, self.alpha, self.beta, mode, 1, None, 0,
                                               noise, self.seed, it)
            return paths[1:], logp
        paths, logp, _, _ = engine.tsp_sample(tau, self.heuristic.detach().float(), self.n_ants, self.alpha, self.beta,
                                              mode=mode, norm_passes=1, fixed_start=0, noise=noise, seed=self.seed,
                                              it=it, require_prob=require_prob, batch=1)
        return (paths[0, 1:], logp[0]) if require_prob else paths[0, 1:]


# =============================================================================== S3 SOP
class SOP(_Base):
    """sop/aco.py:4-180: Hamiltonian path from node 0 under precedence constraints
    (prec_cons[j, k] = 1: k must be visited before j)."""

    def __init__(self, distances, prec_cons, n_ants=20, decay=0.9, alpha=1, beta=1, elitist=False, min_max=False,
                 pheromone=None, heuristic=None, min=None, device='cpu', *, sampler='scan', seed=None):
        self._require_gpu(distances, prec_cons)
        self.problem_size = len(distances)
        self.distances, self.prec_cons = distances, prec_cons
        self._setup_common(n_ants, decay, alpha, beta, elitist, min_max, min, sampler, seed)
        if pheromone is None:
            self.pheromone = torch.ones_like(distances)
            if min_max:
                self.pheromone = self.pheromone * self.min
        else:
            self.pheromone = pheromone
        self.heuristic = 1 / distances if heuristic is None else heuristic
        self.shortest_path, self.lowest_cost, self.device = None, float('inf'), distances.device

    def sample(self):
        paths, log_probs = self.gen_path(require_prob=True)
        return self.gen_path_costs(paths), log_probs

    @torch.no_grad()
    def run(self, n_iterations):
        for _ in range(n_iterations):
            paths = self.gen_path(require_prob=False)
            costs = self.gen_path_costs(paths)
            best_cost, best_idx = costs.min(dim=0)
            if best_cost < self.lowest_cost:
                self.shortest_path, self.lowest_cost = paths[:, best_idx], best_cost
                if self.min_max:
                    max = self.problem_size / self.lowest_cost
                    if self.max is None:
                        self.pheromone *= max / self.pheromone.max()
                    self.max = max
            self.update_pheronome(paths, costs)
        return self.lowest_cost

    @torch.no_grad()
    def update_pheronome(self, paths, costs):
        self._deposit(paths, costs, 1.0 / costs, hub=-1)

    @torch.no_grad()
    def gen_path_costs(self, paths):
        assert paths.shape == (self.problem_size, self.n_ants)
        return engine.tour_costs(self.distances, paths.contiguous().unsqueeze(0), closed=False)[0]

    def gen_path(self, require_prob=False, *, _noise=None, _stepwise=False):
        if not _stepwise and not (self._wants_grad(require_prob) and self.heuristic.shape[-1] > 1024):
            prec = self.prec_cons.float()
            return self._fused("sop", require_prob, _noise, aux_vec=prec.sum(dim=1), aux_mat=prec.T.contiguous())
        self._begin()
        A, n, dev = self.n_ants, self.problem_size, self.device
        rows = torch.arange(A, device=dev)
        before = self.prec_cons.float().T.contiguous()           # before[k] = who waits for k
        pending = self.prec_cons.float().sum(dim=1).repeat(A, 1)  # unvisited predecessors per node
        prev = torch.zeros(A, dtype=torch.long, device=dev)
        pending = pending - before[prev]
        visit = torch.ones(A, n, device=dev)
        visit[:, 0] = 0
        seq, lps = [prev], []
        for t in range(1, n):
            act, lp = self._pick(prev, visit * (pending == 0), t, require_prob, _noise)
            seq.append(act)
            lps.append(lp)
            pending = pending - before[act]
            visit[rows, act] = 0
            prev = act
        return (torch.stack(seq), torch.stack(lps)) if require_prob else torch.stack(seq)


# =============================================================================== S2 PCTSP
class PCTSP(_Base):
    """pctsp/aco.py:6-188: leave the depot, collect at least n/4 prize, return; pay penalties for
    unvisited nodes."""

    def __init__(self, distances, prizes, penalties, n_ants=20, decay=0.9, alpha=1, beta=1, elitist=False,
                 min_max=False, pheromone=None, heuristic=None, min=None, device='cpu', *, sampler='scan', seed=None):
        self._require_gpu(distances, prizes, penalties)
        self.n = prizes.size(0)
        self.distances, self.prizes, self.penalties = distances, prizes, penalties
        self.min_prizes = self.n / 4
        self._setup_common(n_ants, decay, alpha, beta, elitist, min_max, min, sampler, seed)
        if pheromone is None:
            self.pheromone = torch.ones_like(distances)
            if min_max:
                self.pheromone = self.pheromone * self.min
        else:
            self.pheromone = pheromone
        if heuristic is None:
            d = distances.clone()
            d.fill_diagonal_(1e9)
            self.heuristic = (1e-10 + prizes.repeat(self.n, 1)) / d
        else:
            self.heuristic = heuristic
        self.alltime_best_obj, self.alltime_best_sol, self.device = 1e10, None, distances.device

    def sample(self):
        sols, log_probs = self.gen_sol(require_prob=True)
        return self.gen_sol_obj(sols), log_probs

    @torch.no_grad()
    def run(self, n_iterations):
        for _ in range(n_iterations):
            sols = self.gen_sol(require_prob=False)
            objs = self.gen_sol_obj(sols)
            sols = sols.T
            best_obj, best_idx = objs.max(dim=0)                 # (sic) pctsp/aco.py:73
            if best_obj < self.alltime_best_obj:
                self.alltime_best_obj, self.alltime_best_sol = best_obj, sols[best_idx]
                if self.min_max:
                    max = (self.n - 1) / self.alltime_best_obj
                    if self.max is None:
                        self.pheromone *= max / self.pheromone.max()
                    self.max = max
            self.update_pheronome(sols, objs, best_obj, best_idx)
        return self.alltime_best_obj, self.alltime_best_sol

    @torch.no_grad()
    def update_pheronome(self, sols, objs, best_obj, best_idx):
        # the elitist ant is run()'s arg-MAX of the objective (pctsp/aco.py:73,88-90)
        self._deposit(sols.T, -objs, 1.0 / objs, hub=0)

    @torch.no_grad()
    def gen_sol_obj(self, solutions):
        length = engine.tour_costs(self.distances, solutions.contiguous().unsqueeze(0), closed=False)[0]
        seen = torch.zeros(self.n_ants, self.n, device=self.device).scatter_(1, solutions.T, 1.0)
        return length + ((1 - seen) * self.penalties).sum(dim=1)

    def gen_sol(self, require_prob=False, *, _noise=None, _stepwise=False):
        if not _stepwise and not (self._wants_grad(require_prob) and self.heuristic.shape[-1] > 1024):
            return self._fused("pctsp", require_prob, _noise, aux_vec=self.prizes.float(), scalar0=self.min_prizes)
        self._begin()
        A, n, dev = self.n_ants, self.n, self.device
        rows = torch.arange(A, device=dev)
        cur = torch.zeros(A, dtype=torch.long, device=dev)
        visit = torch.ones(A, n, device=dev)
        depot = torch.ones(A, n, device=dev)
        depot[:, 0] = 0
        collected = torch.zeros(A, device=dev)
        seq, lps, t = [cur], [], 0
        while True:
            t += 1
            cur, lp = self._pick(cur, visit * depot, t, require_prob, _noise)
            seq.append(cur)
            lps.append(lp)
            collected = collected + self.prizes[cur]
            visit[rows, cur] = 0
            home = cur == 0
            visit[home, 0] = 1
            visit[home, 1:] = 0
            away = ~home
            depot[away & (collected > self.min_prizes), 0] = 1
            depot[away & (visit[:, 1:] == 0).all(dim=1), 0] = 1
            if bool(home.all()):
                break
        return (torch.stack(seq), torch.stack(lps)) if require_prob else torch.stack(seq)


# =============================================================================== S1 OP
class OP(_Base):
    """op/aco.py:5-224: orienteering -- maximise collected prize on a route from the depot whose
    length stays within max_len; a dummy end node n absorbs finished ants."""

    def __init__(self, distances, prizes, max_len, n_ants=20, decay=0.9, alpha=1, beta=1, elitist=False,
                 min_max=False, pheromone=None, heuristic=None, min=None, device='cpu', k_sparse=None, *,
                 sampler='scan', seed=None):
        self._require_gpu(distances, prizes)
        self.n = len(prizes)
        self.distances, self.prizes, self.max_len = distances, prizes, max_len
        self._setup_common(n_ants, decay, alpha, beta, elitist, min_max, min, sampler, seed)
        dev = self.device = distances.device
        self.Q = 1 / prizes.sum()
        self.alltime_best_sol, self.alltime_best_obj = None, 0
        if heuristic is None:
            assert k_sparse
            self.sparsify(k_sparse)
        else:
            self.heuristic = heuristic
        # dummy end node n (op/aco.py:65-85): reachable from everywhere at no cost, leads nowhere
        n = self.n
        self.prizes = torch.cat((self.prizes, torch.zeros(1, device=dev)))
        d = torch.cat((self.distances, torch.full((1, n), 1e10, device=dev)), dim=0)
        self.distances = torch.cat((d, torch.zeros(n + 1, 1, device=dev)), dim=1)
        h = torch.cat((self.heuristic, torch.zeros(1, n, device=dev)), dim=0)
        self.heuristic = torch.cat((h, torch.ones(n + 1, 1, device=dev)), dim=1)
        self.pheromone = torch.ones_like(self.distances)

    @torch.no_grad()
    def sparsify(self, k_sparse):
        _, idx = torch.topk(self.distances, k=k_sparse, dim=1, largest=False)
        sparse = torch.full_like(self.distances, 1e10)
        sparse.scatter_(1, idx, torch.gather(self.distances, 1, idx))
        self.heuristic = self.prizes.unsqueeze(0) / sparse

    def sample(self):
        sols, log_probs = self.gen_sol(require_prob=True)
        return self.gen_sol_obj(sols), log_probs

    @torch.no_grad()
    def run(self, n_iterations):
        for _ in range(n_iterations):
            sols = self.gen_sol(require_prob=False)
            objs = self.gen_sol_obj(sols)
            sols = sols.T
            best_obj, best_idx = objs.max(dim=0)
            if best_obj > self.alltime_best_obj:
                self.alltime_best_obj, self.alltime_best_sol = best_obj, sols[best_idx]
                if self.min_max:
                    max = self.alltime_best_obj * self.n * self.Q
                    if self.max is None:
                        self.pheromone *= max / self.pheromone.max()
                    self.max = max
            self.update_pheronome(sols, objs, best_obj, best_idx)
        return self.alltime_best_obj, self.alltime_best_sol

    @torch.no_grad()
    def update_pheronome(self, sols, objs, best_obj, best_idx):
        self._deposit(sols.T, -objs, self.Q * objs, hub=self.n)

    @torch.no_grad()
    def gen_sol_obj(self, solutions):
        return self.prizes[solutions.T].sum(dim=1)

    def _close(self, travel, cur, mask):
        """op/aco.py:195-220 for all ants at once: close the current node and every candidate from
        which the depot could not be reached within max_len; open the dummy when nothing is left."""
        A, n = self.n_ants, self.n
        mask[torch.arange(A, device=self.device), cur] = 0
        reach = travel.unsqueeze(1) + self.distances[cur] + self.distances[:, 0].unsqueeze(0)
        too_far = (reach > self.max_len) & (cur != n).unsqueeze(1)
        mask[too_far] = 0
        mask[:, -1] = 0
        mask[(mask[:, :-1] == 0).all(dim=1), -1] = 1
        return mask

    def gen_sol(self, require_prob=False, *, _noise=None, _stepwise=False):
        if not _stepwise and not (self._wants_grad(require_prob) and self.heuristic.shape[-1] > 1024):
            d = self.distances.float().contiguous()
            return self._fused("op", require_prob, _noise, aux_vec=d[:, 0].contiguous(), aux_mat=d,
                               scalar0=float(self.max_len))
        self._begin()
        A, n, dev = self.n_ants, self.n, self.device
        cur = torch.zeros(A, dtype=torch.long, device=dev)
        travel = torch.zeros(A, device=dev)
        mask = self._close(travel, cur, torch.ones(A, n + 1, device=dev))
        seq, lps, t = [cur], [], 0
        while not bool((mask[:, :-1] == 0).all()):
            t += 1
            nxt, lp = self._pick(cur, mask, t, require_prob, _noise)
            seq.append(nxt)
            lps.append(lp)
            travel = travel + self.distances[cur, nxt]
            cur = nxt
            mask = self._close(travel, cur, mask.clone() if require_prob else mask)
        return (torch.stack(seq), torch.stack(lps)) if require_prob else torch.stack(seq)


# =============================================================================== S5 BPP
class BPP(_Base):
    """bpp/aco.py:42-200: bin packing as a CVRP-like sequence (node 0 opens a new bin); the
    construction is the fused CVRP kernel, the fitness is Levine & Ducatelle's."""

    def __init__(self, demand, n_ants=20, decay=0.9, alpha=1, beta=1, elitist=False, pheromone=None, heuristic=None,
                 device='cpu', capacity=150, *, sampler='scan', seed=None):
        self._require_gpu(demand)
        self.problem_size = len(demand)
        self.capacity, self.demand = capacity, demand
        self._setup_common(n_ants, decay, alpha, beta, elitist, False, None, sampler, seed)
        dev = self.device = demand.device
        self.pheromone = torch.ones(self.problem_size, self.problem_size, device=dev) if pheromone is None else pheromone
        self.heuristic = demand.unsqueeze(0).repeat(len(demand), 1).float() if heuristic is None else heuristic
        self.heuristic[:, 0] = 1e-5
        self.shortest_path, self.best_fitness = None, 0

    def sample(self):
        paths, log_probs = self.gen_path(require_prob=True)
        return self.gen_path_costs(paths), log_probs

    @torch.no_grad()
    def run(self, n_iterations):
        for _ in range(n_iterations):
            paths = self.gen_path(require_prob=False)
            costs = self.gen_path_costs(paths)
            best_cost, best_idx = costs.min(dim=0)
            if -best_cost > self.best_fitness:
                self.shortest_path, self.best_fitness = paths[:, best_idx], -best_cost
            self.update_pheronome(paths, -costs)
        return self.best_fitness

    @torch.no_grad()
    def update_pheronome(self, paths, fits):
        # AS: every ant deposits fit/n_ants; elitist: the fittest deposits its fit (bpp/aco.py:109-118)
        w = fits if self.elitist else fits / self.n_ants
        self._deposit(paths, -fits, w, hub=0, floor=1e-10)

    @torch.no_grad()
    def gen_path_costs(self, paths):
        """-fitness, fitness = sum over bins (fill/C)^2 / #bins (bpp/aco.py:26-40,121-126), float64."""
        u = paths.T
        A, L = u.shape
        fill = self.demand.double()[u]
        bin_id = torch.cumsum((u == 0).long(), dim=1) - 1                 # bin index of every position
        n_open = int(bin_id.max()) + 1
        sums = torch.zeros(A, n_open, dtype=torch.float64, device=u.device).scatter_add_(1, bin_id, fill)
        f = ((sums / self.capacity) ** 2).sum(dim=1)
        tail = (torch.flip(u, [1]) != 0).long().argmax(dim=1)             # trailing zeros
        n_bins = L - tail - self.problem_size + 1
        return -(f / n_bins)

    def gen_path(self, require_prob=False, *, _noise=None):
        mode = "race_noise" if _noise is not None else self.sampler
        noise = None if _noise is None else torch.stack(list(_noise)).unsqueeze(0)
        it = self._calls
        self._calls += 1
        tau, eta = self.pheromone.detach().float(), self.heuristic
        if require_prob and torch.is_grad_enabled() and eta.requires_grad:
            from .autograd import CvrpSampleFn
            p_, lp_, lens, flags = CvrpSampleFn.apply(eta, tau, self.demand.float(), float(self.capacity), self.n_ants,
                                                      self.alpha, self.beta, mode, noise, self.seed, it)
            paths, logp, lens = p_.unsqueeze(0), lp_.unsqueeze(0), lens.unsqueeze(0)
        else:
            paths, logp, _, lens, flags = engine.cvrp_sample(tau, eta.detach().float(), self.demand.float(),
                                                             self.capacity, self.n_ants, self.alpha, self.beta,
                                                             mode=mode, noise=noise, seed=self.seed, it=it,
                                                             require_prob=require_prob, batch=1)
        L = int(lens.max())
        if int(flags[0]):
            raise ValueError("ACO.gen_path: infeasible draw or route buffer too short")
        return (paths[0, :L], logp[0, :L - 1]) if require_prob else paths[0, :L]


# =============================================================================== S6 MKP
class MKP(_Base):
    """mkp/aco.py:5-183: multi-dimensional knapsack (every constraint normalised to n//2); items
    are added until nothing fits, then the ant moves to the dummy node n."""

    def __init__(self, prize, weight, n_ants=20, decay=0.9, alpha=1, beta=1, elitist=False, min_max=False,
                 pheromone=None, heuristic=None, min=None, device='cpu', *, sampler='scan', seed=None):
        self._require_gpu(prize, weight)
        self.n, self.m = prize.size(0), weight.size(1)
        self._setup_common(n_ants, decay, alpha, beta, elitist, min_max, min, sampler, seed)
        dev = self.device = prize.device
        if min_max:
            self.max = 20
        if pheromone is None:
            self.pheromone = torch.ones(size=(self.n + 1, self.n + 1), device=dev)
            if min_max:
                self.pheromone = self.pheromone * self.min
        else:
            self.pheromone = pheromone
        heu = (prize / weight.sum(dim=1)).unsqueeze(0).repeat(self.n, 1) if heuristic is None else heuristic
        self.Q = 1 / prize.sum()
        self.alltime_best_sol, self.alltime_best_obj = None, 0
        # dummy node n (mkp/aco.py:60-64)
        self.prize = torch.cat((prize, torch.zeros(1, device=dev)))
        self.weight = torch.cat((weight, torch.zeros(1, self.m, device=dev)), dim=0)
        h = torch.cat((heu, torch.zeros(1, self.n, device=dev)), dim=0)
        self.heuristic = torch.cat((h, 1e-10 * torch.ones(self.n + 1, 1, device=dev)), dim=1)

    def sample(self):
        sols, log_probs = self.gen_sol(require_prob=True)
        return self.gen_sol_obj(sols), log_probs

    @torch.no_grad()
    def run(self, n_iterations):
        for _ in range(n_iterations):
            sols = self.gen_sol(require_prob=False)
            objs = self.gen_sol_obj(sols)
            sols = sols.T
            best_obj, best_idx = objs.max(dim=0)
            if best_obj > self.alltime_best_obj:
                self.alltime_best_obj, self.alltime_best_sol = best_obj, sols[best_idx]
            self.update_pheronome(sols, objs, best_obj.item(), best_idx.item())
        return self.alltime_best_obj, self.alltime_best_sol

    @torch.no_grad()
    def update_pheronome(self, sols, objs, best_obj, best_idx):
        self._deposit(sols.T, -objs, self.Q * objs, hub=self.n, floor=1e-10)

    @torch.no_grad()
    def gen_sol_obj(self, solutions):
        return self.prize[solutions.T].sum(dim=1)

    def _pack(self, mask, knapsack, items):
        """mkp/aco.py:163-183 for all ants at once."""
        A = self.n_ants
        mask[torch.arange(A, device=self.device), items] = 0
        knapsack = knapsack + self.weight[items]
        over = ((knapsack.unsqueeze(1) + self.weight.unsqueeze(0)) > self.n // 2).any(dim=2)     # [A, n+1]
        several = (mask != 0).sum(dim=1, keepdim=True) > 1
        mask[(mask != 0) & over & several] = 0
        mask[:, -1] = 1
        return mask, knapsack

    def gen_sol(self, require_prob=False, *, _noise=None, _start=None, _stepwise=False):
        if not _stepwise and not (self._wants_grad(require_prob) and self.heuristic.shape[-1] > 1024):
            return self._fused("mkp", require_prob, _noise, item_weights=self.weight.float(), scalar0=float(self.n // 2),
                               start=None if _start is None else _start.view(1, -1))
        self._begin()
        A, n, dev = self.n_ants, self.n, self.device
        if _start is not None:
            items = _start.to(dev)
        else:
            g = torch.Generator(device=dev).manual_seed((self.seed + 7919 * self._calls) % (2 ** 63))
            items = torch.randint(low=0, high=n, size=(A,), device=dev, generator=g)
        knapsack = torch.zeros(A, self.m, device=dev)
        mask, knapsack = self._pack(torch.ones(A, n + 1, device=dev), knapsack, items)
        dummy = torch.ones(A, n + 1, device=dev)
        dummy[:, -1] = 0
        dummy[(mask[:, :-1] == 0).all(dim=1)] = 1
        seq, lps, t = [items], [], 0
        while not bool((mask[:, :-1] == 0).all()):
            t += 1
            items, lp = self._pick(items, mask * dummy, t, require_prob, _noise)
            seq.append(items)
            lps.append(lp)
            mask, knapsack = self._pack(mask.clone() if require_prob else mask, knapsack, items)
            dummy = dummy.clone() if require_prob else dummy
            dummy[(mask[:, :-1] == 0).all(dim=1)] = 1
        return (torch.stack(seq), torch.stack(lps)) if require_prob else torch.stack(seq)
