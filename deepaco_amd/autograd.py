"""torch.autograd bridges: tour log-probabilities that carry gradient to the heuristic matrix.

The reference gets this for free from autograd through ~20 aten ops per step
(tsp/aco.py:154-176: mask.clone(), Categorical.log_prob).  Here the forward is one kernel
launch that also saves the row sums, and the backward is one launch of daco_sample_backward.
"""
import torch

from . import engine


class TspSampleFn(torch.autograd.Function):
    """(heuristic [n,n]) -> (paths [n,A], log_probs [n-1,A], flags [1]); grad flows to heuristic only."""

    @staticmethod
    def forward(ctx, heuristic, pheromone, n_ants, alpha, beta, mode, norm_passes, start, fixed_start, noise,
                seed, it):
        eta = heuristic.detach()
        paths, logp, rowsum, flags = engine.tsp_sample(
            pheromone, eta, n_ants, alpha, beta, mode=mode, norm_passes=norm_passes, start=start,
            fixed_start=fixed_start, noise=noise, seed=seed, it=it, require_prob=True, batch=1)
        ctx.save_for_backward(pheromone, eta, paths, rowsum)
        ctx.ab = (alpha, beta)
        ctx.mark_non_differentiable(paths, flags)
        return paths[0], logp[0], flags

    @staticmethod
    def backward(ctx, _gp, glogp, _gf):
        tau, eta, paths, rowsum = ctx.saved_tensors
        grad = engine.sample_backward(tau, eta, ctx.ab[0], ctx.ab[1], paths, rowsum, glogp.contiguous().unsqueeze(0))
        return (grad[0],) + (None,) * 11


class TspBatchSampleFn(torch.autograd.Function):
    """B colonies at once: (heuristic [B,n,n]) -> (paths [B,n,A], log_probs [B,n-1,A], flags [B]); one sampler launch
    forward, one daco_sample_backward launch backward (the batched tsp_nls/train.py step)."""

    @staticmethod
    def forward(ctx, heuristic, pheromone, n_ants, alpha, beta, mode, norm_passes, fixed_start, seed, it, iter_dev=None):
        # (iter_dev: int64 device scalar added to `it` by the kernel -- a captured training step advances it)
        eta = heuristic.detach().contiguous()
        B = eta.shape[0]
        paths, logp, rowsum, flags = engine.tsp_sample(
            pheromone, eta, n_ants, alpha, beta, mode=mode, norm_passes=norm_passes, fixed_start=fixed_start,
            seed=seed, it=it, require_prob=True, batch=B, iter_dev=iter_dev)
        ctx.save_for_backward(pheromone, eta, paths, rowsum)
        ctx.ab = (alpha, beta)
        ctx.mark_non_differentiable(paths, flags)
        return paths, logp, flags

    @staticmethod
    def backward(ctx, _gp, glogp, _gf):
        tau, eta, paths, rowsum = ctx.saved_tensors
        grad = engine.sample_backward(tau, eta, ctx.ab[0], ctx.ab[1], paths, rowsum, glogp.contiguous())
        return (grad,) + (None,) * 10


class CvrpSampleFn(torch.autograd.Function):
    """(heuristic [n,n]) -> (paths [Lmax,A], log_probs [Lmax-1,A], lens [A], flags [1])."""

    @staticmethod
    def forward(ctx, heuristic, pheromone, demand, capacity, n_ants, alpha, beta, mode, noise, seed, it):
        eta = heuristic.detach()
        paths, logp, rowsum, lens, flags = engine.cvrp_sample(
            pheromone, eta, demand, capacity, n_ants, alpha, beta, mode=mode, noise=noise, seed=seed, it=it,
            require_prob=True, batch=1)
        ctx.save_for_backward(pheromone, eta, paths, rowsum, lens, demand)
        ctx.misc = (alpha, beta, capacity)
        ctx.mark_non_differentiable(paths, lens, flags)
        return paths[0], logp[0], lens[0], flags

    @staticmethod
    def backward(ctx, _gp, glogp, _gl, _gf):
        tau, eta, paths, rowsum, lens, demand = ctx.saved_tensors
        a, b, cap = ctx.misc
        grad = engine.sample_backward(tau, eta, a, b, paths, rowsum, glogp.contiguous().unsqueeze(0), lens=lens,
                                      demand=demand, capacity=cap)
        return (grad[0],) + (None,) * 10


class SiblingSampleFn(torch.autograd.Function):
    """Fused sop / pctsp / op / mkp construction whose log-probabilities carry gradient to the heuristic
    (daco_sibling_sample forward, daco_sibling_backward = route replay with the problem's own feasibility rules)."""

    @staticmethod
    def forward(ctx, heuristic, pheromone, kind, n_ants, alpha, beta, mode, noise, seed, it, kw):
        eta = heuristic.detach().float()
        tau = pheromone.detach().float()
        paths, logp, rowsum, lens, flags = engine.sibling_sample(kind, tau, eta, n_ants, alpha, beta, mode=mode,
                                                                 noise=noise, seed=seed, it=it, require_prob=True, **kw)
        ctx.save_for_backward(tau, eta, paths, rowsum, lens if lens is not None else torch.empty(0))
        ctx.meta = (kind, alpha, beta, {k: v for k, v in kw.items() if k in ("aux_vec", "aux_mat", "scalar0", "item_weights")})
        ctx.mark_non_differentiable(paths, flags)
        out_lens = lens if lens is not None else torch.empty(0, dtype=torch.int32, device=paths.device)
        ctx.mark_non_differentiable(out_lens)
        return paths, logp, out_lens, flags

    @staticmethod
    def backward(ctx, _gp, glogp, _gl, _gf):
        tau, eta, paths, rowsum, lens = ctx.saved_tensors
        kind, alpha, beta, kw = ctx.meta
        grad = engine.sibling_backward(kind, tau, eta, alpha, beta, paths, rowsum, glogp.contiguous(),
                                       lens=lens if lens.numel() else None, **kw)
        grad = grad[0] if ctx.needs_input_grad[0] and grad.shape[0] == 1 and eta.dim() == 2 else grad
        return (grad,) + (None,) * 10
