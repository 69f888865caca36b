"""Closed-form gradient of the tour log-probabilities w.r.t. the heuristic (numpy, float64).
TEST INFRASTRUCTURE ONLY.

Restates what autograd computes through the reference's
Categorical(tau^a * eta^b * mask).log_prob(action) (tsp/aco.py:171-177, cvrp/aco.py:167-174):
    d log p / d eta_ik = b * ([k = j] / eta_ik - p_k / (eta_ik * S)),  zero when p_j/S is clamped.
Pinned against heu_mat.grad captured from the reference's REINFORCE loss (fixtures g3_grad_*).
"""
import numpy as np

EPS = np.float32(1.1920928955078125e-07)


def _pw(x, a):
    return x if a == 1 else (x * x if a == 2 else np.power(x, a))


def _dp_deta(tau_row, eta_row, alpha, beta, p):
    """d (tau^a eta^b) / d eta = b tau^a eta^(b-1): b p / eta where eta != 0; at eta == 0 autograd gives tau^a for
    b = 1 and 0 for b > 1 (not 0/0)."""
    safe = np.where(eta_row != 0, eta_row, 1.0)
    at_zero = _pw(tau_row, alpha) * (p == p) if beta == 1 else np.zeros_like(p)
    return np.where(eta_row != 0, beta * p / safe, at_zero)


def tsp_grad(tau, eta, alpha, beta, paths, grad_logp):
    n, A = paths.shape
    tau64, eta64 = tau.astype(np.float64), eta.astype(np.float64)
    out = np.zeros((n, n), np.float64)
    for a in range(A):
        open_ = np.ones(n, bool)
        prev = int(paths[0, a])
        open_[prev] = False
        for t in range(1, n):
            j = int(paths[t, a])
            p = _pw(tau64[prev], alpha) * _pw(eta64[prev], beta) * open_
            S = p.sum()
            pr = np.float32(p[j] / S)
            g = float(grad_logp[t - 1, a])
            if EPS < pr < np.float32(1) - EPS and g != 0.0:
                out[prev] -= g * _dp_deta(tau64[prev], eta64[prev], alpha, beta, p) * open_ / S
                out[prev, j] += g * beta / eta64[prev, j]
            open_[j] = False
            prev = j
    return out


def cvrp_grad(tau, eta, alpha, beta, demand, capacity, paths, grad_logp):
    L, A = paths.shape
    n = tau.shape[0]
    tau64, eta64 = tau.astype(np.float64), eta.astype(np.float64)
    out = np.zeros((n, n), np.float64)
    for a in range(A):
        vis = np.zeros(n, bool)
        prev, remaining, used = 0, n - 1, np.float32(0)
        for t in range(1, L):
            if remaining == 0 and prev == 0:
                break                                   # done: p(depot) = 1 is clamped, no gradient
            j = int(paths[t, a])
            open_ = ~vis
            open_[0] = not (prev == 0 and remaining > 0)
            open_ &= ~(demand > np.float32(capacity) - used)
            p = _pw(tau64[prev], alpha) * _pw(eta64[prev], beta) * open_
            S = p.sum()
            pr = np.float32(p[j] / S)
            g = float(grad_logp[t - 1, a])
            if EPS < pr < np.float32(1) - EPS and g != 0.0:
                out[prev] -= g * beta * p / (eta64[prev] * S)
                out[prev, j] += g * beta / eta64[prev, j]
            if j != 0:
                vis[j] = True
                remaining -= 1
            else:
                used = np.float32(0)
            used = np.float32(used + demand[j])
            prev = j
    return out
