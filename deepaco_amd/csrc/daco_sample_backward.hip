// daco_sample_backward.hip -- gradient of the tour log-probabilities w.r.t. the heuristic.
//
// Reference behaviour replaced: autograd through ACO.gen_path(require_prob=True)
// (tsp/aco.py:154-176, cvrp/aco.py:153-173: Categorical(dist).log_prob(actions) on
// dist = tau^alpha * eta^beta * mask), as used by the REINFORCE loss of the training scripts
// (tsp/train.ipynb:45-49, tsp_nls/train.py:31-44, cvrp/train.ipynb:45-51).
//
// log p_t = log clamp(p_j / S, eps, 1-eps), p_k = tau_ik^a * eta_ik^b * m_k, i = prev, j = action:
//   d log p_t / d eta_ik = b * ( [k = j] / eta_ik  -  p_k / (eta_ik * S) )      (0 when clamped)
// One wavefront per (instance, ant) replays its route (the same lane layout, visited bitset and
// CVRP capacity bookkeeping as the forward kernel), recomputes p_k from tau/eta, and scatters
// g_t * d log p_t / d eta into grad_eta[i][.] with hardware f32 atomics (rows are shared by
// all ants passing through node i).  S is the row sum the forward kernel saved.  Summation
// order across ants is not fixed: the result is reproducible to rounding (1e-6 relative), which
// is inside the 1e-5 tolerance the parity tests use for gradients.
#include <type_traits>
#include <utility>

#include "daco_device.h"
#include "../../include/deepaco_hip.h"

namespace daco {

struct BackwardParams {
  int B, n, A, rows;             // rows of paths per instance (n for TSP, Lmax for CVRP)
  const float *tau, *eta;        // [B][n][n]
  long tau_bs, eta_bs;
  float alpha, beta;
  const int64_t *paths;          // [B][rows][A]
  const float *rowsum;           // [B][rows-1][A]
  const float *grad_logp;        // [B][rows-1][A]
  const int32_t *lens;           // [B][A] (CVRP) or null
  const float *demand;           // [B][n] (CVRP) or null
  float capacity;
  float *grad_eta;               // [B][n][n], accumulated into (caller zeroes)
};

template <bool CVRP>
__global__ void __launch_bounds__(256)
sample_backward_kernel(const BackwardParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bpi = (p.A + 3) >> 2;
  const int b = blockIdx.x / bpi;
  const int a = (blockIdx.x - b * bpi) * 4 + wave;
  if (a >= p.A) return;
  const int n = p.n, A = p.A;
  const float *tau = p.tau + (size_t)b * p.tau_bs, *eta = p.eta + (size_t)b * p.eta_bs;
  const int64_t *path = p.paths + (size_t)b * p.rows * A + a;
  const float *rs = p.rowsum + (size_t)b * (p.rows - 1) * A + a;
  const float *gl = p.grad_logp + (size_t)b * (p.rows - 1) * A + a;
  float *grad = p.grad_eta + (size_t)b * n * n;
  const float *demand = CVRP ? p.demand + (size_t)b * n : nullptr;
  const int len = CVRP ? p.lens[(size_t)b * A + a] : n;
  const int chunks = (n + 63) / 64;            // lane owns k = lane + 64*c (visited bit c)

  uint64_t vis = 0;                            // n <= 4096 -> <= 64 chunks
  int prev = (int)path[0];
  if (!CVRP && (prev & 63) == lane) vis |= 1ull << (prev >> 6);
  int remaining = n - 1;
  float used = CVRP ? demand[0] : 0.0f;
  for (int t = 1; t < len; ++t) {
    const int j = (int)path[(size_t)t * A];
    const float g = gl[(size_t)(t - 1) * A];
    const float S = rs[(size_t)(t - 1) * A];
    const float *trow = tau + (size_t)prev * n, *erow = eta + (size_t)prev * n;
    if (g != 0.0f) {
      const float pj = pw(trow[j], p.alpha) * pw(erow[j], p.beta);
      const float pr = pj / S;
      if (pr > DACO_EPS_F32 && pr < 1.0f - DACO_EPS_F32) {      // inside the clamp: gradient flows
        const float rem = CVRP ? p.capacity - used : 0.0f;
        const float c = g / S;
        float *grow = grad + (size_t)prev * n;
        for (int ch = 0; ch < chunks; ++ch) {
          const int k = lane + 64 * ch;
          if (k >= n) break;
          bool open = !((vis >> ch) & 1);
          if (CVRP) {
            if (k == 0) open = !(prev == 0 && remaining > 0);
            open = open && !(demand[k] > rem);
          }
          if (!open) continue;
          const float e = erow[k];
          const float pk = pw(trow[k], p.alpha) * pw(e, p.beta);
          float val = -c * p.beta * (pk / e);
          if (k == j) val += g * p.beta / e;
          unsafeAtomicAdd(grow + k, val);
        }
      }
    }
    if (CVRP) {
      if (j != 0) { if ((j & 63) == lane) vis |= 1ull << (j >> 6); --remaining; }
      else used = 0.0f;
      used = used + demand[j];
    } else {
      if ((j & 63) == lane) vis |= 1ull << (j >> 6);
    }
    prev = j;
  }
}

}  // namespace daco

using namespace daco;

extern "C" int daco_sample_backward(void *stream, int B, int n, int A, int rows, const float *tau,
                                    long tau_bstride, const float *eta, long eta_bstride, float alpha,
                                    float beta, const int64_t *paths, const float *rowsum,
                                    const float *grad_logp, const int32_t *lens, const float *demand,
                                    float capacity, float *grad_eta) {
  if (B <= 0 || n < 2 || A <= 0 || rows < 2 || !tau || !eta || !paths || !rowsum || !grad_logp || !grad_eta) {
    set_error("daco_sample_backward: bad argument (B=%d n=%d A=%d rows=%d)", B, n, A, rows);
    return DACO_E_BADARG;
  }
  if (n > DACO_MAX_NODES) { set_error("daco_sample_backward: n=%d exceeds DACO_MAX_NODES", n); return DACO_E_TOOLARGE; }
  const bool cvrp = demand != nullptr;
  if (cvrp && !lens) { set_error("daco_sample_backward: CVRP needs the per-ant lengths"); return DACO_E_BADARG; }
  if (!cvrp && rows != n) { set_error("daco_sample_backward: TSP needs rows == n"); return DACO_E_BADARG; }
  BackwardParams bp;
  bp.B = B; bp.n = n; bp.A = A; bp.rows = rows; bp.tau = tau; bp.eta = eta; bp.tau_bs = tau_bstride;
  bp.eta_bs = eta_bstride; bp.alpha = alpha; bp.beta = beta; bp.paths = paths; bp.rowsum = rowsum;
  bp.grad_logp = grad_logp; bp.lens = lens; bp.demand = demand; bp.capacity = capacity; bp.grad_eta = grad_eta;
  dim3 grid((unsigned)(B * ((A + 3) / 4))), block(256);
  if (cvrp) hipLaunchKernelGGL(sample_backward_kernel<true>, grid, block, 0, (hipStream_t)stream, bp);
  else hipLaunchKernelGGL(sample_backward_kernel<false>, grid, block, 0, (hipStream_t)stream, bp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("sample_backward_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}
