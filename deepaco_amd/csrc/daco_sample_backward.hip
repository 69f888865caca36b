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
  const double *demand64;        // [B][n] or null: load bookkeeping in double (cvrp_nls/, as daco_cvrp_sample's demand64)
  double capacity64;
  float *grad_eta;               // [B][n][n], accumulated into (caller zeroes)
  int segs;                      // waves per (instance, ant): each replays the whole route but differentiates 1/segs of its steps
};

template <bool CVRP>
__global__ void __launch_bounds__(256)
sample_backward_kernel(const BackwardParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bpi = (p.A + 3) >> 2;
  // Small batches (a training step has a few hundred ants) leave most of the device idle with one wave per ant, and a wave's
  // route is a chain of ~n dependent steps: `segs` waves share an ant, each replays the cheap bookkeeping of the whole
  // route (visited set, capacity) and differentiates only the steps of its own contiguous segment.
  const int seg = blockIdx.x % p.segs, blk = blockIdx.x / p.segs;
  const int b = blk / bpi;
  const int a = (blk - b * bpi) * 4 + wave;
  if (a >= p.A) return;
  const int n = p.n, A = p.A;
  const float *tau = p.tau + (size_t)b * p.tau_bs, *eta = p.eta + (size_t)b * p.eta_bs;
  const int64_t *path = p.paths + (size_t)b * p.rows * A + a;
  const float *rs = p.rowsum + (size_t)b * (p.rows - 1) * A + a;
  const float *gl = p.grad_logp + (size_t)b * (p.rows - 1) * A + a;
  float *grad = p.grad_eta + (size_t)b * n * n;
  const float *demand = CVRP ? p.demand + (size_t)b * n : nullptr;
  const double *demand64 = (CVRP && p.demand64) ? p.demand64 + (size_t)b * n : nullptr;   // (uniform) load bookkeeping in double
  const int len_all = CVRP ? p.lens[(size_t)b * A + a] : n;
  const int per = (len_all - 1 + p.segs - 1) / p.segs;
  const int t_lo = 1 + seg * per, len = min(len_all, t_lo + per);       // this wave differentiates steps [t_lo, len)
  const int chunks = (n + 63) / 64;            // lane owns k = lane + 64*c (visited bit c)

  uint64_t vis = 0;                            // n <= 4096 -> <= 64 chunks
  int prev = (int)path[0];
  if (!CVRP && (prev & 63) == lane) vis |= 1ull << (prev >> 6);
  int remaining = n - 1;
  float used = CVRP ? demand[0] : 0.0f;
  double used64 = demand64 ? 0.0 + demand64[0] : 0.0;
  // Round 6, last session: a step used to be three dependent memory round trips (its path entry, gradient and row sum; then the
  // chosen candidate's tau / eta; then the lanes' candidates, one chunk of 64 after the other).  Now the NEXT step's three scalars
  // are fetched while this one is worked on, and the chosen candidate's pair goes out together with the first four chunks of
  // candidates (and their demands): one round trip per step up to n = 256, one more per 256 candidates beyond.  Same arithmetic.
  int j = len > 1 ? (int)path[(size_t)A] : 0;
  float g = len > 1 ? gl[0] : 0.0f, S = len > 1 ? rs[0] : 1.0f;
  for (int t = 1; t < len; ++t) {
    const int tn = t + 1 < len ? t + 1 : t;                   // (the last step fetches itself again)
    const int jn = (int)path[(size_t)tn * A];
    const float gn = gl[(size_t)(tn - 1) * A], Sn = rs[(size_t)(tn - 1) * A];
    const float *trow = tau + (size_t)prev * n, *erow = eta + (size_t)prev * n;
    if (t >= t_lo && g != 0.0f) {
      float e4[4], t4[4], d4[4] = {0.f, 0.f, 0.f, 0.f};
      double dd4[4] = {0., 0., 0., 0.};
      const float tj = trow[j], ej = erow[j];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = lane + 64 * u, kc = k < n ? k : 0;
        e4[u] = erow[kc]; t4[u] = trow[kc];
        if (CVRP) d4[u] = demand[kc];
      }
      if (CVRP && demand64) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int k = lane + 64 * u; dd4[u] = demand64[k < n ? k : 0]; }
      }
      const float pj = pw(tj, p.alpha) * pw(ej, p.beta);
      const float pr = pj / S;
      if (pr > DACO_EPS_F32 && pr < 1.0f - DACO_EPS_F32) {      // inside the clamp: gradient flows
        const float rem = CVRP ? p.capacity - used : 0.0f;
        const double rem64 = p.capacity64 - used64;
        const float c = g / S;
        float *grow = grad + (size_t)prev * n;
        for (int c0 = 0; c0 < chunks; c0 += 4) {
          if (c0 > 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int k = lane + 64 * (c0 + u), kc = k < n ? k : 0;
              e4[u] = erow[kc]; t4[u] = trow[kc];
              if (CVRP) d4[u] = demand[kc];
            }
            if (CVRP && demand64) {
#pragma unroll
              for (int u = 0; u < 4; ++u) { const int k = lane + 64 * (c0 + u); dd4[u] = demand64[k < n ? k : 0]; }
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int ch = c0 + u, k = lane + 64 * ch;
            if (ch >= chunks || k >= n) continue;
            bool open = !((vis >> ch) & 1);
            if (CVRP) {
              if (k == 0) open = !(prev == 0 && remaining > 0);
              open = open && !(demand64 ? dd4[u] > rem64 : d4[u] > rem);
            }
            if (!open) continue;
            const float e = e4[u];
            const float tk = t4[u];
            const float pk = pw(tk, p.alpha) * pw(e, p.beta);
            float val = -c * dprob_deta(pk, tk, e, p.alpha, p.beta);
            if (k == j) val += g * p.beta / e;
            unsafeAtomicAdd(grow + k, val);
          }
        }
      }
    }
    if (CVRP) {
      if (j != 0) { if ((j & 63) == lane) vis |= 1ull << (j >> 6); --remaining; }
      else { used = 0.0f; used64 = 0.0; }
      used = used + demand[j];
      if (demand64) used64 = used64 + demand64[j];
    } else {
      if ((j & 63) == lane) vis |= 1ull << (j >> 6);
    }
    prev = j;
    j = jn; g = gn; S = Sn;
  }
}

// ------------------------------------------------------------------ sibling constructions
// The same gradient for the fused sop / pctsp / op / mkp constructions (daco_sibling_sample): the
// wave replays the route with the problem's own feasibility bookkeeping -- the rules of
// daco_sample_kernel.h, restated on the one-candidate-per-lane-and-chunk layout of this file
// (sop/aco.py:128-180, pctsp/aco.py:166-188, op/aco.py:195-224, mkp/aco.py:163-183).
struct SibBackwardParams {
  BackwardParams b;
  const float *aux_vec;          // [B][n]
  const float *aux_mat;          // [B][n][n] (stride aux_bs)
  long aux_bs;
  float scalar0;
  const float *wts;              // MKP [B][n][m]
  int m;
};

constexpr int SIBB_CHUNKS = 16;  // n <= 1024 (larger instances take the draw-by-draw gradient path)

template <int KIND>
__global__ void __launch_bounds__(256)
sibling_backward_kernel(const SibBackwardParams q) {
  constexpr bool SOP = KIND == DACO_SIB_SOP, PCTSP = KIND == DACO_SIB_PCTSP, OP = KIND == DACO_SIB_OP, MKP = KIND == DACO_SIB_MKP;
  constexpr bool DUMMY = OP || MKP;
  const BackwardParams &p = q.b;
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
  const float *avec = (SOP || PCTSP || OP) ? q.aux_vec + (size_t)b * n : nullptr;
  const float *amat = (SOP || OP) ? q.aux_mat + (size_t)b * q.aux_bs : nullptr;
  const float *wts = MKP ? q.wts + (size_t)b * n * q.m : nullptr;
  const int len = SOP ? n : p.lens[(size_t)b * A + a];
  const int chunks = (n + 63) / 64;

  uint32_t vis = 0, sticky = 0;                        // bit c: candidate lane + 64c
  int prev = (int)path[0];
  float cnt[SIBB_CHUNKS];                              // SOP: unvisited predecessors; OP: way home
  float used = 0.0f, knap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int remaining = n - 1;
  if (!PCTSP && (prev & 63) == lane) vis |= 1u << (prev >> 6);
#pragma unroll
  for (int c = 0; c < SIBB_CHUNKS; ++c) {
    const int k = lane + 64 * c;
    cnt[c] = 0.0f;
    if (c < chunks && k < n) {
      if (SOP) cnt[c] = avec[k] - amat[k];               // node 0 is visited first: row 0 of "who waits"
      if (OP) cnt[c] = avec[k];
    }
  }
  if (MKP) for (int dd = 0; dd < 8; ++dd) if (dd < q.m) knap[dd] = wts[(size_t)prev * q.m + dd];

  for (int t = 1; t < len; ++t) {
    const int j = (int)path[(size_t)t * A];
    const float g = gl[(size_t)(t - 1) * A];
    const float S = rs[(size_t)(t - 1) * A];
    const float *trow = tau + (size_t)prev * n, *erow = eta + (size_t)prev * n;
    // ---- candidates open at this step
    uint32_t open = 0;
    const bool depot_open = PCTSP ? (prev != 0 && (used > q.scalar0 || remaining == 0)) : true;
#pragma unroll
    for (int c = 0; c < SIBB_CHUNKS; ++c) {
      const int k = lane + 64 * c;
      if (c < chunks && k < n) {
        bool o = !((vis >> c) & 1u);
        if (SOP) o = o && cnt[c] == 0.0f;
        if (PCTSP && k == 0) o = depot_open;
        if (OP) {
          if (used + amat[(size_t)prev * n + k] + cnt[c] > q.scalar0) sticky |= 1u << c;
          o = o && !((sticky >> c) & 1u) && k < n - 1;
        }
        if (MKP) {
          bool over = false;
          if (k < n - 1)
            for (int dd = 0; dd < 8; ++dd)
              if (dd < q.m) over = over || (knap[dd] + wts[(size_t)k * q.m + dd] > q.scalar0);
          if (over) sticky |= 1u << c;
          o = o && !((sticky >> c) & 1u) && k < n - 1;
        }
        if (o) open |= 1u << c;
      }
    }
    if (g != 0.0f) {
      const float pj = pw(trow[j], p.alpha) * pw(erow[j], p.beta);
      const float pr = pj / S;
      if (pr > DACO_EPS_F32 && pr < 1.0f - DACO_EPS_F32) {      // inside the clamp: gradient flows
        const float cg = g / S;
        float *grow = grad + (size_t)prev * n;
#pragma unroll
        for (int c = 0; c < SIBB_CHUNKS; ++c) {
          const int k = lane + 64 * c;
          if (c < chunks && ((open >> c) & 1u)) {
            const float e = erow[k];
            const float tk = trow[k];
            const float pk = pw(tk, p.alpha) * pw(e, p.beta);
            float val = -cg * dprob_deta(pk, tk, e, p.alpha, p.beta);
            if (k == j) val += g * p.beta / e;
            unsafeAtomicAdd(grow + k, val);
          }
        }
      }
    }
    // ---- the move
    if (SOP) {
      if ((j & 63) == lane) vis |= 1u << (j >> 6);
#pragma unroll
      for (int c = 0; c < SIBB_CHUNKS; ++c) {
        const int k = lane + 64 * c;
        if (c < chunks && k < n) cnt[c] = cnt[c] - amat[(size_t)j * n + k];
      }
    } else if (PCTSP) {
      used = used + avec[j];
      if (j != 0) { if ((j & 63) == lane) vis |= 1u << (j >> 6); --remaining; }
    } else if (OP) {
      used = used + amat[(size_t)prev * n + j];
      if ((j & 63) == lane) vis |= 1u << (j >> 6);
    } else {
      if ((j & 63) == lane) vis |= 1u << (j >> 6);
      for (int dd = 0; dd < 8; ++dd) if (dd < q.m) knap[dd] = knap[dd] + wts[(size_t)j * q.m + dd];
    }
    prev = j;
  }
  (void)DUMMY;
}

}  // namespace daco

using namespace daco;

extern "C" int daco_sibling_backward(void *stream, int kind, int B, int n, int A, int rows, const float *tau,
                                     long tau_bstride, const float *eta, long eta_bstride, float alpha, float beta,
                                     const float *aux_vec, const float *aux_mat, long aux_mat_bstride, float scalar0,
                                     const float *item_weights, int m, const int64_t *paths, const float *rowsum,
                                     const float *grad_logp, const int32_t *lens, float *grad_eta) {
  if (B <= 0 || n < 2 || A <= 0 || rows < 2 || !tau || !eta || !paths || !rowsum || !grad_logp || !grad_eta) {
    set_error("daco_sibling_backward: bad argument (B=%d n=%d A=%d rows=%d)", B, n, A, rows);
    return DACO_E_BADARG;
  }
  if (n > 64 * SIBB_CHUNKS) { set_error("daco_sibling_backward: n=%d exceeds %d", n, 64 * SIBB_CHUNKS); return DACO_E_TOOLARGE; }
  if (kind != DACO_SIB_SOP && !lens) { set_error("daco_sibling_backward: variable-length kinds need lens"); return DACO_E_BADARG; }
  if (kind == DACO_SIB_MKP && (m < 1 || m > 8 || !item_weights)) { set_error("daco_sibling_backward: mkp needs 1..8 weight columns"); return DACO_E_BADARG; }
  if ((kind == DACO_SIB_SOP || kind == DACO_SIB_OP) && (!aux_vec || !aux_mat)) { set_error("daco_sibling_backward: aux_vec / aux_mat missing"); return DACO_E_BADARG; }
  if (kind == DACO_SIB_PCTSP && !aux_vec) { set_error("daco_sibling_backward: aux_vec missing"); return DACO_E_BADARG; }
  SibBackwardParams sp;
  BackwardParams &bp = sp.b;
  bp.B = B; bp.n = n; bp.A = A; bp.rows = rows; bp.tau = tau; bp.eta = eta; bp.tau_bs = tau_bstride;
  bp.eta_bs = eta_bstride; bp.alpha = alpha; bp.beta = beta; bp.paths = paths; bp.rowsum = rowsum;
  bp.grad_logp = grad_logp; bp.lens = lens; bp.demand = nullptr; bp.capacity = 0.0f; bp.demand64 = nullptr; bp.capacity64 = 0.0; bp.grad_eta = grad_eta; bp.segs = 1;
  sp.aux_vec = aux_vec; sp.aux_mat = aux_mat; sp.aux_bs = aux_mat_bstride; sp.scalar0 = scalar0; sp.wts = item_weights; sp.m = m;
  dim3 grid((unsigned)(B * ((A + 3) / 4))), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (kind) {
    case DACO_SIB_SOP: hipLaunchKernelGGL(sibling_backward_kernel<DACO_SIB_SOP>, grid, block, 0, s, sp); break;
    case DACO_SIB_PCTSP: hipLaunchKernelGGL(sibling_backward_kernel<DACO_SIB_PCTSP>, grid, block, 0, s, sp); break;
    case DACO_SIB_OP: hipLaunchKernelGGL(sibling_backward_kernel<DACO_SIB_OP>, grid, block, 0, s, sp); break;
    case DACO_SIB_MKP: hipLaunchKernelGGL(sibling_backward_kernel<DACO_SIB_MKP>, grid, block, 0, s, sp); break;
    default: set_error("daco_sibling_backward: unknown kind %d", kind); return DACO_E_BADARG;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("sibling_backward_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}


extern "C" int daco_sample_backward(void *stream, int B, int n, int A, int rows, const float *tau,
                                    long tau_bstride, const float *eta, long eta_bstride, float alpha,
                                    float beta, const int64_t *paths, const float *rowsum,
                                    const float *grad_logp, const int32_t *lens, const float *demand,
                                    float capacity, float *grad_eta, const double *demand64, double capacity64) {
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
  bp.grad_logp = grad_logp; bp.lens = lens; bp.demand = demand; bp.capacity = capacity; bp.demand64 = demand64; bp.capacity64 = capacity64; bp.grad_eta = grad_eta;
  // waves per ant so that a small batch still fills the device's ~2048 wave slots with independent chains
  int segs = 1;
  while (segs < 8 && (long)B * A * segs * 2 <= 2048) segs *= 2;
  bp.segs = segs;
  dim3 grid((unsigned)(B * ((A + 3) / 4) * segs)), block(256);
  if (cvrp) hipLaunchKernelGGL(sample_backward_kernel<true>, grid, block, 0, (hipStream_t)stream, bp);
  else hipLaunchKernelGGL(sample_backward_kernel<false>, grid, block, 0, (hipStream_t)stream, bp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("sample_backward_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}
