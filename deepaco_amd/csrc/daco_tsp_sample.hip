// daco_tsp_sample.hip -- tour construction (ACO.gen_path / pick_move) for gfx950.
//
// Reference behaviour replaced: tsp/aco.py:134-177, tsp_nls/aco.py:184-220 (torch path) and,
// for the prefix-scan mode, the roulette sampler tsp_nls/aco.py:260-275.
//
// Design: one wavefront per (instance, ant) for the whole tour -- the
// n-1 dependent draws never leave the wave.  Each step streams ONE padded row of the fused
// transition matrix P = tau^alpha * eta^beta (built once per call by prob_matrix_kernel, so
// the per-step traffic is 4*ld bytes instead of the reference's 8n) with 16-byte loads, lane
// l owning candidates (c*64+l)*VEC+v.  The visited set is a per-lane 64-bit register bitset;
// the draw is a DPP reduction / prefix scan; nothing is staged through LDS because no lane
// consumes another lane's candidates.  Workgroups are remapped so each XCD walks whole
// instances (rows stay in its private L2).
#include "daco_sample_kernel.h"

namespace daco {

// P = tau^alpha * eta^beta with zero padding; R = 1/P with +inf padding (optional).  A pure stream (two matrices in, one or
// two out): one workgroup per row and grid-stride over the rows, 16-byte accesses when the rows allow it (n % 4 == 0 and
// aligned bases; the padded output rows always do), no integer division per element.
template <bool VEC4>
__global__ void __launch_bounds__(256)
prob_matrix_kernel(int B, int n, int ld, const float *tau, long tau_bs, const float *eta, long eta_bs,
                   float alpha, float beta, float *P, float *R) {
  const int rows = B * n;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int b = row / n, r = row - b * n;
    const float *tr = tau + b * tau_bs + (long)r * n, *er = eta + b * eta_bs + (long)r * n;
    float *pr = P + (size_t)row * ld, *rr = R ? R + (size_t)row * ld : nullptr;
    if constexpr (VEC4) {
      for (int k = threadIdx.x * 4; k < ld; k += 1024) {
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 q = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff());
        if (k < n) {                                          // (n % 4 == 0: a vector is inside the row or in the padding)
          const float4 t = *reinterpret_cast<const float4 *>(tr + k), e = *reinterpret_cast<const float4 *>(er + k);
          p.x = pw(t.x, alpha) * pw(e.x, beta); p.y = pw(t.y, alpha) * pw(e.y, beta);
          p.z = pw(t.z, alpha) * pw(e.z, beta); p.w = pw(t.w, alpha) * pw(e.w, beta);
          if (rr) { q.x = 1.0f / p.x; q.y = 1.0f / p.y; q.z = 1.0f / p.z; q.w = 1.0f / p.w; }
        }
        *reinterpret_cast<float4 *>(pr + k) = p;
        if (rr) *reinterpret_cast<float4 *>(rr + k) = q;
      }
    } else {
      for (int k = threadIdx.x; k < ld; k += 256) {
        float p = 0.0f;
        if (k < n) p = pw(tr[k], alpha) * pw(er[k], beta);
        pr[k] = p;
        if (rr) rr[k] = k < n ? 1.0f / p : __builtin_inff();
      }
    }
  }
}

void launch_prob_matrix(int B, int n, int ld, const float *tau, long tau_bs, const float *eta, long eta_bs, float alpha,
                               float beta, float *P, float *R, hipStream_t s) {
  const long rows = (long)B * n;
  const int blocks = (int)(rows < 16384 ? rows : 16384);
  const bool vec4 = (n & 3) == 0 && (tau_bs & 3) == 0 && (eta_bs & 3) == 0 && (((uintptr_t)tau | (uintptr_t)eta) & 15) == 0;
  if (vec4) hipLaunchKernelGGL(prob_matrix_kernel<true>, dim3(blocks), dim3(256), 0, s, B, n, ld, tau, tau_bs, eta, eta_bs, alpha, beta, P, R);
  else hipLaunchKernelGGL(prob_matrix_kernel<false>, dim3(blocks), dim3(256), 0, s, B, n, ld, tau, tau_bs, eta, eta_bs, alpha, beta, P, R);
}

}  // namespace daco

using namespace daco;

extern "C" int daco_vec_for_n(int n) { return vec_for_n(n); }
extern "C" int daco_ld_for_n(int n) { return ld_for_n(n); }


extern "C" size_t daco_tsp_sample_workspace_bytes(int B, int n, int mode) {
  if (B <= 0 || n <= 0 || n > DACO_MAX_NODES) return 0;
  const size_t mat = align256((size_t)B * n * ld_alloc(n) * sizeof(float));
  return mode == DACO_RACE_PHILOX ? 2 * mat : mat;
}

extern "C" int daco_tsp_sample(void *stream, int B, int n, int A, const float *tau, long tau_bstride,
                               const float *eta, long eta_bstride, float alpha, float beta, int mode,
                               int norm_passes, const int64_t *start, int fixed_start,
                               const float *noise, uint64_t seed, uint64_t iter, const uint64_t *iter_offset, uint32_t ant_gid0,
                               int ant_gid_bstride,
                               int64_t *paths, float *logp, float *rowsum, int32_t *flags,
                               const float *dist, long dist_bstride, float *costs, uint32_t *nbr,
                               void *workspace, size_t workspace_bytes, void *ev_begin, void *ev_end) {
  if (B <= 0 || n < 2 || A <= 0 || !tau || !eta || !paths || !workspace) {
    set_error("daco_tsp_sample: bad argument (B=%d n=%d A=%d tau=%p eta=%p paths=%p ws=%p)", B, n, A,
              (const void *)tau, (const void *)eta, (void *)paths, workspace);
    return DACO_E_BADARG;
  }
  if (n > DACO_MAX_NODES) { set_error("daco_tsp_sample: n=%d exceeds DACO_MAX_NODES=%d", n, DACO_MAX_NODES); return DACO_E_TOOLARGE; }
  if (mode < 0 || mode > 3 || norm_passes < 0 || norm_passes > 2) { set_error("daco_tsp_sample: bad mode %d / norm_passes %d", mode, norm_passes); return DACO_E_BADARG; }
  const bool packed = mode == DACO_SCAN && (size_t)n * A * 8 < ((size_t)1 << 32);   // several ants per wave, 32-bit offsets
  const bool four_per_wave = packed && n <= scan16_max_n(), two_per_wave = packed && !four_per_wave && n <= tsp_scan32_max_n();
  if (mode == DACO_SCAN_WAVE) mode = DACO_SCAN;
  if (mode == DACO_RACE_NOISE && !noise) { set_error("daco_tsp_sample: DACO_RACE_NOISE needs a noise tensor"); return DACO_E_BADARG; }
  if (fixed_start >= n) { set_error("daco_tsp_sample: fixed_start %d >= n %d", fixed_start, n); return DACO_E_BADARG; }
  if (ant_gid_bstride < 0 || (ant_gid_bstride > 0 && ant_gid_bstride < A)) { set_error("daco_tsp_sample: ant_gid_bstride %d < A %d", ant_gid_bstride, A); return DACO_E_BADARG; }
  if (costs && !dist) { set_error("daco_tsp_sample: fused costs need the distance matrix"); return DACO_E_BADARG; }
  const size_t need = daco_tsp_sample_workspace_bytes(B, n, mode);
  if (workspace_bytes < need) { set_error("daco_tsp_sample: workspace %zu < %zu bytes", workspace_bytes, need); return DACO_E_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  const int vec = vec_for_n(n), CH = inst_chunks(n), ld = ld_alloc(n);
  float *P = (float *)workspace;
  float *R = mode == DACO_RACE_PHILOX ? (float *)((char *)workspace + need / 2) : nullptr;
  {
    launch_prob_matrix(B, n, ld, tau, tau_bstride, eta, eta_bstride, alpha, beta, P, R, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("prob_matrix_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  }
  SampleParams sp;
  sp.B = B; sp.n = n; sp.A = A; sp.ld = ld; sp.CH = CH;
  sp.P = P; sp.R = R; sp.norm_passes = norm_passes; sp.start = start; sp.fixed_start = fixed_start;
  sp.noise = noise; sp.seed = seed; sp.iter = iter; sp.iter_dev = iter_offset; sp.ant_gid0 = ant_gid0; sp.gid_bstride = ant_gid_bstride;
  sp.paths = paths; sp.logp = logp; sp.rowsum = rowsum; sp.flags = flags;
  sp.dist = dist; sp.dist_bs = dist_bstride; sp.costs = costs; sp.nbr = nbr; sp.hubmask = nullptr; sp.tab_lens = nullptr;
  sp.demand = nullptr; sp.capacity = 0.0f; sp.Lmax = 0; sp.noise_steps = 0; sp.lens = nullptr; sp.demand64 = nullptr; sp.capacity64 = 0.0;
  sp.mask = nullptr; sp.step = 0;
  sp.aux_vec = nullptr; sp.aux_mat = nullptr; sp.scalar0 = 0.0f; sp.wts = nullptr; sp.m = 0;
  const bool lp = logp != nullptr;
  hipError_t e;
  if (ev_begin && hipEventRecord((hipEvent_t)ev_begin, s) != hipSuccess) { set_error("hipEventRecord(ev_begin) failed"); return DACO_E_HIP; }
  e = four_per_wave ? launch_tsp_scan16(sp, lp, s) : two_per_wave ? launch_tsp_scan32(sp, lp, s) : dispatch_sample<PROB_TSP>(sp, vec, CH, mode, lp, s);
  if (e != hipSuccess) { set_error("tsp_sample_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  if (ev_end && hipEventRecord((hipEvent_t)ev_end, s) != hipSuccess) { set_error("hipEventRecord(ev_end) failed"); return DACO_E_HIP; }
  return DACO_OK;
}

extern "C" int daco_cvrp_sample(void *stream, int B, int n, int A, const float *tau, long tau_bstride,
                                const float *eta, long eta_bstride, float alpha, float beta,
                                const float *demand, float capacity, int mode, const float *noise,
                                int noise_steps, uint64_t seed, uint64_t iter, const uint64_t *iter_offset, uint32_t ant_gid0, int ant_gid_bstride,
                                int Lmax, int64_t *paths, float *logp, float *rowsum, int32_t *lens, int32_t *flags,
                                const float *dist, long dist_bstride, float *costs, void *next_table,
                                void *workspace, size_t workspace_bytes, const double *demand64, double capacity64,
                                void *ev_begin, void *ev_end) {
  if (B <= 0 || n < 2 || A <= 0 || !tau || !eta || !demand || !paths || !workspace || Lmax < 2) {
    set_error("daco_cvrp_sample: bad argument (B=%d n=%d A=%d Lmax=%d)", B, n, A, Lmax);
    return DACO_E_BADARG;
  }
  if (n > DACO_MAX_NODES) { set_error("daco_cvrp_sample: n=%d exceeds DACO_MAX_NODES=%d", n, DACO_MAX_NODES); return DACO_E_TOOLARGE; }
  if (ant_gid_bstride < 0 || (ant_gid_bstride > 0 && ant_gid_bstride < A)) { set_error("daco_cvrp_sample: ant_gid_bstride %d < A %d", ant_gid_bstride, A); return DACO_E_BADARG; }
  const bool packed = mode == DACO_SCAN && (size_t)n * A * 8 < ((size_t)1 << 32);
  // (float64 load bookkeeping: the one-ant-per-wavefront kernel's PROB_CVRP64 policy)
  // scan16_kernel family (4 / 8 / 16 lanes per ant), float32 or float64 load bookkeeping (above 512 nodes and in the race modes: the
  // one-ant-per-wavefront kernel's PROB_CVRP / PROB_CVRP64 policies)
  const bool four_per_wave = packed && n <= DACO_SCAN32_MAX_N;
  if (mode == DACO_SCAN_WAVE) mode = DACO_SCAN;
  if (mode < 0 || mode > 2) { set_error("daco_cvrp_sample: bad mode %d", mode); return DACO_E_BADARG; }
  if (mode == DACO_RACE_NOISE && (!noise || noise_steps <= 0)) { set_error("daco_cvrp_sample: DACO_RACE_NOISE needs a noise tensor"); return DACO_E_BADARG; }
  if (costs && !dist) { set_error("daco_cvrp_sample: fused costs need the distance matrix"); return DACO_E_BADARG; }
  const size_t need = daco_tsp_sample_workspace_bytes(B, n, mode);
  if (workspace_bytes < need) { set_error("daco_cvrp_sample: workspace %zu < %zu bytes", workspace_bytes, need); return DACO_E_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  uint32_t *hubmask = nullptr;
  int32_t *tab_lens = nullptr;
  if (next_table) {
    // "no successor" everywhere, empty depot sets; the kernel fills in what each ant really does
    const size_t tab = ((size_t)B * n * A * sizeof(uint32_t) + 255) & ~(size_t)255;
    hubmask = (uint32_t *)((char *)next_table + tab);
    tab_lens = (int32_t *)((char *)hubmask + (((size_t)B * A * ((n + 31) / 32) * sizeof(uint32_t) + 255) & ~(size_t)255));
    if (hipMemsetAsync(next_table, 0xFF, (size_t)B * n * A * sizeof(uint32_t), s) != hipSuccess ||
        hipMemsetAsync(hubmask, 0, (size_t)B * A * ((n + 31) / 32) * sizeof(uint32_t), s) != hipSuccess) {
      set_error("daco_cvrp_sample: hipMemsetAsync failed");
      return DACO_E_HIP;
    }
  }
  const int vec = vec_for_n(n), CH = inst_chunks(n), ld = ld_alloc(n);
  float *P = (float *)workspace;
  float *R = mode == DACO_RACE_PHILOX ? (float *)((char *)workspace + need / 2) : nullptr;
  {
    launch_prob_matrix(B, n, ld, tau, tau_bstride, eta, eta_bstride, alpha, beta, P, R, s);
  }
  SampleParams sp;
  sp.B = B; sp.n = n; sp.A = A; sp.ld = ld; sp.CH = CH;
  sp.P = P; sp.R = R; sp.norm_passes = 1; sp.start = nullptr; sp.fixed_start = 0;
  sp.noise = noise; sp.seed = seed; sp.iter = iter; sp.iter_dev = iter_offset; sp.ant_gid0 = ant_gid0; sp.gid_bstride = ant_gid_bstride;
  sp.paths = paths; sp.logp = logp; sp.rowsum = rowsum; sp.flags = flags;
  sp.dist = dist; sp.dist_bs = dist_bstride; sp.costs = costs; sp.nbr = (uint32_t *)next_table; sp.hubmask = hubmask; sp.tab_lens = tab_lens;
  sp.demand = demand; sp.capacity = capacity; sp.Lmax = Lmax; sp.noise_steps = noise_steps; sp.lens = lens;
  sp.demand64 = demand64; sp.capacity64 = capacity64;
  sp.mask = nullptr; sp.step = 0;
  sp.aux_vec = nullptr; sp.aux_mat = nullptr; sp.scalar0 = 0.0f; sp.wts = nullptr; sp.m = 0;
  if (ev_begin && hipEventRecord((hipEvent_t)ev_begin, s) != hipSuccess) { set_error("hipEventRecord(ev_begin) failed"); return DACO_E_HIP; }
  hipError_t e = four_per_wave ? launch_cvrp_scan16(sp, logp != nullptr, s)
               : demand64     ? dispatch_sample<PROB_CVRP64>(sp, vec, CH, mode, logp != nullptr, s)
                              : dispatch_sample<PROB_CVRP>(sp, vec, CH, mode, logp != nullptr, s);
  if (e != hipSuccess) { set_error("cvrp sample kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  if (ev_end && hipEventRecord((hipEvent_t)ev_end, s) != hipSuccess) { set_error("hipEventRecord(ev_end) failed"); return DACO_E_HIP; }
  return DACO_OK;
}

// ------------------------------------------------------------------ step-wise API (sibling problems)
extern "C" int daco_prob_matrix(void *stream, int B, int n, const float *tau, long tau_bstride, const float *eta,
                                long eta_bstride, float alpha, float beta, int mode, void *workspace,
                                size_t workspace_bytes) {
  if (B <= 0 || n < 2 || !tau || !eta || !workspace) { set_error("daco_prob_matrix: bad argument"); return DACO_E_BADARG; }
  if (n > DACO_MAX_NODES) { set_error("daco_prob_matrix: n=%d exceeds DACO_MAX_NODES", n); return DACO_E_TOOLARGE; }
  const size_t need = daco_tsp_sample_workspace_bytes(B, n, mode);
  if (workspace_bytes < need) { set_error("daco_prob_matrix: workspace %zu < %zu bytes", workspace_bytes, need); return DACO_E_WORKSPACE; }
  const int ld = ld_alloc(n);
  float *P = (float *)workspace;
  float *R = mode == DACO_RACE_PHILOX ? (float *)((char *)workspace + need / 2) : nullptr;
  launch_prob_matrix(B, n, ld, tau, tau_bstride, eta, eta_bstride, alpha, beta, P, R, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("prob_matrix_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}

extern "C" int daco_pick_move(void *stream, int B, int n, int A, const void *prob_workspace, size_t workspace_bytes,
                              int mode, const int64_t *prev, const float *mask, const float *noise, uint64_t seed,
                              uint64_t iter, uint32_t ant_gid0, int step, int64_t *actions, float *logp,
                              float *rowsum, int32_t *flags) {
  if (B <= 0 || n < 2 || A <= 0 || !prob_workspace || !prev || !mask || !actions || step < 0) {
    set_error("daco_pick_move: bad argument (B=%d n=%d A=%d step=%d)", B, n, A, step);
    return DACO_E_BADARG;
  }
  if (n > DACO_MAX_NODES) { set_error("daco_pick_move: n=%d exceeds DACO_MAX_NODES", n); return DACO_E_TOOLARGE; }
  if (mode == DACO_SCAN_WAVE) mode = DACO_SCAN;
  if (mode < 0 || mode > 2) { set_error("daco_pick_move: bad mode %d", mode); return DACO_E_BADARG; }
  if (mode == DACO_RACE_NOISE && !noise) { set_error("daco_pick_move: DACO_RACE_NOISE needs a noise tensor"); return DACO_E_BADARG; }
  const size_t need = daco_tsp_sample_workspace_bytes(B, n, mode);
  if (workspace_bytes < need) { set_error("daco_pick_move: workspace %zu < %zu bytes", workspace_bytes, need); return DACO_E_WORKSPACE; }
  const int vec = vec_for_n(n), CH = inst_chunks(n), ld = ld_alloc(n);
  SampleParams sp;
  sp.B = B; sp.n = n; sp.A = A; sp.ld = ld; sp.CH = CH;
  sp.P = (const float *)prob_workspace;
  sp.R = mode == DACO_RACE_PHILOX ? (const float *)((const char *)prob_workspace + need / 2) : nullptr;
  sp.norm_passes = 1; sp.start = prev; sp.fixed_start = -1; sp.noise = noise; sp.seed = seed; sp.iter = iter; sp.iter_dev = nullptr; sp.gid_bstride = 0;
  sp.ant_gid0 = ant_gid0; sp.paths = actions; sp.logp = logp; sp.rowsum = rowsum; sp.flags = flags;
  sp.dist = nullptr; sp.dist_bs = 0; sp.costs = nullptr; sp.nbr = nullptr;
  sp.demand = nullptr; sp.capacity = 0.0f; sp.Lmax = 0; sp.noise_steps = 1; sp.lens = nullptr; sp.demand64 = nullptr; sp.capacity64 = 0.0;
  sp.mask = mask; sp.step = step;
  sp.aux_vec = nullptr; sp.aux_mat = nullptr; sp.scalar0 = 0.0f; sp.wts = nullptr; sp.m = 0;
  hipError_t e = dispatch_sample<PROB_STEP>(sp, vec, CH, mode, logp != nullptr, (hipStream_t)stream);
  if (e != hipSuccess) { set_error("pick_move kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}
