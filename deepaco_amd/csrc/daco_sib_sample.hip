// daco_sib_sample.hip -- fused solution construction for the sibling problems.
//
// Reference behaviour replaced (one launch instead of a Python loop of ~25 aten ops per step with a
// host sync each):  sop/aco.py:114-180 gen_path (precedence constraints), pctsp/aco.py:131-188 gen_sol
// (prize threshold opens the depot), op/aco.py:156-224 gen_sol (length budget incl. the way home;
// the reference checks feasibility in a per-ant Python loop), mkp/aco.py:113-183 gen_sol
// (m-dimensional knapsack; per-ant Python loop in the reference).  SMTWTP needs no kernel of its
// own (TSP kernel with the dummy job as fixed start) and BPP uses the CVRP kernel.
//
// All four are policies of the tour-construction kernel template (daco_sample_kernel.h): the same
// wave-per-ant mapping, lane layout, draws and Philox counters; only the per-step closure rules and
// the bookkeeping after a choice differ.  Per-candidate state (SOP predecessor counters, OP distance
// home) lives in registers next to the visited bitset; OP streams one extra padded distance row
// per step.  For OP and MKP the absorbing dummy node is never drawn: an ant whose candidates are all
// closed stops and its column is padded with the dummy, which is what the reference's remaining
// steps do (the dummy is then the only open node, probability 1).
#include "daco_sample_kernel.h"

namespace daco {

// dst[b][r][0..ld) = src[b][r][0..n) padded with `fill`
__global__ void __launch_bounds__(256)
pad_matrix_kernel(int B, int n, int ld, const float *src, long src_bs, float *dst, float fill) {
  const long total = (long)B * n * ld;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % ld);
    const long row = i / ld;
    const int b = (int)(row / n), r = (int)(row % n);
    dst[i] = k < n ? src[b * src_bs + (long)r * n + k] : fill;
  }
}

}  // namespace daco

using namespace daco;

extern "C" size_t daco_sibling_workspace_bytes(int B, int n, int mode) {
  if (B <= 0 || n <= 0 || n > DACO_MAX_NODES) return 0;
  return daco_tsp_sample_workspace_bytes(B, n, mode) + align256((size_t)B * n * ld_alloc(n) * sizeof(float));
}

extern "C" int daco_sibling_sample(void *stream, int kind, int B, int n, int A, const float *tau, long tau_bstride,
                                   const float *eta, long eta_bstride, float alpha, float beta,
                                   const float *aux_vec, const float *aux_mat, long aux_mat_bstride, float scalar0,
                                   const float *item_weights, int m, int mode, const int64_t *start,
                                   const float *noise, int noise_steps, uint64_t seed, uint64_t iter,
                                   uint32_t ant_gid0, int Lmax, int64_t *paths, float *logp, float *rowsum,
                                   int32_t *lens, int32_t *flags, void *workspace, size_t workspace_bytes) {
  if (B <= 0 || n < 2 || A <= 0 || !tau || !eta || !paths || !workspace) {
    set_error("daco_sibling_sample: bad argument (B=%d n=%d A=%d)", B, n, A);
    return DACO_E_BADARG;
  }
  if (n > DACO_MAX_NODES) { set_error("daco_sibling_sample: n=%d exceeds DACO_MAX_NODES=%d", n, DACO_MAX_NODES); return DACO_E_TOOLARGE; }
  if (mode == DACO_SCAN_WAVE) mode = DACO_SCAN;
  if (mode < 0 || mode > 2) { set_error("daco_sibling_sample: bad mode %d", mode); return DACO_E_BADARG; }
  if (mode == DACO_RACE_NOISE && (!noise || noise_steps <= 0)) { set_error("daco_sibling_sample: DACO_RACE_NOISE needs a noise tensor"); return DACO_E_BADARG; }
  const bool varlen = kind != DACO_SIB_SOP;
  if (varlen && (Lmax < 2 || !lens)) { set_error("daco_sibling_sample: variable-length kinds need Lmax >= 2 and lens"); return DACO_E_BADARG; }
  if ((kind == DACO_SIB_SOP || kind == DACO_SIB_OP) && (!aux_vec || !aux_mat)) { set_error("daco_sibling_sample: kind %d needs aux_vec and aux_mat", kind); return DACO_E_BADARG; }
  if (kind == DACO_SIB_PCTSP && !aux_vec) { set_error("daco_sibling_sample: PCTSP needs the prizes in aux_vec"); return DACO_E_BADARG; }
  if (kind == DACO_SIB_MKP && (!item_weights || m < 1 || m > 8)) { set_error("daco_sibling_sample: MKP needs item_weights and 1 <= m <= 8"); return DACO_E_BADARG; }
  const size_t need = daco_sibling_workspace_bytes(B, n, mode);
  if (workspace_bytes < need) { set_error("daco_sibling_sample: workspace %zu < %zu bytes", workspace_bytes, need); return DACO_E_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  const int vec = vec_for_n(n), CH = inst_chunks(n), ld = ld_alloc(n);
  const size_t pbytes = daco_tsp_sample_workspace_bytes(B, n, mode);
  float *P = (float *)workspace;
  float *R = mode == DACO_RACE_PHILOX ? (float *)((char *)workspace + pbytes / 2) : nullptr;
  float *auxp = (float *)((char *)workspace + pbytes);
  const long total = (long)B * n * ld;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  launch_prob_matrix(B, n, ld, tau, tau_bstride, eta, eta_bstride, alpha, beta, P, R, s);
  if (aux_mat)
    hipLaunchKernelGGL(pad_matrix_kernel, dim3(blocks), dim3(256), 0, s, B, n, ld, aux_mat, aux_mat_bstride, auxp, 0.0f);
  SampleParams sp;
  sp.B = B; sp.n = n; sp.A = A; sp.ld = ld; sp.CH = CH;
  sp.P = P; sp.R = R; sp.norm_passes = 1; sp.start = start; sp.fixed_start = 0;
  sp.noise = noise; sp.seed = seed; sp.iter = iter; sp.iter_dev = nullptr; sp.ant_gid0 = ant_gid0; sp.gid_bstride = 0;
  sp.paths = paths; sp.logp = logp; sp.rowsum = rowsum; sp.flags = flags;
  sp.dist = nullptr; sp.dist_bs = 0; sp.costs = nullptr; sp.nbr = nullptr; sp.hubmask = nullptr; sp.tab_lens = nullptr;
  sp.demand = nullptr; sp.capacity = 0.0f; sp.Lmax = Lmax; sp.noise_steps = noise_steps; sp.lens = lens;
  sp.mask = nullptr; sp.step = 0;
  sp.aux_vec = aux_vec; sp.aux_mat = auxp; sp.scalar0 = scalar0; sp.wts = item_weights; sp.m = m;
  const bool lp = logp != nullptr;
  hipError_t e;
  switch (kind) {
    case DACO_SIB_SOP: e = dispatch_sample<PROB_SOP>(sp, vec, CH, mode, lp, s); break;
    case DACO_SIB_PCTSP: e = dispatch_sample<PROB_PCTSP>(sp, vec, CH, mode, lp, s); break;
    case DACO_SIB_OP: e = dispatch_sample<PROB_OP>(sp, vec, CH, mode, lp, s); break;
    case DACO_SIB_MKP: e = dispatch_sample<PROB_MKP>(sp, vec, CH, mode, lp, s); break;
    default: set_error("daco_sibling_sample: unknown kind %d", kind); return DACO_E_BADARG;
  }
  if (e != hipSuccess) { set_error("sibling sample kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}
