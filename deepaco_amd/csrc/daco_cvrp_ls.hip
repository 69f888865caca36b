// daco_cvrp_ls.hip -- local search for CVRP solutions on the device, one workgroup per (instance, ant).
//
// Reference behaviour replaced: cvrp_nls/aco.py:114-126 (multiple_swap_star: one thread-pool task per ant), :443-448
// (neural_swapstar: search on the distances, a short search on the heuristic-derived matrix as perturbation, search
// on the distances again) and the path it takes into the vendored HGS-CVRP C++ through /tmp files
// (cvrp_nls/swapstar.py:240-271, HGS-CVRP-main/Program/C_Interface.cpp:128-172).  HGS's LocalSearch is third-party
// code with a randomised neighbourhood order; it is not restated here.  This kernel is a deterministic
// best-improvement search over the classical neighbourhoods HGS also uses, specified in this file and restated on
// the CPU in oracle/cvrp_ls.py (PARITY UNPINNED against the reference: the tests pin it against that restatement,
// and against feasibility / monotonicity / local-optimality properties).
//
// A solution is the reference's route sequence: 0 a b c 0 d e 0 ... 0 (cvrp/aco.py:138-165), zero-padded.  After
// removing empty routes it has L entries, s[0] = s[L-1] = 0.  Per iteration every move of three kinds is evaluated
// (f32, the expression order written below) and the one with the smallest change is applied if it is below -1e-6;
// ties go to the smallest (kind, i, j):
//   RELOCATE i -> j   customer u = s[i] moves between s[j] and s[j+1] (j != i-1, i); another route must have room
//                     change = ((d[a][c] - d[a][u]) - d[u][c]) + ((d[v][u] + d[u][w]) - d[v][w])
//   SWAP i <-> j      customers u = s[i], v = s[j], i < j; both routes must keep within capacity
//                     non-adjacent: ((d[a][v] + d[v][c]) - (d[a][u] + d[u][c])) + ((d[e][u] + d[u][g]) - (d[e][v] + d[v][g]))
//                     adjacent    : ((d[a][v] + d[v][u]) + d[u][g]) - ((d[a][u] + d[u][v]) + d[v][g])
//   2-OPT i..j        reverse s[i..j] inside one route, i < j
//                     change = ((d[a][v] + d[u][g]) - (d[a][u] + d[v][g])) + (float)(asym[j] - asym[i])
//                     (asym[k] = sum_{t<k} (d[s[t+1]][s[t]] - d[s[t]][s[t+1]]) in f64: what the inner edges cost more
//                      when walked backwards.  Exactly 0 for a symmetric matrix; the perturbation matrix
//                      1/(eta/rowmax + 1e-5) is not symmetric.  f64 because a difference of f32 running sums carries
//                      ~1e-5 of rounding noise at n = 200 -- above the 1e-6 acceptance threshold, and the search cycled)
// with a = s[i-1], c = s[i+1], e = s[j-1], g = s[j+1], v/w = s[j], s[j+1].  The distance matrix is staged in LDS when
// it fits (n <= 160: every gather of the search is then an LDS read), demands and route loads always are.
#include "daco_device.h"
#include "../../include/deepaco_hip.h"

namespace daco {

constexpr int LS_MAXL = 1024;      // longest sequence (2n+1 entries at most)
constexpr int LS_STAGE_MAX_N = 160;

struct LsBest { float delta; uint32_t code; };
__device__ inline bool ls_better(float d1, uint32_t c1, float d2, uint32_t c2) { return d1 < d2 || (d1 == d2 && c1 < c2); }

template <bool STAGE>
__global__ void __launch_bounds__(256)
cvrp_ls_kernel(int n, int A, int Lmax, const float *dist, long dist_bs, const float *demand, float capacity, int64_t *paths,
               int count, int32_t *lens_out, int32_t *moves_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t *s = reinterpret_cast<uint16_t *>(smem);                 // [LS_MAXL] sequence
  uint16_t *rid = s + LS_MAXL;                                      // [LS_MAXL] route of position k (of the gap after a depot)
  double *asym = reinterpret_cast<double *>(rid + LS_MAXL);         // [LS_MAXL] sum of d[s[t+1]][s[t]] - d[s[t]][s[t+1]], t < k
  float *load = reinterpret_cast<float *>(asym + LS_MAXL);          // [LS_MAXL] route loads
  float *dem = load + LS_MAXL;                                      // [n] demands (padded to n4)
  const int n4 = (n + 3) & ~3;
  float *redd = dem + n4;                                           // [4] reduction
  uint32_t *redc = reinterpret_cast<uint32_t *>(redd + 4);          // [4]
  int *shared_i = reinterpret_cast<int *>(redc + 4);                // [4]: L, applied flag
  float *dl = reinterpret_cast<float *>(shared_i + 4);              // [n*n] staged distances (STAGE)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / A, a = blockIdx.x - b * A;
  const float *dg = dist + (size_t)b * dist_bs;
  int64_t *col = paths + (size_t)b * Lmax * A + a;

  for (int k = tid; k < n; k += 256) dem[k] = demand[(size_t)b * n + k];
  if constexpr (STAGE) for (int k = tid; k < n * n; k += 256) dl[k] = dg[k];
  auto D = [&](int u, int v) -> float { return STAGE ? dl[u * n + v] : dg[(size_t)u * n + v]; };
  // ---- read the column, drop empty routes (serial: the sequence is a few hundred entries)
  if (tid == 0) {
    int L = 0;
    int prev = -1;
    for (int t = 0; t < Lmax; ++t) {
      const int v = (int)col[(size_t)t * A];
      if (v == 0 && prev == 0) continue;
      s[L++] = (uint16_t)v;
      prev = v;
    }
    if (L == 0 || s[L - 1] != 0) s[L++] = 0;
    shared_i[0] = L;
  }
  __syncthreads();
  int L = shared_i[0];
  int moves = 0;
  for (; moves < count; ++moves) {
    // ---- bookkeeping for this sequence: route ids, loads, running edge sums (one thread; L is small)
    if (tid == 0) {
      int r = -1;
      double w = 0.0;
      for (int k = 0; k < L; ++k) {
        if (s[k] == 0) { ++r; load[r] = 0.0f; }
        else load[r] = load[r] + dem[s[k]];
        rid[k] = (uint16_t)r;
        asym[k] = w;
        if (k + 1 < L) w = w + ((double)D(s[k + 1], s[k]) - (double)D(s[k], s[k + 1]));
      }
    }
    __syncthreads();
    float bd = 0.0f;                    // only improving moves qualify
    uint32_t bc = 0xFFFFFFFFu;
    // ---- RELOCATE: i in [1, L-2] customer, j in [0, L-2]
    const int nrel = (L - 2) * (L - 1);
    for (int x = tid; x < nrel; x += 256) {
      const int i = 1 + x / (L - 1), j = x - (i - 1) * (L - 1);
      const int u = s[i];
      if (u == 0 || j == i || j == i - 1) continue;
      if (rid[j] != rid[i] && load[rid[j]] + dem[u] > capacity) continue;
      const int pa = s[i - 1], pc = s[i + 1], v = s[j], w = s[j + 1];
      float rem = D(pa, pc) - D(pa, u);
      rem = rem - D(u, pc);
      float add = D(v, u) + D(u, w);
      add = add - D(v, w);
      const float delta = rem + add;
      const uint32_t code = (0u << 28) | ((uint32_t)i << 14) | (uint32_t)j;
      if (ls_better(delta, code, bd, bc)) { bd = delta; bc = code; }
    }
    // ---- SWAP: i < j customers
    const int nsw = (L - 2) * (L - 2);
    for (int x = tid; x < nsw; x += 256) {
      const int i = 1 + x / (L - 2), j = 1 + x - (i - 1) * (L - 2);
      if (j <= i) continue;
      const int u = s[i], v = s[j];
      if (u == 0 || v == 0) continue;
      if (rid[i] != rid[j]) {
        if (load[rid[i]] - dem[u] + dem[v] > capacity || load[rid[j]] - dem[v] + dem[u] > capacity) continue;
      }
      const int pa = s[i - 1], pg = s[j + 1];
      float delta;
      if (j == i + 1) {
        float nw = D(pa, v) + D(v, u);
        nw = nw + D(u, pg);
        float od = D(pa, u) + D(u, v);
        od = od + D(v, pg);
        delta = nw - od;
      } else {
        const int pc = s[i + 1], pe = s[j - 1];
        const float t1 = (D(pa, v) + D(v, pc)) - (D(pa, u) + D(u, pc));
        const float t2 = (D(pe, u) + D(u, pg)) - (D(pe, v) + D(v, pg));
        delta = t1 + t2;
      }
      const uint32_t code = (1u << 28) | ((uint32_t)i << 14) | (uint32_t)j;
      if (ls_better(delta, code, bd, bc)) { bd = delta; bc = code; }
    }
    // ---- 2-OPT inside a route: i < j, same route, both customers
    for (int x = tid; x < nsw; x += 256) {
      const int i = 1 + x / (L - 2), j = 1 + x - (i - 1) * (L - 2);
      if (j <= i) continue;
      const int u = s[i], v = s[j];
      if (u == 0 || v == 0 || rid[i] != rid[j]) continue;
      const int pa = s[i - 1], pg = s[j + 1];
      const float ends = (D(pa, v) + D(u, pg)) - (D(pa, u) + D(v, pg));
      const float inner = (float)(asym[j] - asym[i]);
      const float delta = ends + inner;
      const uint32_t code = (2u << 28) | ((uint32_t)i << 14) | (uint32_t)j;
      if (ls_better(delta, code, bd, bc)) { bd = delta; bc = code; }
    }
    // ---- workgroup minimum
    for (int o = 32; o >= 1; o >>= 1) {
      const float od = __shfl_xor(bd, o);
      const uint32_t oc = (uint32_t)__shfl_xor((int)bc, o);
      if (ls_better(od, oc, bd, bc)) { bd = od; bc = oc; }
    }
    if (lane == 0) { redd[wave] = bd; redc[wave] = bc; }
    __syncthreads();
    bd = redd[0]; bc = redc[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) if (ls_better(redd[w], redc[w], bd, bc)) { bd = redd[w]; bc = redc[w]; }
    if (!(bd < -1e-6f) || bc == 0xFFFFFFFFu) break;
    __syncthreads();
    // ---- apply (one thread: shifts of a few hundred 2-byte entries)
    if (tid == 0) {
      const int kind = (int)(bc >> 28), i = (int)((bc >> 14) & 0x3FFF), j = (int)(bc & 0x3FFF);
      if (kind == 0) {
        const uint16_t u = s[i];
        if (j > i) { for (int k = i; k < j; ++k) s[k] = s[k + 1]; s[j] = u; }
        else { for (int k = i; k > j + 1; --k) s[k] = s[k - 1]; s[j + 1] = u; }
        // the route u left may be empty now: drop the doubled depot
        int Lnew = 0;
        for (int k = 0; k < L; ++k) { if (k > 0 && s[k] == 0 && s[Lnew - 1] == 0) continue; s[Lnew++] = s[k]; }
        shared_i[0] = Lnew;
      } else if (kind == 1) {
        const uint16_t u = s[i]; s[i] = s[j]; s[j] = u;
      } else {
        for (int x = i, y = j; x < y; ++x, --y) { const uint16_t u = s[x]; s[x] = s[y]; s[y] = u; }
      }
    }
    __syncthreads();
    L = shared_i[0];
  }
  __syncthreads();
  for (int t = tid; t < Lmax; t += 256) col[(size_t)t * A] = t < L ? (int64_t)s[t] : 0;
  if (tid == 0) {
    if (lens_out) lens_out[blockIdx.x] = L;
    if (moves_out) moves_out[blockIdx.x] = moves;
  }
}

}  // namespace daco

using namespace daco;

extern "C" int daco_cvrp_local_search(void *stream, int B, int n, int A, int Lmax, const float *dist, long dist_bstride,
                                      const float *demand, float capacity, int64_t *paths, int max_moves, int32_t *lens,
                                      int32_t *moves) {
  if (B <= 0 || n < 2 || A <= 0 || Lmax < 3 || !dist || !demand || !paths || max_moves < 0) {
    set_error("daco_cvrp_local_search: bad argument (B=%d n=%d A=%d Lmax=%d)", B, n, A, Lmax);
    return DACO_E_BADARG;
  }
  if (Lmax > LS_MAXL || n > 16383) { set_error("daco_cvrp_local_search: Lmax=%d exceeds %d", Lmax, LS_MAXL); return DACO_E_TOOLARGE; }
  const int n4 = (n + 3) & ~3;
  const bool stage = n <= LS_STAGE_MAX_N;
  const size_t lds = (size_t)2 * LS_MAXL * 2 + (size_t)3 * LS_MAXL * 4 + (size_t)n4 * 4 + 12 * 4 + (stage ? (size_t)n * n * 4 : 0);
  hipStream_t s = (hipStream_t)stream;
  if (stage) {
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void *)cvrp_ls_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("daco_cvrp_local_search: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return DACO_E_HIP; }
    }
    hipLaunchKernelGGL((cvrp_ls_kernel<true>), dim3(B * A), dim3(256), lds, s, n, A, Lmax, dist, dist_bstride, demand, capacity, paths,
                       max_moves, lens, moves);
  } else {
    hipLaunchKernelGGL((cvrp_ls_kernel<false>), dim3(B * A), dim3(256), lds, s, n, A, Lmax, dist, dist_bstride, demand, capacity, paths,
                       max_moves, lens, moves);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("cvrp_ls_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}
