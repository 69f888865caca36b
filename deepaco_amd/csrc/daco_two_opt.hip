// daco_two_opt.hip -- best-improvement 2-opt local search, one workgroup per tour.
//
// Reference behaviour replaced: tsp_nls/two_opt.py:6-49 (numba two_opt_once /
// _two_opt_python / batched_two_opt_python, one CPU thread-pool task per tour).
//
// Per sweep every pair 1 <= i < j <= n-1 is evaluated with the reference's expression
//   change = d[t[i-1]][t[j]] + d[t[i]][t[j+1]] - d[t[i-1]][t[i]] - d[t[j]][t[j+1]]
// in f32, left to right (no FMA), and the strict minimum in row-major (i,j) order wins; if
// it is below -1e-6 the segment t[i..j] is reversed.  Results are bit-identical to the
// reference: the same four loads, the same three roundings, ties broken on the flattened
// index.  Layout: the tour lives in LDS as one 8-byte record per position (node, successor, edge
// length) and, per wave, the two distance rows d[t[i-1]][.], d[t[i]][.] of the row it is working on;
// each wave owns a contiguous chunk of rows i (equal pair counts), lanes stride j.
#include <cstdlib>

#include "daco_device.h"
#include "../../include/deepaco_hip.h"

namespace daco {

// Row staging: a wave owns a CONTIGUOUS chunk of rows i (chunks sized for equal pair counts),
// so the distance row it gathered from as d[t[i]][.] becomes the d[t[i-1]][.] row of the next i:
// one new 4n-byte row is copied into the wave's private LDS slot per i, and both gathers of
// every pair are LDS reads (rocprof showed the global-gather version bound by the texture
// addresser at 86 % TA busy; LDS serves a 64-lane random gather ~3x faster).
template <int W, bool STAGE>
__global__ void __launch_bounds__(64 * W)
two_opt_kernel(int n, int T, const float *dist, long dist_bs, uint16_t *tours, long max_iterations,
               int32_t *sweeps_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int np4 = (n + 3) / 4 * 4;
  // the tour as one 8-byte record per position k: {t[k] | t[k+1] << 16, e[k] = d[t[k]][t[k+1]]}
  // (t[n] = t[0]); a pair (i,j) needs record i-1 (once per row) and record j (one ds_read_b64)
  int2 *pe = reinterpret_cast<int2 *>(smem);              // np4 records
  float *redk = reinterpret_cast<float *>(pe + np4);      // W keys
  int *redi = reinterpret_cast<int *>(redk + W);          // W indices
  float *rows = reinterpret_cast<float *>(redi + W + (8 - 2 * W % 8) % 8);   // W x 2 x np4 floats
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / T;
  const float *d = dist + (size_t)b * dist_bs;
  uint16_t *tour = tours + (size_t)blockIdx.x * n;
  float *myrows = rows + (size_t)wave * 2 * np4;

  for (int k = tid; k < n; k += 64 * W) {
    const int u = tour[k], v = tour[k + 1 < n ? k + 1 : 0];
    pe[k] = make_int2(u | (v << 16), __float_as_int(d[(size_t)u * n + v]));
  }
  __syncthreads();

  // equal-area split of the triangular pair space over the W waves: rows [ilo, ihi)
  // pairs in rows >= i ~ (n-i)^2/2  ->  boundary_k = n - (n-1)*sqrt(1 - k/W)
  int ilo = 1, ihi = n - 1;
  if (W > 1) {
    const float m = (float)(n - 1);
    ilo = wave == 0 ? 1 : (int)(n - m * sqrtf(1.0f - (float)wave / W));
    ihi = wave == W - 1 ? n - 1 : (int)(n - m * sqrtf(1.0f - (float)(wave + 1) / W));
    ilo = max(1, min(ilo, n - 1));
    ihi = max(ilo, min(ihi, n - 1));
  }
  auto stage = [&](float *dst, int node) {                // copy row d[node][0..n) into the wave's slot
    const float *src = d + (size_t)node * n;
    for (int k = lane; k < n; k += 64) dst[k] = src[k];
  };

  long it = 0;
  while (it < max_iterations) {
    // ---- one sweep: lane-local best over this wave's rows (rows visited in increasing order)
    float bk = 0.0f;                 // delta = 0: only improving moves qualify
    int bi = 0x7fffffff;
    float *rowA = myrows, *rowB = myrows + np4;
    if (STAGE && ilo < ihi) stage(rowB, pe[ilo - 1].x & 0xFFFF);   // becomes rowA of the first row
    for (int i = ilo; i < ihi; ++i) {
      const int2 ri = pe[i - 1];
      const int na = ri.x & 0xFFFF, nb = (unsigned)ri.x >> 16;      // t[i-1], t[i]
      const float eab = __int_as_float(ri.y);                       // d[t[i-1]][t[i]]
      const float *gA, *gB;
      if constexpr (STAGE) {
        float *tmp = rowA; rowA = rowB; rowB = tmp;       // d[t[i-1]][.] was staged as the previous rowB
        stage(rowB, nb);
        gA = rowA; gB = rowB;
      } else {
        gA = d + (size_t)na * n; gB = d + (size_t)nb * n;  // small n: the rows sit in L1 anyway
      }
      for (int j = i + 1 + lane; j < n; j += 64) {
        const int2 rec = pe[j];
        const int nc = rec.x & 0xFFFF, nd = (unsigned)rec.x >> 16;
        if (na == nc || nd == nb) continue;
        float change = gA[nc] + gB[nd];
        change = change - eab;
        change = change - __int_as_float(rec.y);
        if (change < bk) { bk = change; bi = i * n + j; }
      }
    }
    const KeyIdx r = wave_arg<false>(bk, bi);
    if (lane == 0) { redk[wave] = r.key; redi[wave] = r.idx; }
    __syncthreads();
    float gk = redk[0];
    int gi = redi[0];
#pragma unroll
    for (int w = 1; w < W; ++w) {
      const float k2 = redk[w];
      const int i2 = redi[w];
      if (k2 < gk || (k2 == gk && i2 < gi)) { gk = k2; gi = i2; }
    }
    ++it;
    if (!((double)gk < -1e-6)) break;                     // no improving move: converged
    const int p = gi / n, q = gi - p * n;
    __syncthreads();                                      // everyone has read redk/redi and the records
    // ---- reverse t[p..q]: swap the node halves of the records, then rebuild successors and edges
    const int half = (q - p + 1) >> 1;
    for (int k = tid; k < half; k += 64 * W) {
      const int x = pe[p + k].x, y = pe[q - k].x;
      pe[p + k].x = (x & 0xFFFF0000) | (y & 0xFFFF);
      pe[q - k].x = (y & 0xFFFF0000) | (x & 0xFFFF);
    }
    __syncthreads();
    for (int k = p - 1 + tid; k <= q; k += 64 * W) {      // records p-1 .. q: successor + edge length
      const int u = pe[k].x & 0xFFFF;                     // node halves are final (barrier above); the
      const int v = pe[k + 1 < n ? k + 1 : 0].x & 0xFFFF; // 16-bit store below never touches them
      const float len = d[(size_t)u * n + v];
      pe[k].y = __float_as_int(len);
      reinterpret_cast<unsigned short *>(&pe[k].x)[1] = (unsigned short)v;
    }
    __syncthreads();
  }
  for (int k = tid; k < n; k += 64 * W) tour[k] = (uint16_t)(pe[k].x & 0xFFFF);
  if (sweeps_out && tid == 0) sweeps_out[blockIdx.x] = (int32_t)it;
}

}  // namespace daco

using namespace daco;

extern "C" int daco_two_opt(void *stream, int B, int T, int n, const float *dist, long dist_bstride,
                            uint16_t *tours, long max_iterations, int32_t *sweeps) {
  if (B <= 0 || T <= 0 || n < 4 || !dist || !tours || max_iterations < 0) {
    set_error("daco_two_opt: bad argument (B=%d T=%d n=%d)", B, T, n);
    return DACO_E_BADARG;
  }
  if (n > DACO_MAX_NODES) { set_error("daco_two_opt: n=%d exceeds DACO_MAX_NODES", n); return DACO_E_TOOLARGE; }
  const size_t np4 = (size_t)(n + 3) / 4 * 4;
  auto lds_bytes = [&](int W, bool stage) {
    return (2 * np4 + 2 * W + (8 - 2 * W % 8) % 8 + (stage ? (size_t)W * 2 * np4 : 0)) * 4;
  };
  hipStream_t s = (hipStream_t)stream;
  // variant = waves per tour * 2 + staged; chosen by size (measured on MI355X, tools/bench_two_opt.py);
  // DACO_TWO_OPT_VARIANT overrides it for tuning runs
  int variant = n <= 128 ? 2 : (n <= 256 ? 8 : 9);
  if (const char *ev = getenv("DACO_TWO_OPT_VARIANT")) variant = atoi(ev);
#define DACO_2OPT(W, ST) hipLaunchKernelGGL((two_opt_kernel<W, ST>), dim3(B * T), dim3(64 * W), lds_bytes(W, ST), s, n, T, dist, dist_bstride, tours, max_iterations, sweeps)
  switch (variant) {
    case 2: DACO_2OPT(1, false); break;
    case 3: DACO_2OPT(1, true); break;
    case 4: DACO_2OPT(2, false); break;
    case 5: DACO_2OPT(2, true); break;
    case 8: DACO_2OPT(4, false); break;
    default: DACO_2OPT(4, true); break;
  }
#undef DACO_2OPT
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("two_opt_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}
