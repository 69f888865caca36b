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
// index.  Layout: the tour (int32, with t[n] = t[0]) and the current edge lengths
// e[k] = d[t[k]][t[k+1]] live in LDS (they are re-read by every pair); the two distance-row
// gathers go to the L1/L2-resident matrix of the instance (rows t[i-1], t[i] are shared by a
// whole wave, so a wave's gathers hit two 4n-byte rows).  Wave w takes rows i = 1+w, 1+w+W, ..
// (interleaved: balances the triangular iteration space), lanes stride j.
#include "daco_device.h"
#include "../../include/deepaco_hip.h"

namespace daco {

template <int W>
__global__ void __launch_bounds__(64 * W)
two_opt_kernel(int n, int T, const float *dist, long dist_bs, uint16_t *tours, long max_iterations,
               int32_t *sweeps_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int *t = reinterpret_cast<int *>(smem);                 // n+1 ints
  float *e = reinterpret_cast<float *>(t + (n + 1 + 3) / 4 * 4);   // n floats
  float *redk = e + (n + 3) / 4 * 4;                      // W keys
  int *redi = reinterpret_cast<int *>(redk + W);          // W indices
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / T;
  const float *d = dist + (size_t)b * dist_bs;
  uint16_t *tour = tours + (size_t)blockIdx.x * n;

  for (int k = tid; k < n; k += 64 * W) t[k] = tour[k];
  __syncthreads();
  if (tid == 0) t[n] = t[0];
  __syncthreads();
  for (int k = tid; k < n; k += 64 * W) e[k] = d[(size_t)t[k] * n + t[k + 1]];
  __syncthreads();

  long it = 0;
  while (it < max_iterations) {
    // ---- one sweep: lane-local best over this wave's rows
    float bk = 0.0f;                 // delta = 0: only improving moves qualify
    int bi = 0x7fffffff;
    for (int i = 1 + wave; i < n - 1; i += W) {
      const int na = t[i - 1], nb = t[i];
      const float *rowA = d + (size_t)na * n, *rowB = d + (size_t)nb * n;
      const float eab = e[i - 1];
      for (int j = i + 1 + lane; j < n; j += 64) {
        const int nc = t[j], nd = t[j + 1];
        if (na == nc || nd == nb) continue;
        float change = rowA[nc] + rowB[nd];
        change = change - eab;
        change = change - e[j];
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
    // ---- reverse t[p..q]
    const int half = (q - p + 1) >> 1;
    for (int k = tid; k < half; k += 64 * W) {
      const int x = t[p + k];
      t[p + k] = t[q - k];
      t[q - k] = x;
    }
    __syncthreads();
    if (tid == 0) t[n] = t[0];
    __syncthreads();
    // ---- refresh the edge lengths the reversal touched: k = p-1 .. q
    for (int k = p - 1 + tid; k <= q; k += 64 * W) e[k] = d[(size_t)t[k] * n + t[k + 1]];
    __syncthreads();
  }
  for (int k = tid; k < n; k += 64 * W) tour[k] = (uint16_t)t[k];
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
  const size_t lds = ((size_t)(n + 1 + 3) / 4 * 4 + (size_t)(n + 3) / 4 * 4 + 8) * 4;
  hipStream_t s = (hipStream_t)stream;
  if (n <= 128)
    hipLaunchKernelGGL(two_opt_kernel<1>, dim3(B * T), dim3(64), lds, s, n, T, dist, dist_bstride, tours, max_iterations, sweeps);
  else
    hipLaunchKernelGGL(two_opt_kernel<4>, dim3(B * T), dim3(256), lds, s, n, T, dist, dist_bstride, tours, max_iterations, sweeps);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("two_opt_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}
