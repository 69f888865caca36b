// daco_graph.hip -- batched instance -> graph construction for the TSP family.
//
// Reference behaviour replaced: gen_distance_matrix + gen_pyg_data, tsp/utils.py:4-36 and
// tsp_nls/utils.py:5-45 (one instance at a time: norm of coordinate differences, diagonal 1e9,
// torch.topk(k, largest=False) per row, edge_index = [repeat_interleave(arange n, k); topk indices],
// edge_attr = topk values).  This is the step immediately before Net.forward; here it
// is one launch for B instances: one wavefront per node builds its distance row (kept in registers,
// lane l owns columns l, l+64, ...) and extracts the k nearest by k rounds of a wave arg-min
// (ties -> smaller index).  Output order = ascending distance, like topk's sorted result.
#include "daco_device.h"
#include "../../include/deepaco_hip.h"

namespace daco {

template <int CPL>   // columns per lane
__global__ void __launch_bounds__(256)
knn_graph_kernel(int B, int n, int k, const float *coords, float diag, float *dist, int64_t *edge_src,
                 int64_t *edge_dst, float *edge_attr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wave;          // b*n + i
  if (row >= (long)B * n) return;
  const int b = (int)(row / n), i = (int)(row % n);
  const float *c = coords + (size_t)b * n * 2;
  const float xi = c[2 * i], yi = c[2 * i + 1];
  float dv[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const int j = lane + 64 * q;
    float v = __builtin_inff();
    if (j < n) {
      const float dx = xi - c[2 * j], dy = yi - c[2 * j + 1];
      v = j == i ? diag : sqrtf(dx * dx + dy * dy);
      if (dist) dist[(size_t)row * n + j] = v;
    }
    dv[q] = v;
  }
  for (int r = 0; r < k; ++r) {
    float bk = __builtin_inff();
    int bj = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < CPL; ++q)
      if (dv[q] < bk) { bk = dv[q]; bj = lane + 64 * q; }
    const KeyIdx w = wave_arg<false>(bk, bj);
#pragma unroll
    for (int q = 0; q < CPL; ++q)
      if (lane + 64 * q == w.idx) dv[q] = __builtin_inff();          // taken
    if (lane == 0) {
      const size_t e = (size_t)row * k + r;
      edge_src[e] = i;
      edge_dst[e] = w.idx;
      edge_attr[e] = w.key;
    }
  }
}

}  // namespace daco

using namespace daco;

extern "C" int daco_tsp_knn_graph(void *stream, int B, int n, int k, const float *coords, float diag, float *dist,
                                  int64_t *edge_src, int64_t *edge_dst, float *edge_attr) {
  if (B <= 0 || n < 2 || k < 1 || k > n || !coords || !edge_src || !edge_dst || !edge_attr) {
    set_error("daco_tsp_knn_graph: bad argument (B=%d n=%d k=%d)", B, n, k);
    return DACO_E_BADARG;
  }
  if (n > DACO_MAX_NODES) { set_error("daco_tsp_knn_graph: n=%d exceeds DACO_MAX_NODES", n); return DACO_E_TOOLARGE; }
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)(((long)B * n + 3) / 4)), block(256);
  const int cpl = (n + 63) / 64;
#define DACO_KNN(C) hipLaunchKernelGGL(knn_graph_kernel<C>, grid, block, 0, s, B, n, k, coords, diag, dist, edge_src, edge_dst, edge_attr)
  if (cpl <= 2) DACO_KNN(2);
  else if (cpl <= 4) DACO_KNN(4);
  else if (cpl <= 8) DACO_KNN(8);
  else if (cpl <= 16) DACO_KNN(16);
  else if (cpl <= 32) DACO_KNN(32);
  else DACO_KNN(64);
#undef DACO_KNN
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("knn_graph_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}
