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
                 int64_t *edge_dst, float *edge_attr, int32_t *src32, int32_t *dst32) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wave;          // b*n + i
  if (row >= (long)B * n) return;
  const int b = (int)(row / n), i = (int)(row % n);
  const float *c = coords + (size_t)b * n * 2;
  const float xi = c[2 * i], yi = c[2 * i + 1];
  float dv[CPL];
  // (the row's coordinates first, all of them, unconditionally -- a lane past the row reads node 0 and drops it: read under
  // `if (j < n)` next to the store of the distance, every chunk's pair was a memory round trip of its own)
  float2 cjv[CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) { const int j = lane + 64 * q; cjv[q] = *reinterpret_cast<const float2 *>(c + 2 * (j < n ? j : 0)); }
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const int j = lane + 64 * q;
    const float2 cj = cjv[q];
    float v = __builtin_inff();
    if (j < n) {
      const float dx = xi - cj.x, dy = yi - cj.y;
      v = j == i ? diag : sqrtf(dx * dx + dy * dy);
      if (dist) dist[(size_t)row * n + j] = v;
    }
    dv[q] = v;
  }
  // round r's winner is kept by lane r % 64; every 64 rounds (and at the end) the lanes store their edges side by side
  float rkey = 0.0f;
  int ridx = 0;
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
    if (lane == (r & 63)) { rkey = w.key; ridx = w.idx; }
    if ((r & 63) == 63 || r == k - 1) {
      const int r0 = r & ~63;
      if (lane <= r - r0) {
        const size_t e = (size_t)row * k + r0 + lane;
        edge_src[e] = i;
        edge_dst[e] = ridx;
        edge_attr[e] = rkey;
        if (src32) { src32[e] = b * n + i; dst32[e] = b * n + ridx; }
      }
    }
  }
}

// ---- Net.reshape for a batch (tsp/net.py:94-102) with the "+ eps" of its callers (tsp/train.ipynb:35, tsp_nls/test.py:28):
// out[b] = fill everywhere, heu[b][e] + add at [src_e][dst_e].  Two launches on one stream: the fill, then one thread per edge.
__global__ void __launch_bounds__(256)
heu_fill_kernel(size_t total, float fill, float *out) {
  const size_t quads = total / 4;
  const float4 v = make_float4(fill, fill, fill, fill);
  float4 *out4 = reinterpret_cast<float4 *>(out);
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (size_t)gridDim.x * 256) out4[q] = v;
  if (blockIdx.x == 0 && quads * 4 + threadIdx.x < total) out[quads * 4 + threadIdx.x] = fill;     // (past the last whole float4)
}
__global__ void __launch_bounds__(256)
heu_scatter_kernel(int B, int n, int E, const int64_t *edge_index, const float *heu, float add, float *out, int32_t *bad) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)B * E) return;
  const int b = (int)(idx / E), e = (int)(idx - (size_t)b * E);
  const int64_t sidx = edge_index[((size_t)b * 2) * E + e], didx = edge_index[((size_t)b * 2 + 1) * E + e];
  if (sidx < 0 || sidx >= n || didx < 0 || didx >= n) { if (bad) atomicAdd(bad, 1); return; }
  out[((size_t)b * n + (size_t)sidx) * n + (size_t)didx] = heu[idx] + add;
}

}  // namespace daco

using namespace daco;

static int knn_graph_launch(const char *what, void *stream, int B, int n, int k, const float *coords, float diag, float *dist,
                            int64_t *edge_src, int64_t *edge_dst, float *edge_attr, int32_t *src32, int32_t *dst32) {
  if (B <= 0 || n < 2 || k < 1 || k > n || !coords || !edge_src || !edge_dst || !edge_attr || (src32 == nullptr) != (dst32 == nullptr)) {
    set_error("%s: bad argument (B=%d n=%d k=%d)", what, B, n, k);
    return DACO_E_BADARG;
  }
  if (n > DACO_MAX_NODES) { set_error("%s: n=%d exceeds DACO_MAX_NODES", what, n); return DACO_E_TOOLARGE; }
  if (src32 && (long)B * n > 0x7fffffffL) { set_error("%s: B * n = %ld node ids do not fit 32 bits", what, (long)B * n); return DACO_E_TOOLARGE; }
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)(((long)B * n + 3) / 4)), block(256);
  const int cpl = (n + 63) / 64;
#define DACO_KNN(C) hipLaunchKernelGGL(knn_graph_kernel<C>, grid, block, 0, s, B, n, k, coords, diag, dist, edge_src, edge_dst, edge_attr, src32, dst32)
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

extern "C" int daco_tsp_knn_graph(void *stream, int B, int n, int k, const float *coords, float diag, float *dist,
                                  int64_t *edge_src, int64_t *edge_dst, float *edge_attr) {
  return knn_graph_launch("daco_tsp_knn_graph", stream, B, n, k, coords, diag, dist, edge_src, edge_dst, edge_attr, nullptr, nullptr);
}

extern "C" int daco_tsp_knn_graph_csr(void *stream, int B, int n, int k, const float *coords, float diag, float *dist,
                                      int64_t *edge_src, int64_t *edge_dst, float *edge_attr, int32_t *src32, int32_t *dst32) {
  if (!src32 || !dst32) { set_error("daco_tsp_knn_graph_csr: src32 / dst32 missing"); return DACO_E_BADARG; }
  return knn_graph_launch("daco_tsp_knn_graph_csr", stream, B, n, k, coords, diag, dist, edge_src, edge_dst, edge_attr, src32, dst32);
}

extern "C" int daco_heu_matrix(void *stream, int B, int n, int E, const int64_t *edge_index, const float *heu, float fill, float add,
                               float *out, int32_t *bad) {
  if (B <= 0 || n < 1 || E < 0 || !edge_index || !heu || !out) {
    set_error("daco_heu_matrix: bad argument (B=%d n=%d E=%d)", B, n, E);
    return DACO_E_BADARG;
  }
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)B * n * n, quads = total / 4;
  if (((uintptr_t)out & 15) != 0) { set_error("daco_heu_matrix: out must be 16-byte aligned"); return DACO_E_BADARG; }
  {
    const size_t want = (quads + 255) / 256;
    hipLaunchKernelGGL(heu_fill_kernel, dim3((unsigned)(want < 1 ? 1 : (want < 8192 ? want : 8192))), dim3(256), 0, s, total, fill, out);
  }
  if (E > 0)                                                // (the scatter follows the fill on the same stream)
    hipLaunchKernelGGL(heu_scatter_kernel, dim3((unsigned)(((size_t)B * E + 255) / 256)), dim3(256), 0, s, B, n, E, edge_index, heu, add,
                       out, bad);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("daco_heu_matrix launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}
