// daco_costs_update.hip -- tour costs and the fused evaporate + deposit + clamp kernel.
//
// Reference behaviour replaced: ACO.gen_path_costs (tsp/aco.py:121-132, cvrp/aco.py:133-136)
// and ACO.update_pheronome (tsp/aco.py:95-118, cvrp/aco.py:107-130).
//
// The reference deposits with a sequential Python loop over ants, so the f32 result depends
// on ant order.  Here every row i of tau is owned by one lane pair: lane 2r adds w_a to column
// prev_a(i), lane 2r+1 to column next_a(i), for a = 0..A-1 in order, on a copy of the row
// held in LDS.  Adds to one tau element therefore happen in ant order exactly as in the
// reference, adds to different elements never interact, and no atomics are needed
// (probe-verified bit-identical).  Evaporation is fused into the
// LDS fill and the MMAS clamp / floor into the write-back, so tau makes one round trip.
#include "daco_device.h"
#include "../../include/deepaco_hip.h"

namespace daco {

__global__ void __launch_bounds__(256)
tour_costs_kernel(int B, int n, int len, int A, const float *dist, long dist_bs, const int64_t *paths,
                  int closed, float *costs) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * A) return;
  const int b = idx / A, a = idx - b * A;
  const int64_t *p = paths + (size_t)b * len * A + a;
  const float *d = dist + b * dist_bs;
  float s = 0.0f;
  if (closed) {
    // edges k = 1..len-1 in order, closing edge last (the order the fused sampler produces)
    const long first = p[0];
    long prev = first;
    int k = 1;
    for (; k + 8 <= len; k += 8) {
      long u[8];
      float e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) u[j] = p[(size_t)(k + j) * A];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = d[u[j] * n + (j ? u[j - 1] : prev)];
#pragma unroll
      for (int j = 0; j < 8; ++j) s = s + e[j];
      prev = u[7];
    }
    for (; k < len; ++k) {
      const long u = p[(size_t)k * A];
      s = s + d[u * n + prev];
      prev = u;
    }
    s = s + d[first * n + prev];
  } else {
    long u = p[0];
    for (int k = 0; k + 1 < len; ++k) {
      const long v = p[(size_t)(k + 1) * A];
      s = s + d[u * n + v];
      u = v;
    }
  }
  costs[idx] = s;
}

// nbr[b][node][a] = prev(node) | next(node) << 16 along ant a's closed tour
__global__ void __launch_bounds__(256)
build_nbr_kernel(int B, int n, int A, const int64_t *paths, uint32_t *nbr) {
  const long total = (long)B * n * A;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int a = (int)(i % A);
    const long r = i / A;
    const int k = (int)(r % n), b = (int)(r / n);
    const int64_t *p = paths + (size_t)b * n * A + a;
    const uint32_t u = (uint32_t)p[(size_t)k * A];
    const uint32_t up = (uint32_t)p[(size_t)(k ? k - 1 : n - 1) * A];
    const uint32_t un = (uint32_t)p[(size_t)(k + 1 < n ? k + 1 : 0) * A];
    nbr[((size_t)b * n + u) * A + a] = up | (un << 16);
  }
}

// first index of the minimum cost per instance (torch.min(dim=0) semantics)
__global__ void __launch_bounds__(64) argmin_cost_kernel(int A, const float *costs, int *best) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float bk = __builtin_inff();
  int bi = 0x7fffffff;
  for (int a = lane; a < A; a += 64) {
    const float c = costs[(size_t)b * A + a];
    if (c < bk) { bk = c; bi = a; }
  }
  const KeyIdx r = wave_arg<false>(bk, bi);
  if (lane == 0) best[b] = r.idx == 0x7fffffff ? 0 : r.idx;
}

// Row owners: a workgroup keeps R rows of tau in LDS.  Row i receives, per ant and in ant order,
// +w at column prev_a(i) and +w at column next_a(i) (tsp/aco.py:95-118: each ant's forward and
// backward edges).  Adds to one element must stay in ant order, so each row is a chain of
// dependent LDS read-modify-writes: two lanes per row (prev side / next side; they never meet on
// a column within one ant) walk the ants in lock step.  The chain is LDS-latency bound, so the
// rest is kept off it: the [node][ant] table arrives in coalesced chunks of DEP_CHUNK ants,
// double-buffered in LDS by all four waves, and the chain reads it four ants at a time.
constexpr int DEP_CHUNK = 64, DEP_ROWS = 32;

__global__ void __launch_bounds__(256)
deposit_tsp_kernel(int n, int A, int R, float *tau, const uint32_t *nbr, const float *costs, const float *weights,
                   float decay, const int *best, const float *clamp_min, const float *clamp_max, float floor_val) {
  extern __shared__ __attribute__((aligned(16))) float rows[];
  __shared__ __attribute__((aligned(16))) uint32_t stage[2][DEP_ROWS][DEP_CHUNK];
  __shared__ __attribute__((aligned(16))) float wts[2][DEP_CHUNK];
  const int bpi = (n + R - 1) / R;
  const int b = blockIdx.x / bpi;
  const int i0 = (blockIdx.x - b * bpi) * R;
  const int Rv = min(R, n - i0);
  float *g = tau + ((size_t)b * n + i0) * n;
  const int cnt = Rv * n;
  int alo = 0, ahi = A;
  if (best) { alo = best[b]; ahi = alo + 1; }
  const uint32_t *tab = nbr + ((size_t)b * n + i0) * A;            // rows i0.. of this instance's [n][A] table
  const float *cs = costs + (size_t)b * A, *wt = weights ? weights + (size_t)b * A : nullptr;
  // chunk loader: consecutive threads on consecutive ants of one row (256-byte segments)
  auto load_chunk = [&](int c0, int buf) {
    const int m = min(DEP_CHUNK, ahi - c0);
    for (int i = threadIdx.x; i < Rv * DEP_CHUNK; i += blockDim.x) {
      const int r = i / DEP_CHUNK, j = i - r * DEP_CHUNK;
      stage[buf][r][j] = j < m ? tab[(size_t)r * A + c0 + j] : 0u;
    }
    if (threadIdx.x < DEP_CHUNK)
      wts[buf][threadIdx.x] = threadIdx.x < m ? (wt ? wt[c0 + threadIdx.x] : 1.0f / cs[c0 + threadIdx.x]) : 0.0f;
  };
  load_chunk(alo, 0);
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) rows[i] = g[i] * decay;
  __syncthreads();
  const int r = threadIdx.x >> 1, sh = (threadIdx.x & 1) << 4;     // prev side: low half-word, next side: high
  float *row = rows + (r < Rv ? r : 0) * n;
  int buf = 0;
  for (int c0 = alo; c0 < ahi; c0 += DEP_CHUNK, buf ^= 1) {
    if (c0 + DEP_CHUNK < ahi) load_chunk(c0 + DEP_CHUNK, buf ^ 1);  // next chunk lands while this one is applied
    if (r < Rv) {
      const int m = min(DEP_CHUNK, ahi - c0);
      int j = 0;
      for (; j + 4 <= m; j += 4) {
        const uint4 v = *(const uint4 *)&stage[buf][r][j];
        const float4 w = *(const float4 *)&wts[buf][j];
        const int c0_ = (v.x >> sh) & 0xFFFFu, c1_ = (v.y >> sh) & 0xFFFFu, c2_ = (v.z >> sh) & 0xFFFFu, c3_ = (v.w >> sh) & 0xFFFFu;
        row[c0_] = row[c0_] + w.x;
        row[c1_] = row[c1_] + w.y;
        row[c2_] = row[c2_] + w.z;
        row[c3_] = row[c3_] + w.w;
      }
      for (; j < m; ++j) {
        const int col = (stage[buf][r][j] >> sh) & 0xFFFFu;
        row[col] = row[col] + wts[buf][j];
      }
    }
    __syncthreads();
  }
  const bool clamp = clamp_max != nullptr;
  const float cmin = clamp ? clamp_min[b] : 0.0f, cmax = clamp ? clamp_max[b] : 0.0f;
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
    float x = rows[i];
    if (clamp) { x = x < cmin ? cmin : x; x = x > cmax ? cmax : x; }
    if (floor_val > 0.0f) x = x < floor_val ? floor_val : x;
    g[i] = x;
  }
}

// ------------------------------------------------------------------ directed deposit (CVRP and siblings)
// cvrp/aco.py:107-130: tau[path[:-1], path[1:]] += 1/cost per ant, duplicates of an index pair
// within one ant (the padding edge (0,0)) collapse to ONE add.  Row i >= 1 (a customer) is
// left exactly once per ant -> one lane per row walks the ants in order.  Row 0 (the depot) is
// left once per route: build_next_kernel also collects, per ant, the list of nodes that
// follow the depot; the depot workgroup applies each ant's list (distinct columns -> parallel
// lanes) in ant order, and the (0,0) edge once per ant if it occurs.
__global__ void __launch_bounds__(256)
build_next_kernel(int B, int n, int len, int A, int hub, const int64_t *paths, uint32_t *nbr, uint16_t *dlist, int *dcnt) {
  const long total = (long)B * (len - 1) * A;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int a = (int)(i % A);
    const long r = i / A;
    const int k = (int)(r % (len - 1)), b = (int)(r / (len - 1));
    const int64_t *p = paths + (size_t)b * len * A + a;
    const uint32_t u = (uint32_t)p[(size_t)k * A], v = (uint32_t)p[(size_t)(k + 1) * A];
    if ((int)u != hub) nbr[((size_t)b * A + a) * n + u] = v << 16;
    else {
      const int pos = atomicAdd(dcnt + (size_t)b * A + a, 1);
      dlist[((size_t)b * A + a) * len + pos] = (uint16_t)v;
    }
  }
}

__device__ inline float ant_weight(const float *weights, const float *costs, size_t idx) {
  return weights ? weights[idx] : 1.0f / costs[idx];
}

// rows other than the hub: at most one outgoing edge per ant (0xFFFF in the table = none)
__global__ void __launch_bounds__(256)
deposit_directed_kernel(int n, int A, int R, int hub, float *tau, const uint32_t *nbr, const float *costs,
                        const float *weights, float decay, const int *best, const float *clamp_min,
                        const float *clamp_max, float floor_val) {
  extern __shared__ __attribute__((aligned(16))) float rows[];
  const int bpi = (n + R - 1) / R;
  const int b = blockIdx.x / bpi;
  const int i0 = (blockIdx.x - b * bpi) * R;
  const int Rv = min(R, n - i0);
  float *g = tau + ((size_t)b * n + i0) * n;
  const int cnt = Rv * n;
  const int hub_lo = (hub - i0) * n, hub_hi = hub_lo + n;      // the hub row belongs to the hub kernel
  for (int i = threadIdx.x; i < cnt; i += blockDim.x)
    if (i < hub_lo || i >= hub_hi) rows[i] = g[i] * decay;
  __syncthreads();
  const int r = threadIdx.x;
  if (r < Rv && i0 + r != hub) {
    const uint32_t *nb = nbr + (size_t)b * A * n + i0 + r;
    float *row = rows + r * n;
    int alo = 0, ahi = A;
    if (best) { alo = best[b]; ahi = alo + 1; }
#pragma unroll 4
    for (int a = alo; a < ahi; ++a) {
      const uint32_t col = nb[(size_t)a * n] >> 16;
      if (col != 0xFFFFu) row[col] = row[col] + ant_weight(weights, costs, (size_t)b * A + a);
    }
  }
  __syncthreads();
  const bool clamp = clamp_max != nullptr;
  const float cmin = clamp ? clamp_min[b] : 0.0f, cmax = clamp ? clamp_max[b] : 0.0f;
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
    if (i >= hub_lo && i < hub_hi) continue;
    float x = rows[i];
    if (clamp) { x = x < cmin ? cmin : x; x = x > cmax ? cmax : x; }
    if (floor_val > 0.0f) x = x < floor_val ? floor_val : x;
    g[i] = x;
  }
}

// the hub row (depot / dummy node): one add per route start, (hub,hub) at most once per ant
__global__ void __launch_bounds__(64)
deposit_hub_kernel(int n, int len, int A, int hub, float *tau, const uint16_t *dlist, const int *dcnt, const float *costs,
                   const float *weights, float decay, const int *best, const float *clamp_min, const float *clamp_max,
                   float floor_val) {
  extern __shared__ __attribute__((aligned(16))) float row[];
  const int b = blockIdx.x, lane = threadIdx.x;
  float *g = tau + ((size_t)b * n + hub) * n;
  for (int i = lane; i < n; i += 64) row[i] = g[i] * decay;
  __syncthreads();
  int alo = 0, ahi = A;
  if (best) { alo = best[b]; ahi = alo + 1; }
  for (int a = alo; a < ahi; ++a) {
    const int c = dcnt[(size_t)b * A + a];
    const float w = ant_weight(weights, costs, (size_t)b * A + a);
    const uint16_t *lst = dlist + ((size_t)b * A + a) * len;
    bool self = false;
    for (int j0 = 0; j0 < c; j0 += 64) {
      const int j = j0 + lane;
      const int v = j < c ? (int)lst[j] : -1;
      if (v >= 0 && v != hub) row[v] = row[v] + w;      // distinct successors: no conflicts
      self = self || v == hub;
    }
    if (__ballot(self) != 0 && lane == 0) row[hub] = row[hub] + w;   // (hub,hub) collapses to one add
    __syncthreads();
  }
  const bool clamp = clamp_max != nullptr;
  const float cmin = clamp ? clamp_min[b] : 0.0f, cmax = clamp ? clamp_max[b] : 0.0f;
  for (int i = lane; i < n; i += 64) {
    float x = row[i];
    if (clamp) { x = x < cmin ? cmin : x; x = x > cmax ? cmax : x; }
    if (floor_val > 0.0f) x = x < floor_val ? floor_val : x;
    g[i] = x;
  }
}

}  // namespace daco

using namespace daco;

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" int daco_tour_costs(void *stream, int B, int n, int len, int A, const float *dist,
                               long dist_bstride, const int64_t *paths, int closed, float *costs) {
  if (B <= 0 || n <= 0 || len <= 0 || A <= 0 || !dist || !paths || !costs) {
    set_error("daco_tour_costs: bad argument");
    return DACO_E_BADARG;
  }
  const int total = B * A;
  hipLaunchKernelGGL(tour_costs_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, n,
                     len, A, dist, dist_bstride, paths, closed, costs);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("tour_costs_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}

extern "C" size_t daco_pheromone_update_workspace_bytes(int B, int n, int len, int A) {
  if (B <= 0 || n <= 0 || A <= 0 || len <= 0) return 0;
  // nbr table | best-ant index | (directed) depot successor lists + their counters
  return align256((size_t)B * A * n * sizeof(uint32_t)) + align256((size_t)B * sizeof(int)) +
         align256((size_t)B * A * len * sizeof(uint16_t)) + align256((size_t)B * A * sizeof(int));
}

static int rows_per_block(int n) {
  // two workgroups per CU: 160 KiB of LDS less the two static staging images (2 x 16.5 KiB)
  int R = (63 * 1024) / (4 * n);
  if (R > DEP_ROWS) R = DEP_ROWS;     // two lanes per row, the chains of a workgroup run in one wave
  return R;
}

extern "C" int daco_pheromone_update(void *stream, int B, int n, int len, int A, float *tau,
                                     const int64_t *paths, const float *costs, float decay, int elitist,
                                     int symmetric, const float *clamp_min, const float *clamp_max,
                                     float floor_val, const uint32_t *nbr_in, const float *weights, int hub,
                                     void *workspace, size_t workspace_bytes) {
  if (B <= 0 || n < 3 || A <= 0 || !tau || !paths || !costs || !workspace) {
    set_error("daco_pheromone_update: bad argument (B=%d n=%d A=%d)", B, n, A);
    return DACO_E_BADARG;
  }
  if ((clamp_min == nullptr) != (clamp_max == nullptr)) { set_error("daco_pheromone_update: clamp_min/clamp_max must both be given"); return DACO_E_BADARG; }
  if (n > DACO_MAX_NODES) { set_error("daco_pheromone_update: n=%d exceeds DACO_MAX_NODES", n); return DACO_E_TOOLARGE; }
  if (symmetric && len != n) { set_error("daco_pheromone_update: symmetric deposit needs len == n"); return DACO_E_BADARG; }
  if (hub >= n) { set_error("daco_pheromone_update: hub %d >= n %d", hub, n); return DACO_E_BADARG; }
  if (!symmetric && len < 2) { set_error("daco_pheromone_update: directed deposit needs len >= 2"); return DACO_E_BADARG; }
  if (!symmetric && nbr_in) { set_error("daco_pheromone_update: nbr input is for the symmetric deposit only"); return DACO_E_BADARG; }
  const size_t need = daco_pheromone_update_workspace_bytes(B, n, len, A);
  if (workspace_bytes < need) { set_error("daco_pheromone_update: workspace %zu < %zu", workspace_bytes, need); return DACO_E_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  const uint32_t *nbr = nbr_in ? nbr_in : (const uint32_t *)workspace;
  int *best = (int *)((char *)workspace + align256((size_t)B * A * n * sizeof(uint32_t)));
  if (!symmetric) {
    uint16_t *dlist = (uint16_t *)((char *)best + align256((size_t)B * sizeof(int)));
    int *dcnt = (int *)((char *)dlist + align256((size_t)B * A * len * sizeof(uint16_t)));
    if (hipMemsetAsync(dcnt, 0, (size_t)B * A * sizeof(int), s) != hipSuccess) { set_error("hipMemsetAsync failed"); return DACO_E_HIP; }
    if (hipMemsetAsync(workspace, 0xFF, (size_t)B * A * n * sizeof(uint32_t), s) != hipSuccess) { set_error("hipMemsetAsync failed"); return DACO_E_HIP; }
    const long total = (long)B * (len - 1) * A;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(build_next_kernel, dim3(blocks), dim3(256), 0, s, B, n, len, A, hub, paths, (uint32_t *)workspace, dlist, dcnt);
    if (elitist) hipLaunchKernelGGL(argmin_cost_kernel, dim3(B), dim3(64), 0, s, A, costs, best);
    int R = (64 * 1024) / (4 * n);
    if (R > 256) R = 256;
    const int bpi = (n + R - 1) / R;
    if (hub >= 0)
      hipLaunchKernelGGL(deposit_hub_kernel, dim3(B), dim3(64), (size_t)n * sizeof(float), s, n, len, A, hub, tau, dlist, dcnt,
                         costs, weights, decay, elitist ? best : nullptr, clamp_min, clamp_max, floor_val);
    hipLaunchKernelGGL(deposit_directed_kernel, dim3(B * bpi), dim3(256), (size_t)R * n * sizeof(float), s, n, A, R, hub, tau,
                       (const uint32_t *)workspace, costs, weights, decay, elitist ? best : nullptr, clamp_min, clamp_max,
                       floor_val);
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) { set_error("directed pheromone update launch: %s", hipGetErrorString(e2)); return DACO_E_HIP; }
    return DACO_OK;
  }
  if (!nbr_in) {
    const long total = (long)B * n * A;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(build_nbr_kernel, dim3(blocks), dim3(256), 0, s, B, n, A, paths, (uint32_t *)workspace);
  }
  if (elitist) hipLaunchKernelGGL(argmin_cost_kernel, dim3(B), dim3(64), 0, s, A, costs, best);
  const int R = rows_per_block(n);
  const int bpi = (n + R - 1) / R;
  hipLaunchKernelGGL(deposit_tsp_kernel, dim3(B * bpi), dim3(256), (size_t)R * n * sizeof(float), s, n, A, R, tau,
                     nbr, costs, weights, decay, elitist ? best : nullptr, clamp_min, clamp_max, floor_val);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("pheromone update launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}
