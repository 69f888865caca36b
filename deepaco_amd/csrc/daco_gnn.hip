// daco_gnn.hip -- heuristic network forward (inference): 12-layer edge GNN + MLP head.
//
// Reference behaviour replaced: EmbNet.forward tsp/net.py:27-45, MLP/ParNet.forward :59-66,74-75,
// Net.forward :84-88 (and the feats=1 variants in tsp_nls/net.py, cvrp/net.py), eval mode
// (BatchNorm with running statistics, folded into a per-channel scale/shift by the host).
//
// The reference issues ~25 aten/PyG ops per layer.  Here one launch per layer does everything:
//   edge workgroups : w1 = We*w0 + be on the matrix cores (v_mfma_f32_32x32x2_f32: a 32-edge x
//                     32-channel tile per wave, K = 32 -> 16 MFMAs, exact f32), then in the MFMA
//                     output layout  w' = w0 + silu(bn_e(w1 + x3[src] + x4[dst]))
//   node workgroups : agg_i = mean over i's out-edges of sigmoid(w0_e)*x2[dst_e] (CSR order, no
//                     atomics), x' = x0 + silu(bn_v(x1 + agg)), and -- fused -- the NEXT layer's
//                     four node linears x1..x4 = W*x' + b, so no separate node-linear launch.
// Activations ping-pong between two workspace buffers.  The head (3 linears, silu, silu, sigmoid)
// chains three MFMA GEMMs per 32-edge tile with an LDS transpose between them.
// Everything is f32 (bf16/fp16 MFMA would break 1e-5 parity through 12 residual layers).
#include "daco_device.h"
#include "../../include/deepaco_hip.h"

#include <cstdlib>

namespace daco {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int U = 32;                       // units

// ---- parameter block layout (floats), built by the host (deepaco_amd/net.py pack_params)
// [0]               v_lin0.W [32][feats] | v_lin0.b [32]
// then              e_lin0.W [32]        | e_lin0.b [32]
// then 12 x layer:  WvT [32 c][128 c']  (x1|x2|x3|x4 outputs, transposed) | bv [128]
//                   We [32 o][32 c] | be [32] | bn_v scale[32] shift[32] | bn_e scale[32] shift[32]
// then head:        W1 [32][32] b1 [32] W2 [32][32] b2 [32] W3 [32] b3 [1]
constexpr int LAYER_FLOATS = 32 * 128 + 128 + 32 * 32 + 32 + 4 * 32;
__host__ __device__ inline size_t off_layer(int feats, int l) { return (size_t)32 * feats + 32 + 64 + (size_t)l * LAYER_FLOATS; }
__host__ __device__ inline size_t off_head(int feats) { return off_layer(feats, 12); }
constexpr int HEAD_FLOATS = 2 * (32 * 32 + 32) + 32 + 1;
constexpr int GNN_SPLIT_MIN_EDGES = 200000;   // above this a layer is two launches (edge | node), below it one

// Activations.  e^-x in six full-rate instructions instead of libm's twelve (the two activations are evaluated 2*E*32 times
// per layer and were most of a layer's VALU work): t = -x*log2(e) as the rounded product plus its exact residual (two fmas,
// the second adds the low word of log2 e), 2^t on the hardware exponential (its range reduction is exact), first-order
// correction for the residual: ~1 ulp, like libm.  1/(1+e^-x) with the hardware reciprocal (1 ulp).  The two-wide
// versions are the same arithmetic on v_pk_* instructions.
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.92596299e-8f, LN2F = 0.693147182464599609375f;
__device__ inline float exp_neg(float x) {
  const float nx = fminf(-x, 87.0f);                      // beyond: e^-x > 1e37, sigmoid and silu are 0 to f32 either way
  const float t = nx * L2E_HI;
  const float lo = fmaf(nx, L2E_LO, fmaf(nx, L2E_HI, -t));
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, lo * LN2F, e);
}
__device__ inline float sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.0f + exp_neg(x)); }
__device__ inline float silu(float x) { return x * sigmoidf(x); }
__device__ inline f32x2 sigmoid2(f32x2 x) {
  f32x2 nx;
  nx.x = fminf(-x.x, 87.0f); nx.y = fminf(-x.y, 87.0f);
  const f32x2 hi = {L2E_HI, L2E_HI}, lw = {L2E_LO, L2E_LO}, ln2 = {LN2F, LN2F}, one = {1.0f, 1.0f};
  const f32x2 t = nx * hi;
  const f32x2 lo = __builtin_elementwise_fma(nx, lw, __builtin_elementwise_fma(nx, hi, -t));
  f32x2 e;
  e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
  const f32x2 d = one + __builtin_elementwise_fma(e, lo * ln2, e);
  f32x2 r;
  r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
  return r;
}
__device__ inline f32x2 silu2(f32x2 x) { return x * sigmoid2(x); }

// x = silu(v_lin0(x)); X1234(0) = layer-0 node linears.  8 nodes per 256-thread workgroup.
__global__ void __launch_bounds__(256)
gnn_node_init_kernel(int n, int feats, const float *xin, const float *params, float *x, float *X) {
  __shared__ float xs[8][U];
  const int il = threadIdx.x >> 5, o = threadIdx.x & 31;
  const int i = blockIdx.x * 8 + il;
  const float *W = params, *b = params + 32 * feats;
  float v = 0.0f;
  if (i < n) {
    v = b[o];
    for (int f = 0; f < feats; ++f) v = fmaf(xin[(size_t)i * feats + f], W[o * feats + f], v);
    v = silu(v);
    x[(size_t)i * U + o] = v;
  }
  xs[il][o] = v;
  __syncthreads();
  if (i >= n) return;
  const float *WT = params + off_layer(feats, 0), *bv = WT + 32 * 128;
  // the four linears side by side, eight input channels' weights in flight at a time (one output's fmas stay in channel order)
  float acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = bv[q * 32 + o];
#pragma unroll 8
  for (int c = 0; c < U; ++c) {
    const float xc = xs[il][c];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = fmaf(xc, WT[c * 128 + q * 32 + o], acc[q]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) X[(size_t)i * 128 + q * 32 + o] = acc[q];
}

// w = silu(e_lin0(edge_attr)); four channels per thread (16-byte stores)
__global__ void __launch_bounds__(256)
gnn_edge_init_kernel(int E, int feats, const float *attr, const float *params, float *w) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)E * 8) return;
  const int e = (int)(idx >> 3), c0 = (int)(idx & 7) * 4;
  const float *W = params + 32 * feats + 32, *b = W + 32;
  const float a = attr[e];
  float4 out;
  out.x = silu(fmaf(a, W[c0 + 0], b[c0 + 0]));
  out.y = silu(fmaf(a, W[c0 + 1], b[c0 + 1]));
  out.z = silu(fmaf(a, W[c0 + 2], b[c0 + 2]));
  out.w = silu(fmaf(a, W[c0 + 3], b[c0 + 3]));
  *reinterpret_cast<float4 *>(w + (size_t)e * U + c0) = out;
}

// one 32-edge x 32-channel tile: acc = Wm * a-rows, with the K split (0..15 | 16..31) over the two
// lane halves.  `rowptr_` points at this lane's edge row (32 floats).
__device__ inline f32x16 tile_gemm(const float *row, const float *Wm, int lane) {
  const int h = lane >> 5, o = lane & 31;
  float a[16], bw[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 t = *reinterpret_cast<const float4 *>(row + h * 16 + q * 4);
    a[q * 4 + 0] = t.x; a[q * 4 + 1] = t.y; a[q * 4 + 2] = t.z; a[q * 4 + 3] = t.w;
    const float4 u = *reinterpret_cast<const float4 *>(Wm + o * U + h * 16 + q * 4);
    bw[q * 4 + 0] = u.x; bw[q * 4 + 1] = u.y; bw[q * 4 + 2] = u.z; bw[q * 4 + 3] = u.w;
  }
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], bw[kk], acc, 0, 0, 0);
  return acc;
}
// MFMA 32x32 output layout: register r of lane l holds D[row][col], col = l & 31,
__device__ inline int drow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---------------- edge update: 4 waves, one 32-edge tile each.
// The MFMA result (lane = channel, 16 edge rows per lane) is turned through a wave-private LDS tile
// so that the epilogue runs edge-major with 16-byte accesses: the residual, the two node terms
// and the store are four accesses per array and lane instead of sixteen 4-byte ones.
__device__ inline void edge_update(int E, int wg, const int *src, const int *dst, const float *We, const float *be,
                                   const float *se, const float *te, const float *X, const float *w0, float *w1out,
                                   float (*tile)[36]) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int e0 = (wg * 4 + wave) * 32;
  if (e0 >= E) return;
  const int o = lane & 31;
  // eight lanes per edge (4 channels each), eight edges per pass: every 16-byte access of a pass is
  // one contiguous 1 KiB run for the edge arrays and whole 128-byte node rows for the gathers
  const int c0 = (lane & 7) * 4;
  float4 res[4];                                         // this lane's part of the old edge state (tile + residual)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int el = q * 8 + (lane >> 3), e = min(e0 + el, E - 1);
    res[q] = *reinterpret_cast<const float4 *>(w0 + (size_t)e * U + c0);
    *reinterpret_cast<float4 *>(&tile[el][c0]) = res[q];
  }
  __builtin_amdgcn_wave_barrier();
  const f32x16 acc = tile_gemm(&tile[o][0], We, lane);   // A rows out of LDS: the global read above was coalesced
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 16; ++r) tile[drow(r, lane)][o] = acc[r];
  __builtin_amdgcn_wave_barrier();
  const float4 bb = *reinterpret_cast<const float4 *>(be + c0);
  const float4 sc = *reinterpret_cast<const float4 *>(se + c0), sh = *reinterpret_cast<const float4 *>(te + c0);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int el = q * 8 + (lane >> 3), e = e0 + el;
    if (e < E) {
      const int s = src[e], d = dst[e];
      const float4 g = *reinterpret_cast<const float4 *>(&tile[el][c0]);
      const float4 a3 = *reinterpret_cast<const float4 *>(X + (size_t)s * 128 + 64 + c0);
      const float4 a4 = *reinterpret_cast<const float4 *>(X + (size_t)d * 128 + 96 + c0);
      float4 out;
      out.x = res[q].x + silu(fmaf(g.x + bb.x + a3.x + a4.x, sc.x, sh.x));
      out.y = res[q].y + silu(fmaf(g.y + bb.y + a3.y + a4.y, sc.y, sh.y));
      out.z = res[q].z + silu(fmaf(g.z + bb.z + a3.z + a4.z, sc.z, sh.z));
      out.w = res[q].w + silu(fmaf(g.w + bb.w + a3.w + a4.w, sc.w, sh.w));
      *reinterpret_cast<float4 *>(w1out + (size_t)e * U + c0) = out;
    }
  }
}

// ---------------- node update (+ next layer's node linears): 8 nodes per workgroup, 32 lanes per node.
// The neighbour ids of a node are fetched 32 at a time by its lanes (one coalesced load), so the
// feature gathers X[dst] depend on a lane exchange, not on a second trip to memory, and the
// compiler can keep several of them in flight.
__device__ inline void node_update(int n, int wg, int feats, int layer, const int *dst, const int *rowptr, const int *perm,
                                   const float *params, const float *sv, const float *tv, const float *x0, const float *X,
                                   const float *w0, float *x1out, float *Xnext, float (*xs)[U]) {
  const int il = threadIdx.x >> 5, o = threadIdx.x & 31;
  const int i = wg * 8 + il;
  float xn = 0.0f;
  if (i < n) {
    const int lo = rowptr[i], hi = rowptr[i + 1];
    // eight lanes per incident edge (4 channels each), four edges per pass: 16-byte loads of whole
    // 128-byte rows.  Edge j of the list goes to lane group j % 4; the four partial sums are added
    // at the end (a fixed order, independent of the graph's size).
    const int grp = o >> 3, c0 = (o & 7) * 4;
    float4 part = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int q0 = lo; q0 < hi; q0 += 32) {
      const int m = min(32, hi - q0);
      int ev = 0, dv = 0;
      if (o < m) {
        ev = perm ? perm[q0 + o] : q0 + o;
        dv = dst[ev];
      }
#pragma unroll 2
      for (int j = 0; j < m; j += 4) {
        const int jj = j + grp;
        const int e = __shfl(ev, jj, 32), d = __shfl(dv, jj, 32);
        if (jj < m) {
          const float4 wv = *reinterpret_cast<const float4 *>(w0 + (size_t)e * U + c0);
          const float4 xv = *reinterpret_cast<const float4 *>(X + (size_t)d * 128 + 32 + c0);
          part.x = fmaf(sigmoidf(wv.x), xv.x, part.x);
          part.y = fmaf(sigmoidf(wv.y), xv.y, part.y);
          part.z = fmaf(sigmoidf(wv.z), xv.z, part.z);
          part.w = fmaf(sigmoidf(wv.w), xv.w, part.w);
        }
      }
    }
    part.x += __shfl_xor(part.x, 8, 32);  part.y += __shfl_xor(part.y, 8, 32);
    part.z += __shfl_xor(part.z, 8, 32);  part.w += __shfl_xor(part.w, 8, 32);
    part.x += __shfl_xor(part.x, 16, 32); part.y += __shfl_xor(part.y, 16, 32);
    part.z += __shfl_xor(part.z, 16, 32); part.w += __shfl_xor(part.w, 16, 32);
    // back to one channel per lane: channel o is component o%4 of the lanes holding channels (o/4)*4..
    const int from = o >> 2;
    const float p0 = __shfl(part.x, from, 32), p1 = __shfl(part.y, from, 32), p2 = __shfl(part.z, from, 32),
                p3 = __shfl(part.w, from, 32);
    float agg = (o & 2) ? ((o & 1) ? p3 : p2) : ((o & 1) ? p1 : p0);
    agg = agg / (float)max(hi - lo, 1);
    const float y = fmaf(X[(size_t)i * 128 + o] + agg, sv[o], tv[o]);
    xn = x0[(size_t)i * U + o] + silu(y);
    x1out[(size_t)i * U + o] = xn;
  }
  if (layer == 11) return;
  xs[il][o] = xn;
  __syncthreads();
  if (i >= n) return;
  const float *WT = params + off_layer(feats, layer + 1), *bv = WT + 32 * 128;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float acc = bv[q * 32 + o];
    for (int c = 0; c < U; ++c) acc = fmaf(xs[il][c], WT[c * 128 + q * 32 + o], acc);
    Xnext[(size_t)i * 128 + q * 32 + o] = acc;
  }
}

// one launch per layer (small graphs are launch-bound): edge workgroups first, then node workgroups
__global__ void __launch_bounds__(256)
gnn_layer_kernel(int n, int E, int feats, int layer, int edge_blocks, const int *src, const int *dst, const int *rowptr,
                 const int *perm, const float *params, const float *x0, const float *X, const float *w0,
                 float *x1out, float *Xnext, float *w1out) {
  const float *lp = params + off_layer(feats, layer);
  const float *We = lp + 32 * 128 + 128, *be = We + 32 * 32;
  const float *sv = be + 32, *tv = sv + 32, *se = tv + 32, *te = se + 32;
  __shared__ float xs[8][U];
  __shared__ __attribute__((aligned(16))) float tile[4][32][36];
  if ((int)blockIdx.x < edge_blocks) { edge_update(E, blockIdx.x, src, dst, We, be, se, te, X, w0, w1out, tile[threadIdx.x >> 6]); return; }
  node_update(n, (int)blockIdx.x - edge_blocks, feats, layer, dst, rowptr, perm, params, sv, tv, x0, X, w0, x1out, Xnext, xs);
}

// large (batched) graphs: the two halves as separate launches, so the node half -- latency-bound
// gathers -- runs at its own, much smaller register budget and full occupancy
__global__ void __launch_bounds__(256)
gnn_edge_kernel(int E, int feats, int layer, const int *src, const int *dst, const float *params, const float *X,
                const float *w0, float *w1out) {
  const float *lp = params + off_layer(feats, layer);
  const float *We = lp + 32 * 128 + 128, *be = We + 32 * 32, *se = be + 32 + 64, *te = se + 32;
  __shared__ __attribute__((aligned(16))) float tile[4][32][36];
  edge_update(E, blockIdx.x, src, dst, We, be, se, te, X, w0, w1out, tile[threadIdx.x >> 6]);
}
__global__ void __launch_bounds__(256)
gnn_node_kernel(int n, int feats, int layer, const int *dst, const int *rowptr, const int *perm, const float *params,
                const float *x0, const float *X, const float *w0, float *x1out, float *Xnext) {
  const float *lp = params + off_layer(feats, layer);
  const float *sv = lp + 32 * 128 + 128 + 32 * 32 + 32, *tv = sv + 32;
  __shared__ float xs[8][U];
  node_update(n, blockIdx.x, feats, layer, dst, rowptr, perm, params, sv, tv, x0, X, w0, x1out, Xnext, xs);
}

// ---------------- fused layer for src-sorted edge lists (perm == nullptr; the reference's k-NN graphs are built that way,
// tsp/utils.py:16-27): a wave owns NPW consecutive nodes AND their out-edges, so the old edge state is read once -- it feeds
// the edge update (w') and, through gate = sigmoid(w0), the node aggregation -- instead of once per half.  Per 32-edge tile:
// the MFMA tile GEMM and the edge-major epilogue of edge_update, plus gate * x2[dst] written channel-major into a second
// wave-private LDS tile; lane = channel then adds the tile's edges in edge order, cutting at the nodes' CSR boundaries
// (wave-uniform scalars).  The wave finishes its nodes itself: x' and the next layer's four node linears.
// Sum order of the aggregation: the node's edges in list order (a fixed order; the split kernels add four interleaved
// partial sums -- both are within the fixtures' tolerance of the reference's scatter-mean).
constexpr int FUSED_MAX_NPW = 16;
__global__ void __launch_bounds__(256, 3)
gnn_fused_layer_kernel(int n, int E, int feats, int layer, int npw, const int *src, const int *dst, const int *rowptr,
                       const float *params, const float *x0, const float *X, const float *w0, float *x1out, float *Xnext,
                       float *w1out) {
  __shared__ __attribute__((aligned(16))) float tile_s[4][32][36];
  __shared__ __attribute__((aligned(16))) float prod_s[4][32][36];            // [channel][edge of the tile]
  __shared__ float agg_s[4][FUSED_MAX_NPW][U];
  __shared__ __attribute__((aligned(16))) float we_s[32][36];                 // We, re-read per tile (16 registers less)
  const float *lp = params + off_layer(feats, layer);
  const float *We = lp + 32 * 128 + 128, *be = We + 32 * 32;
  const float *sv = be + 32, *tv = sv + 32, *se = tv + 32, *te = se + 32;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  *reinterpret_cast<float4 *>(&we_s[threadIdx.x >> 3][(threadIdx.x & 7) * 4]) = *reinterpret_cast<const float4 *>(We + threadIdx.x * 4);
  __syncthreads();                                                            // the only workgroup barrier
  // Workgroups are dealt round-robin to the 8 XCDs, each with its own 4 MiB L2.  All workgroups are resident at once, so
  // with the natural numbering every XCD would gather node rows of every graph (64 x TSP-500: 16 MB).  Renumbered, XCD x
  // owns one contiguous eighth of the nodes -- its gathers stay inside a few graphs' rows (2 MB).
  const int per_xcd = (int)gridDim.x >> 3;                                    // the grid is a multiple of 8
  const int block = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  const int i0 = (block * 4 + wave) * npw;
  if (i0 >= n) return;
  const int cnt = min(npw, n - i0);
  // the wave's CSR boundaries live in lanes 0..cnt of one register (read back with v_readlane: the walk below must not
  // touch memory, a vector load there would drain the prefetch queue)
  const int rp = rowptr[i0 + min(lane, cnt)];
  auto row_at = [&](int j) { return __builtin_amdgcn_readlane(rp, j); };
  const int ebeg = row_at(0), eend = row_at(cnt);
  float (*tile)[36] = tile_s[wave], (*prod)[36] = prod_s[wave];
  const int o = lane & 31, h = lane >> 5, c0 = (lane & 7) * 4;
#pragma unroll
  for (int j = 0; j < FUSED_MAX_NPW; ++j) agg_s[wave][j][o] = 0.0f;           // nodes without out-edges aggregate 0
  const float bias_h = be[o];                       // the accumulator starts from the bias (lane = output channel)
  const float4 sc = *reinterpret_cast<const float4 *>(se + c0), sh = *reinterpret_cast<const float4 *>(te + c0);
  // walk state of the aggregation: node i0 + cur collects edges up to seg_end
  int cur = 0, seg_end = row_at(1);
  while (cur < cnt && seg_end <= ebeg) { ++cur; seg_end = cur < cnt ? row_at(cur + 1) : 0x7fffffff; }
  float run = 0.0f;
  // Every load of the loop is unconditional (edge ids clamped to the wave's last edge; the stores and the products are
  // masked instead), so one iteration's twelve gathers and the next tile's rows are all in flight together: eight lanes per
  // edge (4 channels each), eight edges per pass, four passes per tile.
  const int g8 = lane >> 3, elast = eend - 1;
  float4 res[4];
  int sn[4], dn[4];
  if (ebeg < eend) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = min(ebeg + q * 8 + g8, elast);
      res[q] = *reinterpret_cast<const float4 *>(w0 + (size_t)e * U + c0);
      sn[q] = src[e]; dn[q] = dst[e];
    }
  }
  for (int e0 = ebeg; e0 < eend; e0 += 32) {
    float4 a3[4], a4[4], x2[4], old[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      old[q] = res[q];
      *reinterpret_cast<float4 *>(&tile[q * 8 + g8][c0]) = old[q];
    }
    __builtin_amdgcn_wave_barrier();
    // (values the compiler would otherwise keep across iterations -- a 16-register accumulator seed, the We operands --
    // and then spill are made to look loop-variant)
    f32x16 acc;
    float seed = bias_h;
    int opaque = 0;
    asm volatile("" : "+v"(seed), "+s"(opaque));
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = seed;
    {
      float a[16], bw[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4 *>(&tile[o][h * 16 + q * 4]);
        a[q * 4 + 0] = t.x; a[q * 4 + 1] = t.y; a[q * 4 + 2] = t.z; a[q * 4 + 3] = t.w;
        const float4 u = *reinterpret_cast<const float4 *>(&we_s[o][h * 16 + q * 4 + opaque * 4]);
        bw[q * 4 + 0] = u.x; bw[q * 4 + 1] = u.y; bw[q * 4 + 2] = u.z; bw[q * 4 + 3] = u.w;
      }
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], bw[kk], acc, 0, 0, 0);
    }
    // The gates sigmoid(w0) need nothing but the old rows: they are computed here, between the sixteen dependent MFMAs
    // (64 cycles each, during which the wave would otherwise issue nothing) -- half of the tile's activation work.
    f32x2 gate[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x2 w01 = {old[q].x, old[q].y}, w23 = {old[q].z, old[q].w};
      gate[q][0] = sigmoid2(w01);
      gate[q][1] = sigmoid2(w23);
      asm volatile("" : "+v"(gate[q][0]), "+v"(gate[q][1]));                   // computed here, not sunk to their use
    }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // one MFMA ...
      __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);                     // ... then ten VALU instructions of the gates
    }
    __builtin_amdgcn_sched_barrier(0);                                        // (nothing else moves into the chain)
    // behind the matrix work: this tile's twelve gathers, then the next tile's rows and node ids
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a3[q] = *reinterpret_cast<const float4 *>(X + (size_t)sn[q] * 128 + 64 + c0);
      a4[q] = *reinterpret_cast<const float4 *>(X + (size_t)dn[q] * 128 + 96 + c0);
      x2[q] = *reinterpret_cast<const float4 *>(X + (size_t)dn[q] * 128 + 32 + c0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = min(e0 + 32 + q * 8 + g8, elast);
      res[q] = *reinterpret_cast<const float4 *>(w0 + (size_t)e * U + c0);
      sn[q] = src[e]; dn[q] = dst[e];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[drow(r, lane)][o] = acc[r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int el = q * 8 + g8, e = e0 + el;
      const bool live = e < eend;
      const float4 g = *reinterpret_cast<const float4 *>(&tile[el][c0]);
      const f32x2 y01 = {fmaf(g.x + a3[q].x + a4[q].x, sc.x, sh.x), fmaf(g.y + a3[q].y + a4[q].y, sc.y, sh.y)};
      const f32x2 y23 = {fmaf(g.z + a3[q].z + a4[q].z, sc.z, sh.z), fmaf(g.w + a3[q].w + a4[q].w, sc.w, sh.w)};
      const f32x2 s01 = silu2(y01), s23 = silu2(y23);
      const f32x2 g01 = gate[q][0], g23 = gate[q][1];
      if (live) {
        float4 out;
        out.x = old[q].x + s01.x; out.y = old[q].y + s01.y; out.z = old[q].z + s23.x; out.w = old[q].w + s23.y;
        *reinterpret_cast<float4 *>(w1out + (size_t)e * U + c0) = out;
      }
      prod[c0 + 0][el] = live ? g01.x * x2[q].x : 0.0f;
      prod[c0 + 1][el] = live ? g01.y * x2[q].y : 0.0f;
      prod[c0 + 2][el] = live ? g23.x * x2[q].z : 0.0f;
      prod[c0 + 3][el] = live ? g23.y * x2[q].w : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();
    // lane = channel: the tile's 32 products in edge order, cut at the CSR boundaries
#pragma unroll 1
    for (int j = 0; j < 8; ++j) {
      const float4 v4 = *reinterpret_cast<const float4 *>(&prod[o][j * 4]);
      const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        run += v[k];
        const int enext = e0 + j * 4 + k + 1;
        if (enext == seg_end) {
          agg_s[wave][cur][o] = run;
          run = 0.0f;
          do { ++cur; seg_end = cur < cnt ? row_at(cur + 1) : 0x7fffffff; } while (cur < cnt && seg_end <= enext);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // the wave's nodes, four at a time: x' = x0 + silu(bn_v(x1 + mean)), then the next layer's node linears (lane -> outputs
  // lane and lane + 64; the weights are read once per group of four nodes)
  const float *WT = params + off_layer(feats, layer + 1), *bv = WT + 32 * 128;
  for (int j0 = 0; j0 < cnt; j0 += 4) {
    float xn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xn[j] = 0.0f;
      if (j0 + j < cnt) {
        const int i = i0 + j0 + j;
        const int deg = row_at(j0 + j + 1) - row_at(j0 + j);
        const float agg = agg_s[wave][j0 + j][o] / (float)max(deg, 1);
        const float y = fmaf(X[(size_t)i * 128 + o] + agg, sv[o], tv[o]);
        xn[j] = x0[(size_t)i * U + o] + silu(y);
        if (h == 0) x1out[(size_t)i * U + o] = xn[j];
      }
    }
    if (layer == 11) continue;
    float y0[4], y1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { y0[j] = bv[lane]; y1[j] = bv[64 + lane]; }
    for (int c = 0; c < U; ++c) {
      const float wa = WT[c * 128 + lane], wb = WT[c * 128 + 64 + lane];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xc = __shfl(xn[j], c, 64);
        y0[j] = fmaf(xc, wa, y0[j]);
        y1[j] = fmaf(xc, wb, y1[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j0 + j < cnt) {
        Xnext[(size_t)(i0 + j0 + j) * 128 + lane] = y0[j];
        Xnext[(size_t)(i0 + j0 + j) * 128 + 64 + lane] = y1[j];
      }
  }
}

// ---------------- fused layer, second version (round 4).  Same ownership and arithmetic structure as gnn_fused_layer_kernel;
// what changed is what the counters of round 3 pointed at (154-168 registers -> 3 waves per SIMD, VALU 0.47 of the time, a
// third of it 64-bit address arithmetic and quarter-rate transcendentals):
//   * four waves per SIMD: <= 128 registers and <= 40 KB of LDS per workgroup.  The products gate * x2[dst] go back into the
//     MFMA tile (all four result rows of a lane are read first), the gate is not kept across the MFMA chain, the MFMA
//     operands are fetched from LDS in two K-halves;
//   * the src-side term x3[src] is not gathered: the edge list is src-sorted, a wave's edges start at its own <= 16 nodes,
//     whose x3 rows sit in 2 KB of LDS;
//   * the dst-side gathers (x2, x4) are issued BEFORE the MFMA chain of their tile (the ids arrive with the previous
//     tile's prefetch), so the chain's 1000 cycles cover them;
//   * one reciprocal serves both activations of an element: r = 1 / ((1 + e^-w0)(1 + e^-y)), sigmoid(w0) = r (1 + e^-y),
//     sigmoid(y) = r (1 + e^-w0) -- 3 instead of 4 quarter-rate instructions per element (arguments clamped to e^44 so the
//     product stays finite; sigmoid(-44) = 8e-20 is zero at the network's 1e-5 tolerance either way);
//   * the accumulator starts from zero (inline constant) and the bias is folded into the BatchNorm shift; every address is
//     a uniform base plus a 32-bit byte offset (host guarantees E * 128 < 4 GB).
// Round 5, after an ablation of this kernel (profiles/r05_gnn_layer_ablation.txt: memory alone and compute alone ~85 us each,
// together 114; no single part is the bound) -- 120 -> 113 us per launch in alternating profiler runs:
//   * the aggregate is S x P on the matrix unit (S[node][edge] = 1 where the edge leaves the node), accumulated over the
//     wave's tiles in two 16x16x4 blocks: the products are read back once, nothing is cut by scalar compares and branches;
//   * the node phase has all of a wavefront's loads in flight at once (it waited once per node and per input channel, and
//     every wavefront of the one-round launch is in that phase at the same time: 15 us of the launch);
//   * e^-x without the residual term of exp_neg (see one_plus_exp_neg2).
// NOTE: a register spilled INSIDE the tile loop is reloaded behind `s_waitcnt vmcnt(0)`, i.e. behind the next tile's row
// prefetch (loads return in order) -- that serialises the loop (measured: +5 us with two such reloads).  Check the .s.
constexpr int F2_MAX_NPW = 16;
__device__ inline f32x2 one_plus_exp_neg2(f32x2 x) {       // 1 + e^-x, x clamped at -44
  // -min(-x, 44) = max(x, -44): one v_max with a literal each (fminf would add a second instruction that quiets NaNs)
  f32x2 m;
  asm("v_max_f32 %0, 0xc2300000, %1" : "=v"(m.x) : "v"(x.x));
  asm("v_max_f32 %0, 0xc2300000, %1" : "=v"(m.y) : "v"(x.y));
  // 2^(-x log2 e) on the rounded product alone (no residual term, unlike exp_neg above): the argument's rounding moves
  // sigmoid(x) by at most s (1 - s) |x| 2^-24 <= 1.4e-8 and silu(x) by at most 2.7e-8 -- below half an ulp of either wherever
  // they are not themselves below 1e-7 -- and takes four of ten instructions off each of the 2 * 32 * E evaluations of a layer
  const f32x2 nl2e = {-L2E_HI, -L2E_HI}, one = {1.0f, 1.0f};
  const f32x2 t = m * nl2e;
  f32x2 e;
  e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
  return one + e;
}
__device__ inline f32x2 silu2_fast(f32x2 x) {             // x sigmoid(x) on the same exponential (silu(-44) = -3e-18 either way)
  const f32x2 d = one_plus_exp_neg2(x);
  f32x2 r;
  r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
  return x * r;
}
// INIT (layer 0 only): the old edge state is not read but made on the fly, w0 = silu(e_lin0(edge_attr)) (tsp/net.py:31) --
// 4 bytes per edge instead of 128, and no separate launch that writes E * 128 bytes first.
// HEAD (round 6): the LAST layer with the output head in it (tsp/net.py:59-66 on top of :27-45).  The head only takes the edge
// state, and layer 12's node state feeds nothing, so this variant runs the edge update of a tile, keeps the new rows in the
// wavefront's LDS tile and sends them through the head's two 32x32 linears and the 32 -> 1 row sum right there (the tile code
// of gnn_head_kernel, the same MFMA order per row: the same bits): it writes heu[E] -- 4 bytes per edge -- and neither the edge
// state (128 bytes per edge written here and read again by the head launch) nor the node state; no x2 gathers, no aggregate, no
// node phase.  Used when the caller does not ask for the embedding.
#ifndef DACO_GNN_F2_OCC
#define DACO_GNN_F2_OCC 4                 // workgroups per CU the kernel is compiled for (39 KB of LDS each: four fit)
#endif
template <bool INIT, bool HEAD = false>
__global__ void __launch_bounds__(256, DACO_GNN_F2_OCC)
gnn_fused2_layer_kernel(int n, int E, int feats, int layer, int npw, const int *src, const int *dst, const int *rowptr,
                        const float *params, const float *x0, const float *X, const float *w0, float *x1out, float *Xnext,
                        float *w1out, const float *attr, int nt, float *heu = nullptr) {
  __shared__ __attribute__((aligned(16))) float tile_s[4][32][36];            // A rows -> MFMA result (edge-major) -> products (channel-major)
  __shared__ __attribute__((aligned(16))) float agg_s[4][F2_MAX_NPW][U];      // per node: the aggregate, then x' (node phase)
  __shared__ __attribute__((aligned(16))) float x3_s[4][F2_MAX_NPW][U];       // x3 rows of the wave's own nodes
  __shared__ __attribute__((aligned(16))) float we_s[32][36];                 // We
  const float *lp = params + off_layer(feats, layer);
  const float *We = lp + 32 * 128 + 128, *be = We + 32 * 32;
  const float *sv = be + 32, *tv = sv + 32, *se = tv + 32, *te = se + 32;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  *reinterpret_cast<float4 *>(&we_s[threadIdx.x >> 3][(threadIdx.x & 7) * 4]) = *reinterpret_cast<const float4 *>(We + threadIdx.x * 4);
  // INIT: e_lin0's weight and bias (a row is silu(attr * W + b)) are read from LDS when a tile starts: kept in registers they
  // were eight more than the loop has (reloads inside the loop wait behind the row prefetch, see the note above)
  __shared__ __attribute__((aligned(16))) float init_s[INIT ? 2 : 1][INIT ? 32 : 4];
  if (INIT && threadIdx.x < 64) init_s[threadIdx.x >> 5][threadIdx.x & 31] = params[32 * feats + 32 + threadIdx.x];
  __syncthreads();                                                            // the only workgroup barrier
  const int per_xcd = (int)gridDim.x >> 3;                                    // XCD x owns one contiguous eighth of the nodes
  const int block = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  const int i0 = (block * 4 + wave) * npw;
  if (i0 >= n) return;
  const int cnt = min(npw, n - i0);
  const int rp = rowptr[i0 + min(lane, cnt)];
  auto row_at = [&](int j) { return __builtin_amdgcn_readlane(rp, j); };
  const int ebeg = row_at(0), eend = row_at(cnt);
  float (*tile)[36] = tile_s[wave];
  const int o = lane & 31, h = lane >> 5, c0 = (lane & 7) * 4, g8 = lane >> 3;
  const char *w0b = reinterpret_cast<const char *>(w0), *Xb = reinterpret_cast<const char *>(X);
  char *w1b = reinterpret_cast<char *>(w1out);
  {
    // the wave's x3 rows: 16 rows x 128 B, two 16-byte pieces per lane (rows past cnt: clamped, never used)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int j = k * 8 + g8, jn = min(i0 + min(j, cnt - 1), n - 1);
      *reinterpret_cast<float4 *>(&x3_s[wave][j][c0]) = *reinterpret_cast<const float4 *>(Xb + (uint32_t)jn * 512u + 256u + (uint32_t)c0 * 4u);
    }
  }
  const float4 sc = *reinterpret_cast<const float4 *>(se + c0);
  float4 sh = *reinterpret_cast<const float4 *>(te + c0);
  {
    const float4 bb = *reinterpret_cast<const float4 *>(be + c0);             // (g + b + a3 + a4) s + t = (g + a3 + a4) s + (b s + t)
    sh.x = fmaf(bb.x, sc.x, sh.x); sh.y = fmaf(bb.y, sc.y, sh.y); sh.z = fmaf(bb.z, sc.z, sh.z); sh.w = fmaf(bb.w, sc.w, sh.w);
  }
  // the aggregate: agg[node][channel] = sum over the node's edges of the products, as S x P on the matrix unit -- S[node][edge]
  // = 1 where the edge leaves the node -- accumulated over all tiles in two 16 x 16 blocks (channels 0-15, 16-31; four
  // registers each), so a tile's 32 x 32 products are read back once (four 16-byte reads per lane) and nothing is carried
  // through scalar compares.  Lane (r = lane / 16, i = lane % 16): node i's edge range [elo, elo + edeg), K-slot r.
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v aggA = {0, 0, 0, 0}, aggB = {0, 0, 0, 0};
  const int r16 = lane >> 4, i16 = lane & 15;
  const int elo = rowptr[i0 + min(i16, cnt)];
  const uint32_t edeg = (uint32_t)(rowptr[i0 + min(i16 + 1, cnt)] - elo);
  const int elast = eend - 1;
  float4 res[4];
  auto load_row = [&](int e) -> float4 {
    if constexpr (INIT) {                                                              // (expanded when the tile starts)
      const float av = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(attr) + (uint32_t)e * 4u);
      return make_float4(av, av, av, av);
    }
    else {
      // the edge rows are a stream (each read once per layer, by one lane): with the non-temporal policy (DACO_GNN_NT=0 turns
      // it off) they do not push the node rows -- which every edge of a node gathers again -- out of the L2
      const float4 *ptr = reinterpret_cast<const float4 *>(w0b + (uint32_t)e * 128u + (uint32_t)c0 * 4u);
      if (nt) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(ptr));
        return make_float4(v.x, v.y, v.z, v.w);
      }
      return *ptr;
    }
  };
  auto made_row = [&](float4 r) -> float4 {
    if constexpr (INIT) {
      const float4 iw = *reinterpret_cast<const float4 *>(&init_s[0][c0]), ib = *reinterpret_cast<const float4 *>(&init_s[INIT ? 1 : 0][c0]);
      const f32x2 p01 = {fmaf(r.x, iw.x, ib.x), fmaf(r.y, iw.y, ib.y)}, p23 = {fmaf(r.z, iw.z, ib.z), fmaf(r.w, iw.w, ib.w)};
      const f32x2 s01 = silu2_fast(p01), s23 = silu2_fast(p23);
      return make_float4(s01.x, s01.y, s23.x, s23.y);
    } else return r;
  };
  // node ids of the next tile's edges, as loaded: they are looked at only when that tile starts (touching them right after
  // the load would make the wave wait for the row prefetch issued just before -- loads return in order)
  int sn[4], dn[4];
  if (ebeg < eend) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = min(ebeg + q * 8 + g8, elast);
      res[q] = load_row(e);
      sn[q] = src[e]; dn[q] = dst[e];
    }
  }
  __builtin_amdgcn_wave_barrier();
  for (int e0 = ebeg; e0 < eend; e0 += 32) {
    float4 a4[4], x2[4], old[4];
    uint32_t sl = 0;                                                           // the edges' own nodes (src - i0), four bits per pass
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      old[q] = made_row(res[q]);
      *reinterpret_cast<float4 *>(&tile[q * 8 + g8][c0]) = old[q];
      const uint32_t drow_off = (uint32_t)dn[q] * 512u + (uint32_t)c0 * 4u;
      a4[q] = *reinterpret_cast<const float4 *>(Xb + drow_off + 384u);
      if constexpr (!HEAD) x2[q] = *reinterpret_cast<const float4 *>(Xb + drow_off + 128u);
      sl |= (uint32_t)min(max(sn[q] - i0, 0), F2_MAX_NPW - 1) << (4 * q);
    }
    __builtin_amdgcn_wave_barrier();
    f32x16 acc;
    {
      int opaque = 0;
      asm volatile("" : "+s"(opaque));                                          // (keeps the We operands from being hoisted out of the loop and spilled)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float a[8], bw[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float4 t = *reinterpret_cast<const float4 *>(&tile[o][h * 16 + half * 8 + q * 4]);
          a[q * 4 + 0] = t.x; a[q * 4 + 1] = t.y; a[q * 4 + 2] = t.z; a[q * 4 + 3] = t.w;
          const float4 u = *reinterpret_cast<const float4 *>(&we_s[o][h * 16 + half * 8 + q * 4 + opaque * 4]);
          bw[q * 4 + 0] = u.x; bw[q * 4 + 1] = u.y; bw[q * 4 + 2] = u.z; bw[q * 4 + 3] = u.w;
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          if (half == 0 && kk == 0) {
            const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], bw[kk], zero, 0, 0, 0);
          } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], bw[kk], acc, 0, 0, 0);
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[drow(r, lane)][o] = acc[r];
    __builtin_amdgcn_wave_barrier();
    // the next tile's rows and node ids (its gathers are issued at its start)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = min(e0 + 32 + q * 8 + g8, elast);
      res[q] = load_row(e);
      sn[q] = src[e]; dn[q] = dst[e];
    }
    // epilogue, eight edges per pass.  Pass q reads result rows 8q .. 8q+7 and then writes its 8 x 32 products into the same
    // rows (channel c, edge j of the pass at tile[8q + c/4][(c%4)*8 + j]): the wave's LDS operations execute in order, a
    // pass's reads are done before its writes.
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int el = q * 8 + g8, e = e0 + el;
      const bool live = e < eend;
      const float4 gq = *reinterpret_cast<const float4 *>(&tile[el][c0]);
      const float4 a3 = *reinterpret_cast<const float4 *>(&x3_s[wave][(sl >> (4 * q)) & 15u][c0]);
      const f32x2 y01 = {fmaf(gq.x + a3.x + a4[q].x, sc.x, sh.x), fmaf(gq.y + a3.y + a4[q].y, sc.y, sh.y)};
      const f32x2 y23 = {fmaf(gq.z + a3.z + a4[q].z, sc.z, sh.z), fmaf(gq.w + a3.w + a4[q].w, sc.w, sh.w)};
      const f32x2 w01 = {old[q].x, old[q].y}, w23 = {old[q].z, old[q].w};
      const f32x2 gd01 = one_plus_exp_neg2(w01), gd23 = one_plus_exp_neg2(w23);     // 1 + e^-w0
      const f32x2 sd01 = one_plus_exp_neg2(y01), sd23 = one_plus_exp_neg2(y23);     // 1 + e^-y
      const f32x2 p01 = gd01 * sd01, p23 = gd23 * sd23;
      f32x2 r01, r23;
      r01.x = __builtin_amdgcn_rcpf(p01.x); r01.y = __builtin_amdgcn_rcpf(p01.y);
      r23.x = __builtin_amdgcn_rcpf(p23.x); r23.y = __builtin_amdgcn_rcpf(p23.y);
      const f32x2 gate01 = r01 * sd01, gate23 = r23 * sd23;                           // sigmoid(w0)
      const f32x2 s01 = y01 * (r01 * gd01), s23 = y23 * (r23 * gd23);               // silu(y)
      if constexpr (HEAD) {
        // the new row stays in the tile (row = edge, as the head's first linear reads it); rows past the wave's edges: zeros
        float4 out;
        out.x = old[q].x + s01.x; out.y = old[q].y + s01.y; out.z = old[q].z + s23.x; out.w = old[q].w + s23.y;
        asm volatile("" ::: "memory");                                       // (this pass's row reads stay above the write)
        *reinterpret_cast<float4 *>(&tile[el][c0]) = live ? out : make_float4(0.f, 0.f, 0.f, 0.f);
        continue;
      }
      if (live) {
        float4 out;
        out.x = old[q].x + s01.x; out.y = old[q].y + s01.y; out.z = old[q].z + s23.x; out.w = old[q].w + s23.y;
        if (nt) {
          typedef float f4v __attribute__((ext_vector_type(4)));
          const f4v ov = {out.x, out.y, out.z, out.w};
          __builtin_nontemporal_store(ov, reinterpret_cast<f4v *>(w1b + (uint32_t)e * 128u + (uint32_t)c0 * 4u));
        } else {
          *reinterpret_cast<float4 *>(w1b + (uint32_t)e * 128u + (uint32_t)c0 * 4u) = out;
        }
      }
      asm volatile("" ::: "memory");                                         // (the row reads above stay above these writes)
      float *pr = &tile[q * 8 + (c0 >> 2)][g8];
      pr[0] = live ? gate01.x * x2[q].x : 0.0f;
      pr[8] = live ? gate01.y * x2[q].y : 0.0f;
      pr[16] = live ? gate23.x * x2[q].z : 0.0f;
      pr[24] = live ? gate23.y * x2[q].w : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();
    if constexpr (HEAD) {
      // the head on this tile: heu = sigmoid(W3 silu(W2 silu(W1 w + b1) + b2) + b3)
      const float *hp = params + off_head(feats);
      const float *W1 = hp, *b1 = W1 + 1024, *W2 = b1 + 32, *b2 = W2 + 1024, *W3 = b2 + 32, *b3 = W3 + 32;
      __builtin_amdgcn_s_waitcnt(0xc07f);                                      // lgkmcnt(0): the rows are in the tile
      f32x16 hacc = tile_gemm(&tile[o][0], W1, lane);
      __builtin_amdgcn_wave_barrier();
      {
        const float bo = b1[o];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 v = {hacc[r] + bo, hacc[r + 1] + bo};
          const f32x2 a = silu2_fast(v);
          tile[drow(r, lane)][o] = a.x; tile[drow(r + 1, lane)][o] = a.y;
        }
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);
      hacc = tile_gemm(&tile[o][0], W2, lane);
      __builtin_amdgcn_wave_barrier();
      {
        const float bo = b2[o], wo = W3[o];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 v = {hacc[r] + bo, hacc[r + 1] + bo};
          const f32x2 a = silu2_fast(v);
          tile[drow(r, lane)][o] = a.x * wo; tile[drow(r + 1, lane)][o] = a.y * wo;
        }
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);
      if (lane < 32) {                                                         // 32 -> 1: row sums, fixed channel order
        float sum = 0.0f;
#pragma unroll
        for (int c = 0; c < U; ++c) sum = sum + tile[lane][c];
        const int e = e0 + lane;
        if (e < eend) heu[e] = sigmoidf(sum + b3[0]);
      }
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    // S x P: K-slot r of step kk is the edge 8r + kk of the tile (pass r's eight edges: the lane's products of a channel are
    // eight consecutive floats of one row)
    {
      const uint32_t eb = (uint32_t)(e0 + 8 * r16 - elo);                       // (an edge before the node's range wraps to a large number)
      const float *pb = &tile[8 * r16 + (i16 >> 2)][(i16 & 3) * 8];
      const float4 p0 = *reinterpret_cast<const float4 *>(pb), p1 = *reinterpret_cast<const float4 *>(pb + 4);
      const float4 p2 = *reinterpret_cast<const float4 *>(pb + 4 * 36), p3 = *reinterpret_cast<const float4 *>(pb + 4 * 36 + 4);
      const float pa[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
      const float pc[8] = {p2.x, p2.y, p2.z, p2.w, p3.x, p3.y, p3.z, p3.w};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const float sel = eb + (uint32_t)kk < edeg ? 1.0f : 0.0f;
        aggA = __builtin_amdgcn_mfma_f32_16x16x4f32(sel, pa[kk], aggA, 0, 0, 0);
        aggB = __builtin_amdgcn_mfma_f32_16x16x4f32(sel, pc[kk], aggB, 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if constexpr (HEAD) return;                                                   // (layer 12's node state feeds nothing)
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {                                              // (D: lane holds nodes 4 r + rr, channel i / 16 + i)
    agg_s[wave][4 * r16 + rr][i16] = aggA[rr];
    agg_s[wave][4 * r16 + rr][16 + i16] = aggB[rr];
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  // the wave's nodes.  Every wavefront of the launch reaches this point at about the same time (one round of workgroups), so
  // nothing hides a load's latency here but the wavefront's own other loads: (1) x' = x0 + silu(bn_v(x1 + mean)) for ALL of
  // the wave's nodes with their rows in flight together (lane = channel, half-wave h takes the nodes 2t + h; x' replaces the
  // aggregate in LDS); (2) the next layer's node linears for eight nodes at a time, the weights sixteen loads per batch and
  // x' read back from LDS as broadcasts -- the same fma order per output as the first version's four-at-a-time loop, which
  // waited for memory once per node and once per input channel (15 us of a layer's 109).
  const float *WT = params + off_layer(feats, layer + 1), *bv = WT + 32 * 128;
  {
    constexpr int T = F2_MAX_NPW / 2;
    float xv[T], x0v[T];
    const float svo = sv[o], tvo = tv[o];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      xv[t] = x0v[t] = 0.0f;
      if (2 * t < cnt) {
        const int i = i0 + min(2 * t + h, cnt - 1);
        xv[t] = X[(size_t)i * 128 + o];
        x0v[t] = x0[(size_t)i * U + o];
      }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (2 * t < cnt) {
        const int d0 = row_at(2 * t + 1) - row_at(2 * t), d1 = row_at(2 * t + 2) - row_at(2 * t + 1);
        const int jn = 2 * t + h;
        if (jn < cnt) {
          const float agg = agg_s[wave][jn][o] / (float)max(h ? d1 : d0, 1);
          const float y = fmaf(xv[t] + agg, svo, tvo);
          const float xn = x0v[t] + silu(y);
          x1out[(size_t)(i0 + jn) * U + o] = xn;
          agg_s[wave][jn][o] = xn;
        }
      }
    }
  }
  if (layer == 11) return;
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  for (int j0 = 0; j0 < cnt; j0 += 8) {
    float y0[8], y1[8];
    {
      const float b0 = bv[lane], b1 = bv[64 + lane];
#pragma unroll
      for (int j = 0; j < 8; ++j) { y0[j] = b0; y1[j] = b1; }
    }
#pragma unroll 2
    for (int c4 = 0; c4 < U; c4 += 4) {
      float wa[4], wb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { wa[k] = WT[(c4 + k) * 128 + lane]; wb[k] = WT[(c4 + k) * 128 + 64 + lane]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 xq = *reinterpret_cast<const float4 *>(&agg_s[wave][j0 + j][c4]);     // (rows past cnt: zeros, not stored)
        const float xc[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          y0[j] = fmaf(xc[k], wa[k], y0[j]);
          y1[j] = fmaf(xc[k], wb[k], y1[j]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j0 + j < cnt) {
        Xnext[(size_t)(i0 + j0 + j) * 128 + lane] = y0[j];
        Xnext[(size_t)(i0 + j0 + j) * 128 + 64 + lane] = y1[j];
      }
  }
}

// head: heu = sigmoid(W3 silu(W2 silu(W1 w + b1) + b2) + b3), three chained tile GEMMs
__global__ void __launch_bounds__(256)
gnn_head_kernel(int E, int feats, const float *params, const float *w, float *heu) {
  __shared__ __attribute__((aligned(16))) float tile[4][32][36];   // 36: keeps float4 rows 16-B aligned
  const float *hp = params + off_head(feats);
  const float *W1 = hp, *b1 = W1 + 1024, *W2 = b1 + 32, *b2 = W2 + 1024, *W3 = b2 + 32, *b3 = W3 + 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e0 = (blockIdx.x * 4 + wave) * 32;
  if (e0 >= E) return;
  const int o = lane & 31;
  float (*t)[36] = tile[wave];
  {                                                       // the 32 x 32 input tile: coalesced 1 KiB runs into LDS
    const int c0 = (lane & 7) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int el = q * 8 + (lane >> 3), e = min(e0 + el, E - 1);
      *reinterpret_cast<float4 *>(&t[el][c0]) = *reinterpret_cast<const float4 *>(w + (size_t)e * U + c0);
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  f32x16 acc = tile_gemm(&t[o][0], W1, lane);
  __builtin_amdgcn_wave_barrier();
  {
    const float bo = b1[o];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 v = {acc[r] + bo, acc[r + 1] + bo};
      const f32x2 a = silu2_fast(v);
      t[drow(r, lane)][o] = a.x; t[drow(r + 1, lane)][o] = a.y;
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): LDS writes of this wave landed
  acc = tile_gemm(&t[o][0], W2, lane);
  __builtin_amdgcn_wave_barrier();
  {
    const float bo = b2[o], wo = W3[o];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 v = {acc[r] + bo, acc[r + 1] + bo};
      const f32x2 a = silu2_fast(v);
      t[drow(r, lane)][o] = a.x * wo; t[drow(r + 1, lane)][o] = a.y * wo;
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  // final 32 -> 1: row sums of the tile, one edge per lane (lanes 0..31), fixed channel order
  if (lane < 32) {
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < U; ++c) s = s + t[lane][c];
    const int e = e0 + lane;
    if (e < E) heu[e] = sigmoidf(s + b3[0]);
  }
}

}  // namespace daco

using namespace daco;

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t daco_gnn_param_floats(int feats) { return off_head(feats) + HEAD_FLOATS; }

extern "C" size_t daco_gnn_workspace_bytes(int n, int E) {
  if (n <= 0 || E <= 0) return 0;
  return 2 * align256((size_t)n * 32 * 4) + 2 * align256((size_t)n * 128 * 4) + 2 * align256((size_t)E * 32 * 4);
}

extern "C" int daco_gnn_forward(void *stream, int n, int E, int feats, const float *x, const int32_t *src,
                                const int32_t *dst, const int32_t *rowptr, const int32_t *perm,
                                const float *edge_attr, const float *params, float *heu, float *emb,
                                void *workspace, size_t workspace_bytes) {
  if (n <= 0 || E <= 0 || feats < 1 || feats > 8 || !x || !src || !dst || !rowptr || !edge_attr || !params || !heu || !workspace) {
    set_error("daco_gnn_forward: bad argument (n=%d E=%d feats=%d)", n, E, feats);
    return DACO_E_BADARG;
  }
  const size_t need = daco_gnn_workspace_bytes(n, E);
  if (workspace_bytes < need) { set_error("daco_gnn_forward: workspace %zu < %zu", workspace_bytes, need); return DACO_E_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  char *p = (char *)workspace;
  float *xb[2], *Xb[2], *wb[2];
  for (int k = 0; k < 2; ++k) { xb[k] = (float *)p; p += align256((size_t)n * 32 * 4); }
  for (int k = 0; k < 2; ++k) { Xb[k] = (float *)p; p += align256((size_t)n * 128 * 4); }
  for (int k = 0; k < 2; ++k) { wb[k] = (float *)p; p += align256((size_t)E * 32 * 4); }
  const int node_blocks = (n + 7) / 8, edge_blocks = (E + 127) / 128;
  hipLaunchKernelGGL(gnn_node_init_kernel, dim3(node_blocks), dim3(256), 0, s, n, feats, x, params, xb[0], Xb[0]);
  int cur = 0;
  // tuning / test knobs, read per call: the edge count from which a layer leaves the single-launch kernel, and the fused
  // kernel's nodes per wave (0: the split edge | node kernels)
  const int split_min = getenv("DACO_GNN_SPLIT_MIN_EDGES") ? atoi(getenv("DACO_GNN_SPLIT_MIN_EDGES")) : GNN_SPLIT_MIN_EDGES;
  const int fused_npw = getenv("DACO_GNN_FUSED_NPW") ? atoi(getenv("DACO_GNN_FUSED_NPW")) : -1;
  // second version of the fused kernel unless the edge array outgrows 32-bit byte offsets (DACO_GNN_FUSED_V=1: the first)
  const int fused_v = (getenv("DACO_GNN_FUSED_V") && atoi(getenv("DACO_GNN_FUSED_V")) == 1) || (size_t)E * 128 >= ((size_t)1 << 32) || (size_t)n * 512 >= ((size_t)1 << 32) ? 1 : 2;
  // nodes per wave of the fused kernel: the launch should fill the device's wave slots (3 per SIMD at its register
  // budget) a whole number of times -- 1.3 rounds of workgroups cost as much as 2 -- with at most 16 nodes per wave
  int npw = fused_npw;
  if (npw < 0) {
    static int slots = 0;
    if (!slots) {
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
      slots = cus * 4;
    }
    const int slots_v = slots * (fused_v == 2 ? 4 : 3);                 // wave slots: 4 (second version) / 3 waves per SIMD
    const int rounds = (n + FUSED_MAX_NPW * slots_v - 1) / (FUSED_MAX_NPW * slots_v);
    npw = (n + rounds * slots_v - 1) / (rounds * slots_v);
    if (npw < 4) npw = 4;
  }
  if (npw > FUSED_MAX_NPW) npw = FUSED_MAX_NPW;
  // EXPERIMENT (DACO_GNN_INPLACE=1, fused kernels only): the edge state is updated in place -- every row is read once, by the
  // lane that writes it, before it is written -- so the layers cycle through E * 128 B instead of twice that
  const int nt_rows = getenv("DACO_GNN_NT") ? atoi(getenv("DACO_GNN_NT")) : 1;      // non-temporal edge rows (measured 1.81 -> 1.72 ms per forward)
  const bool fused2 = E >= split_min && !perm && fused_npw != 0 && fused_v == 2;
  // (the second fused kernel makes layer 0's edge state itself; every other path reads it from the init launch)
  if (!fused2) hipLaunchKernelGGL(gnn_edge_init_kernel, dim3((unsigned)(((long)E * 8 + 255) / 256)), dim3(256), 0, s, E, feats, edge_attr, params, wb[0]);
  // the output head inside the last layer's launch (no embedding asked for; DACO_GNN_HEAD_FUSED=0: the separate head launch)
  const bool head_fused = fused2 && !emb && !(getenv("DACO_GNN_HEAD_FUSED") && atoi(getenv("DACO_GNN_HEAD_FUSED")) == 0);
  const bool inplace = getenv("DACO_GNN_INPLACE") && atoi(getenv("DACO_GNN_INPLACE")) == 1 && E >= split_min && !perm && fused_npw != 0;
  int wcur = 0;
  for (int l = 0; l < 12; ++l) {
    float *wout = (l == 11 && emb) ? emb : wb[inplace ? wcur : (wcur ^ 1)];
    if (fused2) {
      const dim3 grid((unsigned)(((n + 4 * npw - 1) / (4 * npw) + 7) / 8 * 8));
      if (l == 0) hipLaunchKernelGGL((gnn_fused2_layer_kernel<true, false>), grid, dim3(256), 0, s, n, E, feats, l, npw, src, dst, rowptr, params, xb[cur],
                                     Xb[cur], wb[wcur], xb[cur ^ 1], Xb[cur ^ 1], wout, edge_attr, nt_rows, nullptr);
      else if (l == 11 && head_fused) hipLaunchKernelGGL((gnn_fused2_layer_kernel<false, true>), grid, dim3(256), 0, s, n, E, feats, l, npw, src, dst, rowptr,
                                                         params, xb[cur], Xb[cur], wb[wcur], xb[cur ^ 1], Xb[cur ^ 1], wout, edge_attr, nt_rows, heu);
      else hipLaunchKernelGGL((gnn_fused2_layer_kernel<false, false>), grid, dim3(256), 0, s, n, E, feats, l, npw, src, dst, rowptr, params, xb[cur],
                              Xb[cur], wb[wcur], xb[cur ^ 1], Xb[cur ^ 1], wout, edge_attr, nt_rows, nullptr);
    } else if (E >= split_min && !perm && fused_npw != 0) {
      hipLaunchKernelGGL(gnn_fused_layer_kernel, dim3((unsigned)(((n + 4 * npw - 1) / (4 * npw) + 7) / 8 * 8)), dim3(256), 0, s, n, E, feats, l,
                         npw, src, dst, rowptr, params, xb[cur], Xb[cur], wb[wcur], xb[cur ^ 1], Xb[cur ^ 1], wout);
    } else if (E < split_min) {
      hipLaunchKernelGGL(gnn_layer_kernel, dim3(edge_blocks + node_blocks), dim3(256), 0, s, n, E, feats, l, edge_blocks, src,
                         dst, rowptr, perm, params, xb[cur], Xb[cur], wb[wcur], xb[cur ^ 1], Xb[cur ^ 1], wout);
    } else {
      hipLaunchKernelGGL(gnn_node_kernel, dim3(node_blocks), dim3(256), 0, s, n, feats, l, dst, rowptr, perm, params, xb[cur],
                         Xb[cur], wb[wcur], xb[cur ^ 1], Xb[cur ^ 1]);
      hipLaunchKernelGGL(gnn_edge_kernel, dim3(edge_blocks), dim3(256), 0, s, E, feats, l, src, dst, params, Xb[cur], wb[wcur],
                         wout);
    }
    if (l == 11 && emb) { wb[wcur ^ 1] = emb; wcur ^= 1; }
    else if (!inplace) wcur ^= 1;
    cur ^= 1;
  }
  // (a head that walks several tiles per wave with the next rows in flight, W1 / W2 in LDS, was measured: 139 us against
  // this kernel's 131 at 64 x TSP-500 -- not kept)
  if (!head_fused) hipLaunchKernelGGL(gnn_head_kernel, dim3(edge_blocks), dim3(256), 0, s, E, feats, params, wb[wcur], heu);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("gnn kernels launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}
