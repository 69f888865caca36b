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

// ------------------------------------------------------------------ incremental sweeps
// A 2-opt move reverses t[p..q]; change(i,j) only reads positions i-1, i, j, j+1 and the edge
// lengths e[i-1], e[j], so after the move
//   rows i in [p, q+1]            change completely                          -> recompute the row
//   rows i <  p                   change only at j in [p-1, q]               -> re-evaluate that range
//   rows i >  q+1                 do not change                              -> keep the cached row minimum
// Every row keeps its minimum as one 64-bit key (order-preserving image of the f32 change in the high
// word, j in the low word), so "strict minimum, first in row-major order" is an integer lexicographic
// minimum.  A row below p whose cached minimiser lies in [p-1, q] is recomputed in full; otherwise its new
// minimum is min(cached key, minimum over the changed range).  The values are recomputed with the same
// loads and roundings as a full sweep, so the chosen moves -- and the tours -- are bit-identical to the
// reference; only the number of pair evaluations drops (about 2x on ACO-sampled tours, where most moves
// reverse short segments).
__device__ inline uint32_t ord_f32(float x) {          // monotone f32 -> u32 (x is never NaN here)
  const uint32_t u = __float_as_uint(x + 0.0f);       // -0.0f -> +0.0f
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ inline uint64_t make_key(float change, int j) { return ((uint64_t)ord_f32(change) << 32) | (uint32_t)j; }
constexpr uint64_t KEY_NONE = ~0ull;

template <int W>
__global__ void __launch_bounds__(64 * W)
two_opt_incr_kernel(int n, int T, const float *dist, long dist_bs, uint16_t *tours, long max_iterations,
                    int32_t *sweeps_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int np4 = (n + 3) / 4 * 4;
  int2 *pe = reinterpret_cast<int2 *>(smem);                       // tour records (see two_opt_kernel)
  uint64_t *rb = reinterpret_cast<uint64_t *>(pe + np4);           // per-row minimum key, rows 1..n-2
  int *full_list = reinterpret_cast<int *>(rb + np4);              // rows to recompute
  int *part_list = full_list + np4;                                // rows to patch in [p-1, q]
  uint64_t *red = reinterpret_cast<uint64_t *>(part_list + np4);   // W reduction slots (+ row id)
  int *cnt = reinterpret_cast<int *>(red + 2 * W);                 // [0] full count, [1] partial count
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = 64 * W;
  const int b = blockIdx.x / T;
  const float *d = dist + (size_t)b * dist_bs;
  uint16_t *tour = tours + (size_t)blockIdx.x * n;

  for (int k = tid; k < n; k += NT) {
    const int u = tour[k], v = tour[k + 1 < n ? k + 1 : 0];
    pe[k] = make_int2(u | (v << 16), __float_as_int(d[(size_t)u * n + v]));
  }
  __syncthreads();

  // change(i, j) exactly as tsp_nls/two_opt.py:18-21
  auto pair_change = [&](int na, int nb, float eab, int j) {
    const int2 rec = pe[j];
    const int nc = rec.x & 0xFFFF, nd = (unsigned)rec.x >> 16;
    float change = d[(size_t)na * n + nc] + d[(size_t)nb * n + nd];
    change = change - eab;
    change = change - __int_as_float(rec.y);
    return (na == nc || nd == nb) ? __builtin_inff() : change;
  };

  int p = 1, q = n - 1;                                   // "everything changed" for the first sweep
  bool first = true;
  long it = 0;
  while (it < max_iterations) {
    // ---- classify the rows
    if (tid < 2) cnt[tid] = 0;
    __syncthreads();
    for (int i = 1 + tid; i < n - 1; i += NT) {
      bool full = first || (i >= p && i <= q + 1);
      bool part = false;
      if (!full && i < p) {
        const int jb = (int)(uint32_t)rb[i];
        if (rb[i] != KEY_NONE && jb >= p - 1 && jb <= q) full = true; else part = true;
      }
      if (full) full_list[atomicAdd(&cnt[0], 1)] = i;
      else if (part) part_list[atomicAdd(&cnt[1], 1)] = i;
    }
    __syncthreads();
    const int nfull = cnt[0], npart = cnt[1];
    // ---- rows recomputed in full: one wave per row, lanes stride j
    for (int r = wave; r < nfull; r += W) {
      const int i = full_list[r];
      const int2 ri = pe[i - 1];
      const int na = ri.x & 0xFFFF, nb = (unsigned)ri.x >> 16;
      const float eab = __int_as_float(ri.y);
      float bk = __builtin_inff();
      int bj = 0x7fffffff;
      for (int j = i + 1 + lane; j < n; j += 64) {
        const float c = pair_change(na, nb, eab, j);
        if (c < bk) { bk = c; bj = j; }
      }
      const KeyIdx w = wave_arg<false>(bk, bj);
      if (lane == 0) rb[i] = w.idx == 0x7fffffff ? KEY_NONE : make_key(w.key, w.idx);
    }
    // ---- rows patched in the changed range [max(i+1, p-1), q]: groups of G lanes per row
    if (npart > 0) {
      const int len = q - (p - 1) + 1;
      int G = 1;
      while (G < len && G < 64) G <<= 1;
      const int per_wave = 64 / G, sub = lane / G, off = lane % G;
      for (int r0 = wave * per_wave; r0 < npart; r0 += W * per_wave) {
        const int r = r0 + sub;
        float bk = __builtin_inff();
        int bj = 0x7fffffff, i = 0;
        if (r < npart) {
          i = part_list[r];
          const int2 ri = pe[i - 1];
          const int na = ri.x & 0xFFFF, nb = (unsigned)ri.x >> 16;
          const float eab = __int_as_float(ri.y);
          const int jlo = max(i + 1, p - 1);
          for (int j = jlo + off; j <= q; j += G) {
            const float c = pair_change(na, nb, eab, j);
            if (c < bk) { bk = c; bj = j; }
          }
        }
        for (int o = 1; o < G; o <<= 1) {                 // lexicographic minimum inside the group
          const float ok = __shfl_xor(bk, o);
          const int oj = __shfl_xor(bj, o);
          if (ok < bk || (ok == bk && oj < bj)) { bk = ok; bj = oj; }
        }
        if (off == 0 && r < npart && bj != 0x7fffffff) {
          const uint64_t k = make_key(bk, bj);
          if (k < rb[i]) rb[i] = k;
        }
      }
    }
    __syncthreads();
    // ---- global minimum over the rows: (key's change, row i) lexicographic, then j from the key
    uint64_t best = KEY_NONE;
    int bi = 0x7fffffff;
    for (int i = 1 + tid; i < n - 1; i += NT) {
      const uint64_t k = rb[i];
      if (k != KEY_NONE && (k >> 32) < (best >> 32)) { best = k; bi = i; }   // rows ascend: first minimum kept
    }
    for (int o = 32; o >= 1; o >>= 1) {
      const uint64_t ok = __shfl_xor(best, o);
      const int oi = __shfl_xor(bi, o);
      if ((ok >> 32) < (best >> 32) || ((ok >> 32) == (best >> 32) && oi < bi)) { best = ok; bi = oi; }
    }
    if (lane == 0) { red[2 * wave] = best; red[2 * wave + 1] = (uint64_t)(uint32_t)bi; }
    __syncthreads();
    best = red[0];
    bi = (int)red[1];
#pragma unroll
    for (int w = 1; w < W; ++w) {
      const uint64_t ok = red[2 * w];
      const int oi = (int)red[2 * w + 1];
      if ((ok >> 32) < (best >> 32) || ((ok >> 32) == (best >> 32) && oi < bi)) { best = ok; bi = oi; }
    }
    ++it;
    // the move qualifies if change < 0 (strict, delta starts at 0) and then if change < -1e-6
    const uint32_t o32 = (uint32_t)(best >> 32);
    const float gk = best == KEY_NONE ? 0.0f : __uint_as_float((o32 >> 31) ? (o32 ^ 0x80000000u) : ~o32);
    if (!(gk < 0.0f) || !((double)gk < -1e-6)) break;
    p = bi;
    q = (int)(uint32_t)best;
    first = false;
    __syncthreads();                                      // everyone has read red[] and the records
    const int half = (q - p + 1) >> 1;
    for (int k = tid; k < half; k += NT) {
      const int x = pe[p + k].x, y = pe[q - k].x;
      pe[p + k].x = (x & 0xFFFF0000) | (y & 0xFFFF);
      pe[q - k].x = (y & 0xFFFF0000) | (x & 0xFFFF);
    }
    __syncthreads();
    for (int k = p - 1 + tid; k <= q; k += NT) {
      const int u = pe[k].x & 0xFFFF;
      const int v = pe[k + 1 < n ? k + 1 : 0].x & 0xFFFF;
      const float len = d[(size_t)u * n + v];
      pe[k].y = __float_as_int(len);
      reinterpret_cast<unsigned short *>(&pe[k].x)[1] = (unsigned short)v;
    }
    __syncthreads();
  }
  for (int k = tid; k < n; k += NT) tour[k] = (uint16_t)(pe[k].x & 0xFFFF);
  if (sweeps_out && tid == 0) sweeps_out[blockIdx.x] = (int32_t)it;
}

// ------------------------------------------------------------------ incremental sweeps, row traffic made explicit
// rocprofv3 counters of two_opt_incr_kernel on config 3 (profiles/r02_pmc_2opt_incr_v1.txt): 12 L2 line requests per
// gather instruction, 23 TB/s out of L2 -- every 64-lane gather d[t[i-1]][t[j]] pulls most of a 2 KB matrix row
// through the 32 KB L1 for 256 useful bytes, and the patch phase (rows i < p, columns j in [p-1, q]) touches a
// different matrix row for every tour row.  Same algorithm and arithmetic as two_opt_incr_kernel, other data movement:
//   * block rows i in [p, q+1] (contiguous): each wave owns a run of them and keeps d[t[i-1]][.] and d[t[i]][.] in LDS,
//     one coalesced 2 KB copy per new row (the row gathered from as d[t[i]][.] is the d[t[i-1]][.] of the next i);
//     both gathers of a pair are LDS reads;
//   * scattered rows recomputed in full (cached minimiser inside the changed range): the same, two copies per row;
//   * patch phase: ONE LANE PER TOUR ROW, lanes walk the changed columns j together and read
//     d[t[i-1]][t[j]] as dT[t[j]][t[i-1]] (dT = the transposed matrix, or d itself when it is symmetric): all 64 lanes
//     of a gather read the SAME matrix row t[j], and only the q-p+3 rows of the changed segment are touched per
//     sweep instead of ~p different ones.  Every lane keeps its own running minimum (j ascends, strict <), so the
//     phase needs no cross-lane reduction.
template <int W>
__global__ void __launch_bounds__(64 * W)
two_opt_incr2_kernel(int n, int T, const float *dist, const float *distT, long dist_bs, uint16_t *tours, long max_iterations,
                     int32_t *sweeps_out, int32_t *state, int budget, const unsigned char *tabs, const unsigned char *tabsT,
                     size_t tab_stride, int w_exit) {
  // state (daco_two_opt_auto's hand-over between this kernel and the candidate-list kernel): sweeps done so far per tour,
  // TWO_OPT_DONE set once the search ended; this launch resumes there and does at most `budget` sweeps
  const int blk = xcd_remap(blockIdx.x, gridDim.x);      // an XCD walks consecutive tours: few instances' rows in its L2 at a time
  if (state && (state[blk] & TWO_OPT_DONE)) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int np4 = (n + 3) / 4 * 4;
  int2 *pe = reinterpret_cast<int2 *>(smem);                       // tour records (see two_opt_kernel)
  uint64_t *rb = reinterpret_cast<uint64_t *>(pe + np4);           // per-row minimum key, rows 1..n-2
  int *full_list = reinterpret_cast<int *>(rb + np4);              // scattered rows to recompute
  int *part_list = full_list + np4;                                // rows to patch in [p-1, q]
  uint64_t *red = reinterpret_cast<uint64_t *>(part_list + np4);   // W reduction slots (+ row id)
  int *cnt = reinterpret_cast<int *>(red + 2 * W);                 // [0] full count, [1] partial count
  float *rows = reinterpret_cast<float *>(cnt + 8);                // W x 2 x np4 staged matrix rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = 64 * W;
  const int b = blk / T;
  const float *d = dist + (size_t)b * dist_bs;
  const float *dT = distT ? distT + (size_t)b * dist_bs : nullptr;
  uint16_t *tour = tours + (size_t)blk * n;
  float *rowA = rows + (size_t)wave * 2 * np4, *rowB = rowA + np4;
  const bool symmetric = distT == dist;                  // (the caller passes the matrix itself as its transpose)

  for (int k = tid; k < n; k += NT) {
    const int u = tour[k], v = tour[k + 1 < n ? k + 1 : 0];
    pe[k] = make_int2(u | (v << 16), __float_as_int(d[(size_t)u * n + v]));
  }
  __syncthreads();
  // tabs (daco_two_opt_auto): the neighbour tables' ranks of the tour's edges, summed = the number of list entries the
  // candidate-list kernel would walk per sweep (x 1/2 when one list serves both sides).  It is kept up to date move by
  // move (symmetric matrix: two edges change; otherwise rebuilt per sweep) and the tour is handed
  // back (the loop below ends, the state word stays unfinished) once it has fallen under w_exit.
  const uint16_t *rk = tabs ? nbr_rk(tabs + (size_t)b * tab_stride, n) : nullptr;
  const uint16_t *rkT = tabs ? nbr_rk(tabsT + (size_t)b * tab_stride, n) : nullptr;
  auto rank_sum = [&](int x, int y) { return (int)rk[(size_t)x * n + y] + (int)rkT[(size_t)y * n + x]; };
  if (tabs) {
    if (tid == 0) cnt[2] = 0;
    __syncthreads();
    int w = 0;
    for (int k = tid; k < n; k += NT) w += rank_sum(pe[k].x & 0xFFFF, (unsigned)pe[k].x >> 16);
    for (int o = 32; o >= 1; o >>= 1) w += __shfl_xor(w, o);
    if (lane == 0) atomicAdd(&cnt[2], w);
    __syncthreads();
  }
  const int w_scale = tabs == tabsT ? 2 : 1;              // symmetric: one walk per node list
  auto stage = [&](float *dst, int node) {                // coalesced copy of row d[node][0..n) into the wave's slot
    const float *src = d + (size_t)node * n;
    for (int k = lane; k < n; k += 64) dst[k] = src[k];
  };
  // minimum of row i over j in (i, n) with both distance rows in LDS; lanes stride j
  auto row_min_staged = [&](int i, const float *gA, const float *gB) {
    const int2 ri = pe[i - 1];
    const int na = ri.x & 0xFFFF, nb = (unsigned)ri.x >> 16;
    const float eab = __int_as_float(ri.y);
    float bk = __builtin_inff();
    int bj = 0x7fffffff;
    for (int j = i + 1 + lane; j < n; j += 64) {
      const int2 rec = pe[j];
      const int nc = rec.x & 0xFFFF, nd = (unsigned)rec.x >> 16;
      float change = gA[nc] + gB[nd];
      change = change - eab;
      change = change - __int_as_float(rec.y);
      if (na == nc || nd == nb) change = __builtin_inff();
      if (change < bk) { bk = change; bj = j; }
    }
    const KeyIdx w = wave_arg<false>(bk, bj);
    if (lane == 0) rb[i] = w.idx == 0x7fffffff ? KEY_NONE : make_key(w.key, w.idx);
  };

  int p = 1, q = n - 1;                                   // "everything changed" for the first sweep
  bool first = true;
  long it = state ? state[blk] : 0;
  const long stop_at = state ? min(max_iterations, it + (long)budget) : max_iterations;
  bool ended = false;
  while (it < stop_at) {
    if (tabs && tabs != tabsT && !first) {
      // non-symmetric matrix: the inner edges of a reversed segment change their ranks too (rank_sum(y, x) != rank_sum(x, y)),
      // so the two-edge update below would drift -- the sum is rebuilt from the records instead (n / NT gathers per thread)
      if (tid == 0) cnt[2] = 0;
      __syncthreads();
      int w = 0;
      for (int k = tid; k < n; k += NT) w += rank_sum(pe[k].x & 0xFFFF, (unsigned)pe[k].x >> 16);
      for (int o = 32; o >= 1; o >>= 1) w += __shfl_xor(w, o);
      if (lane == 0) atomicAdd(&cnt[2], w);
      __syncthreads();
    }
    if (tabs && (long)cnt[2] < (long)w_exit * w_scale) break;         // uniform (cnt[2] was last written before the previous barrier)
    const int blo = first ? 1 : p, bhi = first ? n - 1 : min(q + 2, n - 1);       // block rows [blo, bhi)
    // (the rows below the block were classified while the previous move was applied: lists and counts are ready)
    const int nfull = first ? 0 : cnt[0], npart = first ? 0 : cnt[1];
    // ---- block rows: each wave a contiguous run with (about) the same number of pairs; rolling rows in LDS
    {
      const int R = bhi - blo;
      int lo = blo, hi = bhi;
      if (W > 1 && R > 0) {
        // pairs of row i = n-1-i; cumulative pairs from blo: S(i) = sum_{r=blo}^{i-1} (n-1-r); split S into W equal parts
        const float tot = (float)R * (float)(n - 1 - blo) - 0.5f * (float)R * (float)(R - 1);
        auto bound = [&](int k) {                          // first row whose cumulative pair count reaches k/W of the total
          if (k <= 0) return blo;
          if (k >= W) return bhi;
          const float target = tot * (float)k / (float)W, a = (float)(n - 1 - blo) + 0.5f;
          float x = a - sqrtf(fmaxf(a * a - 2.0f * target, 0.0f));          // solves x*a' - x^2/2 = target
          int r = blo + (int)x;
          return max(blo, min(r, bhi));
        };
        lo = bound(wave); hi = bound(wave + 1);
      }
      if (lo < hi) {
        if (n <= 512) {
          // the next row travels through registers while the current one is evaluated out of LDS (the chain of a
          // sweep is a sequence of L2 round trips: overlap them with the LDS work instead of adding them up)
          float nx[8];
          auto fetch = [&](int node) {
            const float *src = d + (size_t)node * n;
#pragma unroll
            for (int m = 0; m < 8; ++m) { const int k = lane + 64 * m; nx[m] = k < n ? src[k] : 0.0f; }
          };
          auto put = [&](float *dst) {
#pragma unroll
            for (int m = 0; m < 8; ++m) { const int k = lane + 64 * m; if (k < n) dst[k] = nx[m]; }
          };
          fetch(pe[lo - 1].x & 0xFFFF);
          put(rowB);
          fetch((unsigned)pe[lo - 1].x >> 16);
          for (int i = lo; i < hi; ++i) {
            float *tmp = rowA; rowA = rowB; rowB = tmp;
            put(rowB);                                      // d[t[i]][.]
            if (i + 1 < hi) fetch((unsigned)pe[i].x >> 16); // d[t[i+1]][.], in flight during this row's pairs
            row_min_staged(i, rowA, rowB);
          }
        } else {
          stage(rowB, pe[lo - 1].x & 0xFFFF);             // becomes rowA of the first row
          for (int i = lo; i < hi; ++i) {
            float *tmp = rowA; rowA = rowB; rowB = tmp;
            stage(rowB, (unsigned)pe[i - 1].x >> 16);
            row_min_staged(i, rowA, rowB);
          }
        }
      }
    }
    // ---- scattered full rows: one wave per row, both rows staged (prefetching the next pair through registers, as the
    // block rows do, was measured 5 % slower here: these rows are few)
    for (int r = wave; r < nfull; r += W) {
      const int i = full_list[r];
      const int2 ri = pe[i - 1];
      stage(rowA, ri.x & 0xFFFF);
      stage(rowB, (unsigned)ri.x >> 16);
      row_min_staged(i, rowA, rowB);
    }
    // ---- patch phase: one lane per row, all lanes walk j = p-1 .. q together; with the transposed matrix the two
    // values of a pair are dT[t[j]][t[i-1]] and dT[t[j+1]][t[i]], i.e. all 64 lanes of a gather read one matrix row.
    // (Copying the segment's rows into LDS first was tried: the extra barriers and the load -> barrier -> compute
    // chain per chunk made the sweep 1.6x slower; a sweep is bound by its chain of L2 round trips, not by bytes.)
    for (int r = tid; r < ((npart + NT - 1) / NT) * NT; r += NT) {
      const bool act = r < npart;
      const int i = act ? part_list[r] : 1;
      const int2 ri = pe[i - 1];
      const int na = ri.x & 0xFFFF, nb = (unsigned)ri.x >> 16;
      const float eab = __int_as_float(ri.y);
      float bk = __builtin_inff();
      int bj = 0x7fffffff;
      constexpr int UN = 8;                                // columns per batch: 2 * UN gathers in flight per lane
      for (int j0 = p - 1; j0 <= q; j0 += UN) {
        int2 rec[UN];
        float A[UN], Bv[UN];
#pragma unroll
        for (int m = 0; m < UN; ++m) {
          rec[m] = pe[min(j0 + m, q)];                     // (uniform address: one LDS broadcast)
          const int nc = rec[m].x & 0xFFFF, nd = (unsigned)rec[m].x >> 16;
          if (dT) { A[m] = dT[(size_t)nc * n + na]; Bv[m] = dT[(size_t)nd * n + nb]; }
          else { A[m] = d[(size_t)na * n + nc]; Bv[m] = d[(size_t)nb * n + nd]; }
        }
#pragma unroll
        for (int m = 0; m < UN; ++m) {
          const int j = j0 + m;
          const int nc = rec[m].x & 0xFFFF, nd = (unsigned)rec[m].x >> 16;
          float change = A[m] + Bv[m];
          change = change - eab;
          change = change - __int_as_float(rec[m].y);
          if (j > q || j <= i || na == nc || nd == nb) change = __builtin_inff();
          if (change < bk) { bk = change; bj = j; }
        }
      }
      if (act && bj != 0x7fffffff) {
        const uint64_t k = make_key(bk, bj);
        if (k < rb[i]) rb[i] = k;
      }
    }
    __syncthreads();
    // ---- global minimum over the rows: (key's change, row i) lexicographic, then j from the key
    uint64_t best = KEY_NONE;
    int bi = 0x7fffffff;
    for (int i = 1 + tid; i < n - 1; i += NT) {
      const uint64_t k = rb[i];
      if (k != KEY_NONE && (k >> 32) < (best >> 32)) { best = k; bi = i; }   // rows ascend: first minimum kept
    }
    for (int o = 32; o >= 1; o >>= 1) {
      const uint64_t ok = __shfl_xor(best, o);
      const int oi = __shfl_xor(bi, o);
      if ((ok >> 32) < (best >> 32) || ((ok >> 32) == (best >> 32) && oi < bi)) { best = ok; bi = oi; }
    }
    if (lane == 0) { red[2 * wave] = best; red[2 * wave + 1] = (uint64_t)(uint32_t)bi; }
    if (tid < 2) cnt[tid] = 0;                            // (everyone read the counts before the barrier above)
    __syncthreads();
    best = red[0];
    bi = (int)red[1];
#pragma unroll
    for (int w = 1; w < W; ++w) {
      const uint64_t ok = red[2 * w];
      const int oi = (int)red[2 * w + 1];
      if ((ok >> 32) < (best >> 32) || ((ok >> 32) == (best >> 32) && oi < bi)) { best = ok; bi = oi; }
    }
    ++it;
    const uint32_t o32 = (uint32_t)(best >> 32);
    const float gk = best == KEY_NONE ? 0.0f : __uint_as_float((o32 >> 31) ? (o32 ^ 0x80000000u) : ~o32);
    if (!(gk < 0.0f) || !((double)gk < -1e-6)) { ended = true; break; }
    p = bi;
    q = (int)(uint32_t)best;
    first = false;
    // ---- apply the move and, in the same phase, classify the rows below the block for the next sweep: recompute in
    // full (cached minimiser inside the changed range) or patch.  (pe was last read before the previous barrier.)
    const int half = (q - p + 1) >> 1;
    if (tabs && tabs == tabsT && tid == 0) {
      // symmetric matrix (tabs == tabsT; only then: the non-symmetric case rebuilds cnt[2] from all n edges at the top of every
      // sweep, above): a reversal changes the ranks of edges p-1 and q only, the inner edges swap sides
      // (thread 0 owns the swap of positions p and q below; t[p-1] and t[q+1] are not written in this phase)
      const int a = pe[p - 1].x & 0xFFFF, bq = pe[p].x & 0xFFFF, c = pe[q].x & 0xFFFF, dn = pe[q + 1 < n ? q + 1 : 0].x & 0xFFFF;
      atomicAdd(&cnt[2], rank_sum(a, c) + rank_sum(bq, dn) - rank_sum(a, bq) - rank_sum(c, dn));
    }
    for (int k = tid; k < half; k += NT) {
      const int x = pe[p + k].x, y = pe[q - k].x;
      pe[p + k].x = (x & 0xFFFF0000) | (y & 0xFFFF);
      pe[q - k].x = (y & 0xFFFF0000) | (x & 0xFFFF);
    }
    if (symmetric) {
      // the inner edges are the old ones walked backwards: same lengths, mirrored (e'[k] = e[p+q-1-k], k in [p, q-1])
      const int ehalf = (q - p) >> 1;
      for (int k = tid; k < ehalf; k += NT) {
        const int x = pe[p + k].y, y = pe[q - 1 - k].y;
        pe[p + k].y = y;
        pe[q - 1 - k].y = x;
      }
    }
    for (int i = 1 + tid; i < p; i += NT) {
      const uint64_t key = rb[i];
      const int jb = (int)(uint32_t)key;
      if (key != KEY_NONE && jb >= p - 1 && jb <= q) full_list[atomicAdd(&cnt[0], 1)] = i;
      else part_list[atomicAdd(&cnt[1], 1)] = i;
    }
    __syncthreads();
    for (int k = p - 1 + tid; k <= q; k += NT) {          // records p-1 .. q: successor (+ edge length where it is new)
      const int u = pe[k].x & 0xFFFF;
      const int v = pe[k + 1 < n ? k + 1 : 0].x & 0xFFFF;
      if (!symmetric || k == p - 1 || k == q) pe[k].y = __float_as_int(d[(size_t)u * n + v]);
      reinterpret_cast<unsigned short *>(&pe[k].x)[1] = (unsigned short)v;
    }
    __syncthreads();
  }
  for (int k = tid; k < n; k += NT) tour[k] = (uint16_t)(pe[k].x & 0xFFFF);
  if (sweeps_out && tid == 0) sweeps_out[blk] = (int32_t)it;
  if (state && tid == 0) state[blk] = (int32_t)it | ((ended || it >= max_iterations) ? TWO_OPT_DONE : 0);
}

}  // namespace daco

using namespace daco;

extern "C" int daco_two_opt(void *stream, int B, int T, int n, const float *dist, const float *dist_T, long dist_bstride,
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
  // default: the incremental kernel (16 + log2(waves per tour)); the full-sweep kernels (variant = waves per
  // tour * 2 + staged) stay selectable with DACO_TWO_OPT_VARIANT for tuning / cross-checks.  Measured on the
  // NLS workload (tools/measure_configs.py c3:n): n=100 25 vs 45 ms, n=200 62 vs 189 ms, n=500 507 vs 1985 ms.
  // 32: two_opt_incr2_kernel (explicit row traffic; dist_T lets its patch phase read matrix rows of the changed segment)
  int variant = n <= 128 ? 17 : 32;
  if (const char *ev = getenv("DACO_TWO_OPT_VARIANT")) variant = atoi(ev);
#define DACO_2OPT(W, ST) hipLaunchKernelGGL((two_opt_kernel<W, ST>), dim3(B * T), dim3(64 * W), lds_bytes(W, ST), s, n, T, dist, dist_bstride, tours, max_iterations, sweeps)
  auto lds_incr = [&](int W) { return (2 * np4 + 2 * np4 + np4 + np4 + 4 * W + 2 + 6) * sizeof(int); };
  auto lds_incr2 = [&](int W) { return lds_incr(W) + 8 * sizeof(int) + (size_t)W * 2 * np4 * sizeof(float); };
  switch (variant) {
    case 32: hipLaunchKernelGGL((two_opt_incr2_kernel<4>), dim3(B * T), dim3(256), lds_incr2(4), s, n, T, dist, dist_T, dist_bstride, tours, max_iterations, sweeps, (int32_t *)nullptr, 0, (const unsigned char *)nullptr, (const unsigned char *)nullptr, (size_t)0, 0); break;
    case 33: hipLaunchKernelGGL((two_opt_incr2_kernel<2>), dim3(B * T), dim3(128), lds_incr2(2), s, n, T, dist, dist_T, dist_bstride, tours, max_iterations, sweeps, (int32_t *)nullptr, 0, (const unsigned char *)nullptr, (const unsigned char *)nullptr, (size_t)0, 0); break;
    case 16: hipLaunchKernelGGL((two_opt_incr_kernel<1>), dim3(B * T), dim3(64), lds_incr(1), s, n, T, dist, dist_bstride, tours, max_iterations, sweeps); break;
    case 17: hipLaunchKernelGGL((two_opt_incr_kernel<2>), dim3(B * T), dim3(128), lds_incr(2), s, n, T, dist, dist_bstride, tours, max_iterations, sweeps); break;
    case 18: hipLaunchKernelGGL((two_opt_incr_kernel<4>), dim3(B * T), dim3(256), lds_incr(4), s, n, T, dist, dist_bstride, tours, max_iterations, sweeps); break;
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

namespace daco {
int launch_two_opt_nbr(hipStream_t s, int B, int T, int n, const float *dist, long dist_bstride, const void *tables,
                       const void *tables_T, uint16_t *tours, long max_iterations, int32_t *sweeps, int32_t *state,
                       uint32_t w_switch, int final_pass);
static size_t tables_bytes_per_instance(int n) { return nbr_instance_bytes(n); }
}

// Both kernels on one call: the candidate-list kernel while a tour's candidate count W stays below w_switch, the dense
// incremental kernel in slices of `slice` sweeps while it is above (tours fresh from sampling: every pair is a candidate),
// alternating without a host round trip -- a per-tour state word carries the sweep count across launches, finished
// tours leave at once.  Whatever is still unfinished after the last slice is completed by the candidate-list kernel.
extern "C" int daco_two_opt_auto(void *stream, int B, int T, int n, const float *dist, const float *dist_T, long dist_bstride,
                                 const void *tables, const void *tables_T, uint16_t *tours, long max_iterations,
                                 int32_t *sweeps) {
  if (B <= 0 || T <= 0 || n < 4 || !dist || !tours || !tables || !tables_T || !sweeps || max_iterations < 0) {
    set_error("daco_two_opt_auto: bad argument (B=%d T=%d n=%d)", B, T, n);
    return DACO_E_BADARG;
  }
  if (n > 1024) { set_error("daco_two_opt_auto: n=%d above 1024", n); return DACO_E_TOOLARGE; }
  if (max_iterations >= TWO_OPT_DONE) max_iterations = TWO_OPT_DONE - 1;
  hipStream_t s = (hipStream_t)stream;
  if (zero_async(sweeps, (size_t)B * T * sizeof(int32_t), s) != hipSuccess) { set_error("clearing the sweep counts failed"); return DACO_E_HIP; }
  // measured crossovers at n = 500 (tools/bench_two_opt_nbr.py, profiles/r02_two_opt_nbr.txt): the dense kernel's cost grows with
  // the length of the reversed segment, the candidate kernel's with the list entries it walks.  Symmetric matrix (every
  // entry is one candidate; repairs reverse long segments): the dense kernel wins from ~40 k entries per sweep; the 20
  // perturbation sweeps on the heuristic-derived (non-symmetric: two lists per edge) matrix reverse short segments: ~25 k.
  uint32_t w_switch = (uint32_t)((double)n * n / (tables == tables_T ? 6.0 : 10.0));
  // ... when the device is full.  With at most one workgroup round (256 CUs x 8) a sweep is latency, not throughput, and the
  // candidate kernel's chain is several times shorter than the dense kernel's at any list length (training steps, 240-600
  // tours: 13.6 -> 9.4 ms at TSP-100, 68 -> 54 ms at TSP-500 without the dense kernel)
  if ((long)B * T <= 2048) w_switch = 0xffffffffu;
  if (const char *ev = getenv("DACO_TWO_OPT_SWITCH")) w_switch = (uint32_t)atol(ev);
  // hand-over hysteresis: a tour goes to the dense kernel above w_switch walked entries per sweep and comes back below half
  uint32_t w_back = w_switch / 2;
  if (const char *ev = getenv("DACO_TWO_OPT_BACK")) w_back = (uint32_t)atol(ev);
  const size_t np4 = (size_t)(n + 3) / 4 * 4;
  const size_t lds = (2 * np4 + 2 * np4 + np4 + np4 + 4 * 4 + 2 + 6) * sizeof(int) + 8 * sizeof(int) + (size_t)4 * 2 * np4 * sizeof(float);
  int rc = launch_two_opt_nbr(s, B, T, n, dist, dist_bstride, tables, tables_T, tours, max_iterations, nullptr, sweeps, w_switch, 0);
  if (rc != DACO_OK) return rc;
  hipLaunchKernelGGL((two_opt_incr2_kernel<4>), dim3(B * T), dim3(256), lds, s, n, T, dist, dist_T, dist_bstride, tours,
                     max_iterations, (int32_t *)nullptr, sweeps, (int)(TWO_OPT_DONE - 1), (const unsigned char *)tables,
                     (const unsigned char *)tables_T, tables_bytes_per_instance(n), (int)w_back);
  rc = launch_two_opt_nbr(s, B, T, n, dist, dist_bstride, tables, tables_T, tours, max_iterations, nullptr, sweeps, 0xffffffffu, 1);
  if (rc != DACO_OK) return rc;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("two_opt_auto launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}
