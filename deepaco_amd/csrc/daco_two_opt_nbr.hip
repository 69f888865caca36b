// daco_two_opt_nbr.hip -- best-improvement 2-opt with neighbour-list candidate pruning, one workgroup per tour.
//
// Reference behaviour replaced: tsp_nls/two_opt.py:6-49 (two_opt_once evaluates every pair 1 <= i < j <= n-1 per sweep).
// This kernel applies exactly the reference's moves but evaluates only the pairs that can win.  With
//   change(i,j) = ((a + b) - c) - e,  a = d[t[i-1]][t[j]], b = d[t[i]][t[j+1]], c = d[t[i-1]][t[i]], e = d[t[j]][t[j+1]]
// a pair with a >= c + tol and b >= e + tol has a real-valued change >= 2 tol; the three f32 roundings of the
// reference's expression move it by at most 2.5 ulp(2M) (M = largest off-diagonal |d|), so with tol = 4 ulp(2M) its
// computed change is > 0 and it can never be the strict minimum below the reference's initial delta = 0.  Every other
// pair is evaluated with the reference's own expression (same loads, same three roundings, no FMA) and the minimum is
// taken with ties to the first (i,j) in row-major order -- so the chosen move, the tours and the sweep counts are
// bit-identical to the dense kernels (daco_two_opt.hip) and to the reference; only the work differs.
//
// Candidates come from tables built once per distance matrix (daco_two_opt_prepare):
//   nb[x][k]  = (d[x][v], v) for the k-th nearest v of x (ascending (d, v))        8 bytes per entry
//   rk[x][y]  = #{v : d[x][v] < d[x][y] + tol}                                       uint16
// For the tour edge at positions (m, m+1) = (x, y):
//   side A (row i = m+1):  the rk[x][y] nearest v of x are the t[j] with a < c + tol          -> pairs (m+1, pos[v])
//   side B (column j = m): the rkT[y][x] nearest u of y IN THE TRANSPOSED matrix are the t[i] with b < e + tol
//                                                                                              -> pairs (pos[u], m)
// (for a symmetric matrix the transposed tables are the same tables).  The value stored in the table is one of the
// two loads of the pair; the other is a gather from the matrix.  A sweep is: prefix sum of the 2n candidate counts,
// then per 2048 candidates an expansion of the lists into (list, k) records in LDS and a flat, balanced loop over the
// records (two 8-byte LDS records, one table entry, one matrix gather per candidate), a workgroup arg-min on the packed
// key (ordered change, i, j), the reversal, and the refresh of the records / ranks of the edges p-1 .. q.
// W is a few thousand for tours near a local optimum (perturbation and repair sweeps of the NLS: ~3-6 k at n = 500
// against 125 k pairs) and approaches 2 n^2 / 2 for tours with many long edges: daco_two_opt_auto (daco_two_opt.hip) hands such
// tours to the dense incremental kernel and takes them back when their lists have become short.
#include <cstdio>
#include <cstdlib>

#include "daco_device.h"
#include "../../include/deepaco_hip.h"

namespace daco {

// f32 -> u32 with the same order (negative values reversed, sign flipped)
__device__ inline uint32_t ord_f32(float x) {
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float unord_f32(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

// ------------------------------------------------------------------ tables
__global__ void __launch_bounds__(256)
nbr_maxabs_kernel(int n, const float *dist, long bstride, unsigned char *tabs, size_t tab_stride) {
  const int b = blockIdx.y;
  const float *d = dist + (size_t)b * bstride;
  float m = 0.0f, lo = __builtin_inff();
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < (long)n * n; idx += (long)gridDim.x * 256) {
    const int x = (int)(idx / n), y = (int)(idx - (long)x * n);
    if (x != y) { m = fmaxf(m, fabsf(d[idx])); lo = fminf(lo, d[idx]); }   // (NaN entries are ignored; inf is kept)
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) { m = fmaxf(m, __shfl_xor(m, s, 64)); lo = fminf(lo, __shfl_xor(lo, s, 64)); }
  if ((threadIdx.x & 63) == 0) {
    unsigned int *hdr = reinterpret_cast<unsigned int *>(tabs + (size_t)b * tab_stride);
    atomicMax(hdr, __float_as_uint(m));
    atomicMax(hdr + 1, ~ord_f32(lo));                     // a maximum of the complement = the minimum (header starts zeroed)
  }
}

__device__ inline float nbr_tol(float M) {
  const float v = 2.0f * M;
  if (!(v < 3.0e38f)) return __uint_as_float(0x7f800000u);                   // inf / overflow: every pair is a candidate
  return 4.0f * (__uint_as_float(__float_as_uint(v) & 0x7f800000u) * 1.1920928955078125e-07f);   // 4 ulp(2M)
}

// one workgroup per matrix row: bitonic sort of (ordered d, id) keys in LDS, then the tolerance ranks
__global__ void __launch_bounds__(256)
nbr_sort_rows_kernel(int n, int P2, const float *dist, long bstride, unsigned char *tabs, size_t tab_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t *key = reinterpret_cast<uint64_t *>(smem);                          // P2 keys
  const int b = blockIdx.x / n, x = blockIdx.x - b * n, tid = threadIdx.x;
  const float *row = dist + (size_t)b * bstride + (size_t)x * n;
  unsigned char *tab = tabs + (size_t)b * tab_stride;
  for (int y = tid; y < P2; y += 256)
    key[y] = y < n ? ((uint64_t)ord_f32(row[y]) << 32) | (uint32_t)y : ~(uint64_t)0;
  __syncthreads();
  for (int k = 2; k <= P2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int idx = tid; idx < P2; idx += 256) {
        const int ixj = idx ^ j;
        if (ixj > idx) {
          const uint64_t a = key[idx], c = key[ixj];
          if ((a > c) == ((idx & k) == 0)) { key[idx] = c; key[ixj] = a; }
        }
      }
      __syncthreads();
    }
  NbrEntry *nb = const_cast<NbrEntry *>(nbr_nb(tab)) + (size_t)x * n;
  for (int k = tid; k < n; k += 256) {
    NbrEntry e;
    e.d = unord_f32((uint32_t)(key[k] >> 32));
    e.id = (uint32_t)key[k];
    nb[k] = e;
  }
  const float tol = nbr_tol(__uint_as_float(*reinterpret_cast<const unsigned int *>(tab)));
  uint16_t *rk = const_cast<uint16_t *>(nbr_rk(tab, n)) + (size_t)x * n;
  for (int y = tid; y < n; y += 256) {
    const uint32_t thr = ord_f32(row[y] + tol);
    int lo = 0, hi = n;                                                         // first k with d_k >= thr
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((uint32_t)(key[mid] >> 32) < thr) lo = mid + 1; else hi = mid;
    }
    rk[y] = (uint16_t)lo;
  }
}

// ------------------------------------------------------------------ the search
// LDS: rec[n] {t[k] | t[k+1] << 16, e[k]} | t[n+1] u16 | pos[n] u16 | rA[n] u16 | rB[n] u16 | pre[2n+1] u32 |
//      queue[NBR_QUEUE] u32 | red[NT/64] u64 | wsum[NT/64] u32 | incumbent u32
constexpr int NBR_QUEUE = 2048;               // candidates expanded per pass, one u32 (item << 16 | k) each

// minimum of a 64-bit key over the wave on the DPP network of daco_device.h (no LDS traffic), broadcast from lane 63
template <int CTRL, int ROW_MASK>
__device__ inline void min_step_u64(uint32_t &hi, uint32_t &lo) {
  const uint32_t ohi = (uint32_t)dpp_i<CTRL, ROW_MASK, false>((int)hi, (int)hi);     // lanes without a source read themselves
  const uint32_t olo = (uint32_t)dpp_i<CTRL, ROW_MASK, false>((int)lo, (int)lo);
  if (ohi < hi || (ohi == hi && olo < lo)) { hi = ohi; lo = olo; }
}
__device__ inline uint64_t wave_min_u64(uint64_t v) {
  uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
  min_step_u64<DPP_ROW_SHR(1), 0xF>(hi, lo);
  min_step_u64<DPP_ROW_SHR(2), 0xF>(hi, lo);
  min_step_u64<DPP_ROW_SHR(4), 0xF>(hi, lo);
  min_step_u64<DPP_ROW_SHR(8), 0xF>(hi, lo);
  min_step_u64<DPP_ROW_BCAST15, 0xA>(hi, lo);
  min_step_u64<DPP_ROW_BCAST31, 0xC>(hi, lo);
  return ((uint64_t)(uint32_t)readlane_i((int)hi, 63) << 32) | (uint32_t)readlane_i((int)lo, 63);
}

// SYM (the matrix equals its transpose, one table set): the two lists a tour node is the centre of -- side A of the edge
// leaving it, side B of the edge entering it -- are prefixes of the SAME sorted list, wanted at later resp. earlier tour
// positions.  They are walked once, up to the longer of the two ranks, and every entry goes to the side its position
// selects: half the table loads and position look-ups, and no candidate that is rejected on position alone.
// NT threads per tour: 256 when the device is full (eight tours per CU hide each other's latencies), 1024 when there are fewer
// tours than CUs can hold (training batches, one instance with a few dozen ants): a sweep is then a latency chain whose
// evaluation part shrinks with the threads walking it.
template <bool SYM, int NT>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_num_sgpr(96)))    // 8 waves per SIMD at NT = 256 (800 SGPRs per SIMD)
two_opt_nbr_kernel(int n, int T, const float *dist, long dist_bs, const unsigned char *tabs, const unsigned char *tabsT,
                   size_t tab_stride, uint16_t *tours, long max_iterations, int32_t *sweeps_out, int32_t *state,
                   uint32_t w_switch, int final_pass, unsigned long long *prof) {
  // state / w_switch / final_pass: daco_two_opt_auto's hand-over with the dense kernel (see there).  A tour whose
  // candidate count exceeds w_switch is left for the dense kernel with its sweep count in the state word.
  // (tour index: XCD x walks a contiguous eighth of the tours, so the few instances it works on at a time -- matrix
  // rows and neighbour lists -- stay in its own 4 MiB L2; with the plain numbering every XCD touched every instance:
  // L2 hit rate 0.76, 1.3 GB of fabric reads per launch, waves parked on memory 70 % of the time)
  const int blk = xcd_remap(blockIdx.x, gridDim.x);
  if (state) {
    const int st = state[blk];
    if (st & TWO_OPT_DONE) {
      if (final_pass && threadIdx.x == 0) state[blk] = st & ~TWO_OPT_DONE;
      return;
    }
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int np2 = (n + 2) & ~1;
  int2 *rec = reinterpret_cast<int2 *>(smem);                 // position k: {t[k] | t[k+1] << 16, bits of e[k] = d[t[k]][t[k+1]]}
  uint16_t *t = reinterpret_cast<uint16_t *>(rec + np2);      // n + 1 (t[n] = t[0])
  uint16_t *pos = t + np2;
  uint16_t *rA = pos + np2;
  uint16_t *rB = rA + np2;
  uint32_t *pre = reinterpret_cast<uint32_t *>(rB + np2);     // 2n + 1
  uint32_t *queue = pre + 2 * np2 + 2;                        // NBR_QUEUE records: item << 16 | k
  uint64_t *red = reinterpret_cast<uint64_t *>(queue + NBR_QUEUE);
  constexpr int NW = NT / 64;
  uint32_t *wsum = reinterpret_cast<uint32_t *>(red + NW);
  uint32_t *incumbent = wsum + NW;                             // ordered image of the best change any thread has seen this sweep
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // prof (DACO_TWO_OPT_PROFILE=1, a debugging aid): shader-clock cycles per phase as thread 0 sees them, summed over tours
  unsigned long long tmark = prof ? clock64() : 0;
  auto lap = [&](int slot) {
    if (prof && tid == 0) { const unsigned long long now = clock64(); atomicAdd(prof + slot, now - tmark); tmark = now; }
  };
  const int b = blk / T;
  const float *d = dist + (size_t)b * dist_bs;
  const NbrEntry *nb = nbr_nb(tabs + (size_t)b * tab_stride), *nbT = nbr_nb(tabsT + (size_t)b * tab_stride);
  const uint16_t *rk = nbr_rk(tabs + (size_t)b * tab_stride, n), *rkT = nbr_rk(tabsT + (size_t)b * tab_stride, n);
  uint16_t *tour = tours + (size_t)blk * n;
  // smallest off-diagonal entry: the unknown load of a candidate is at least this
  const float dmin = unord_f32(~reinterpret_cast<const unsigned int *>(tabs + (size_t)b * tab_stride)[1]);

  for (int k = tid; k < n; k += NT) { const uint16_t v = tour[k]; t[k] = v; pos[v] = (uint16_t)k; }
  __syncthreads();
  if (tid == 0) t[n] = t[0];
  __syncthreads();
  // edge m = (t[m], t[m+1]): its record and the candidate counts of its two sides
  auto refresh_edge = [&](int m) {
    const int x = t[m], y = t[m + 1];
    rec[m] = make_int2(x | (y << 16), __float_as_int(d[(size_t)x * n + y]));
    rA[m] = rk[(size_t)x * n + y];
    rB[m] = rkT[(size_t)y * n + x];
  };
  for (int m = tid; m < n; m += NT) refresh_edge(m);
  __syncthreads();

  // candidate lists.  General: item m < n = side A of edge m, item n + m = side B of edge m (all A lists first: the lanes of
  // a wave are then on the same side, except in the one wave that straddles the boundary, and each side's code runs
  // unselected).  SYM: item m = the list of node t[m], m = 0 .. n (t[n] = t[0]: the closing edge's side B), serving side A
  // of edge m and side B of edge m-1.
  const int items = SYM ? n + 1 : 2 * n, ipt = (items + NT - 1) / NT;
  auto count_of = [&](int item) -> uint32_t {
    if (SYM) {
      const uint32_t ca = item <= n - 3 ? rA[item] : 0, cb = item >= 3 ? rB[item - 1] : 0;
      return ca > cb ? ca : cb;
    }
    const int m = item < n ? item : item - n;
    return item >= n ? (m >= 2 ? rB[m] : 0) : (m <= n - 3 ? rA[m] : 0);
  };
  long it = state ? state[blk] : 0;
  bool handed_over = false;
  while (it < max_iterations) {
    // ---- prefix sum of the candidate counts (thread tid owns items tid*ipt ..)
    {
      const int i0 = tid * ipt;
      uint32_t local = 0;
      for (int q = 0; q < ipt; ++q)
        if (i0 + q < items) local += count_of(i0 + q);
      uint32_t inc = local;
#pragma unroll
      for (int s = 1; s < 64; s <<= 1) { const uint32_t o = __shfl_up(inc, s, 64); if (lane >= s) inc += o; }
      if (lane == 63) wsum[wave] = inc;
      __syncthreads();
      uint32_t base = inc - local;
      for (int w = 0; w < wave; ++w) base += wsum[w];
      for (int q = 0; q < ipt; ++q)
        if (i0 + q < items) {
          const uint32_t cq = count_of(i0 + q);
          pre[i0 + q] = base;
          // the first NBR_QUEUE candidates are expanded right here (most sweeps near a local optimum have no more)
          for (uint32_t w = base; w < base + cq && w < (uint32_t)NBR_QUEUE; ++w) queue[w] = ((uint32_t)(i0 + q) << 16) | (w - base);
          base += cq;
        }
      if (tid == NT - 1) pre[items] = base;             // (the last thread's running offset is the total)
      if (tid == 0) *incumbent = ord_f32(0.0f);               // the reference's `delta = 0`: only negative changes can win
      __syncthreads();
    }
    const uint32_t W = pre[items];
    lap(1);                                                   // prefix sum (+ first chunk's expansion)
    if (W > w_switch) { handed_over = true; break; }          // uniform
    ++it;
    if (prof && tid == 0) { atomicAdd(prof + 6, 1ull); atomicAdd(prof + 7, (unsigned long long)W); }
    uint64_t best = ~(uint64_t)0;
    for (uint32_t c0 = 0; c0 < W; c0 += NBR_QUEUE) {
      // ---- expand the items overlapping [c0, c0 + NBR_QUEUE) into (item, k) records (the first chunk: done above)
      if (c0 > 0) {
        __syncthreads();                                      // the previous chunk's records have been consumed
        for (int item = tid; item < items; item += NT) {
          const uint32_t lo = pre[item], hi = pre[item + 1];
          if (hi > c0 && lo < c0 + NBR_QUEUE) {
            const uint32_t from = lo > c0 ? lo : c0, to = hi < c0 + NBR_QUEUE ? hi : c0 + NBR_QUEUE;
            for (uint32_t w = from; w < to; ++w) queue[w - c0] = ((uint32_t)item << 16) | (w - lo);
          }
        }
        __syncthreads();
      }
      // ---- evaluate them: both sides read the record of their own edge m and the record at / before the candidate's position.
      // (Tried and measured no faster: eight candidates per thread with all loads of a stage in flight -- half the
      // occupancy, 10 % slower; requesting the next candidate's table entry before the current gather is consumed -- equal.)
      const uint32_t cnt = W - c0 < (uint32_t)NBR_QUEUE ? W - c0 : (uint32_t)NBR_QUEUE;
      // Incumbent filter: f32 addition and subtraction are monotone, so with the unknown load u >= dmin the computed change
      // ((table + u) - c) - e is >= ((table + dmin) - c) - e =: lb, exactly.  If lb is strictly above a change some
      // candidate has already achieved (or above the reference's initial delta = 0) this candidate can neither be nor tie
      // the minimum, and its matrix gather -- the expensive access: one 128-byte L2 line for 4 bytes -- is skipped.  The
      // incumbent is shared through one LDS word (atomic minimum of the ordered image); which candidates get skipped
      // depends on timing, the minimum that is found does not.
      const uint32_t un = (uint32_t)n;
      for (uint32_t w = tid; w < cnt; w += NT) {
        const uint32_t qr = queue[w];
        const uint32_t item = qr >> 16, k = qr & 0xffff;
        const uint32_t inc = *incumbent;
        if (SYM) {
          const uint32_t m = item;                                                           // list of node t[m]
          const NbrEntry en = nb[(uint32_t)t[m] * un + k];
          const uint32_t pw = pos[en.id];
          if (pw > m + 1) {
            // side A of edge m: i = m + 1, j = pw: a = table, b = d[t[i]][t[j+1]] gathered
            if (m + 3 <= un && k < rA[m]) {
              const int2 r1 = rec[m], r2 = rec[pw];
              const float c = __int_as_float(r1.y), ej = __int_as_float(r2.y);
              const float lb = ((en.d + dmin) - c) - ej;
              if (!(ord_f32(lb) > inc)) {
                const float g = d[((uint32_t)r1.x >> 16) * un + ((uint32_t)r2.x >> 16)];
                const uint32_t oc = ord_f32(((en.d + g) - c) - ej);
                const uint64_t key = ((uint64_t)oc << 32) | ((m + 1) << 16) | pw;
                best = key < best ? key : best;
                if (oc < inc) atomicMin(incumbent, oc);
              }
            }
          } else if (pw >= 1 && pw + 1 < m) {
            // side B of edge m - 1: i = pw, j = m - 1: b = d[t[i]][t[j+1]] = table (by symmetry), a = d[t[i-1]][t[j]] gathered
            if (k < rB[m - 1]) {
              const int2 rm = rec[m - 1], r2 = rec[pw - 1];
              const float c = __int_as_float(r2.y), ej = __int_as_float(rm.y);
              const float lb = ((en.d + dmin) - c) - ej;
              if (!(ord_f32(lb) > inc)) {
                const float g = d[(uint32_t)(r2.x & 0xffff) * un + (uint32_t)(rm.x & 0xffff)];
                const uint32_t oc = ord_f32(((g + en.d) - c) - ej);
                const uint64_t key = ((uint64_t)oc << 32) | (pw << 16) | (m - 1);
                best = key < best ? key : best;
                if (oc < inc) atomicMin(incumbent, oc);
              }
            }
          }
        } else if (item < un) {
          // side A: i = m + 1, j = pos[v]: a = d[t[m]][v] from the table, b = d[t[i]][t[j+1]] gathered
          const uint32_t m = item;
          const int2 r1 = rec[m];
          const NbrEntry en = nb[(uint32_t)(r1.x & 0xffff) * un + k];
          const uint32_t j = pos[en.id];
          const int2 r2 = rec[j];
          const float c = __int_as_float(r1.y), ej = __int_as_float(r2.y);
          const float lb = ((en.d + dmin) - c) - ej;
          if (j > m + 1 && !(ord_f32(lb) > inc)) {
            const float g = d[((uint32_t)r1.x >> 16) * un + ((uint32_t)r2.x >> 16)];
            const uint32_t oc = ord_f32(((en.d + g) - c) - ej);
            const uint64_t key = ((uint64_t)oc << 32) | ((m + 1) << 16) | j;
            best = key < best ? key : best;
            if (oc < inc) atomicMin(incumbent, oc);
          }
        } else {
          // side B: j = m, i = pos[u]: b = d[u][t[m+1]] from the (transposed) table, a = d[t[i-1]][t[j]] gathered
          const uint32_t m = item - un;
          const int2 r1 = rec[m];
          const NbrEntry en = nbT[((uint32_t)r1.x >> 16) * un + k];
          const uint32_t i = pos[en.id];
          const int2 r2 = rec[i > 0 ? i - 1 : 0];
          const float c = __int_as_float(r2.y), ej = __int_as_float(r1.y);
          const float lb = ((en.d + dmin) - c) - ej;
          if (i >= 1 && i < m && !(ord_f32(lb) > inc)) {
            const float g = d[(uint32_t)(r2.x & 0xffff) * un + (uint32_t)(r1.x & 0xffff)];
            const uint32_t oc = ord_f32(((g + en.d) - c) - ej);
            const uint64_t key = ((uint64_t)oc << 32) | (i << 16) | m;
            best = key < best ? key : best;
            if (oc < inc) atomicMin(incumbent, oc);
          }
        }
      }
    }
    lap(2);                                                   // evaluation (this wave's share)
    best = wave_min_u64(best);
    if (lane == 0) red[wave] = best;
    __syncthreads();
    lap(3);                                                   // waiting for the other waves + reduction
    uint64_t g = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) g = red[w] < g ? red[w] : g;
    const float delta = g == ~(uint64_t)0 ? 0.0f : unord_f32((uint32_t)(g >> 32));
    if (!((double)delta < -1e-6)) break;                      // `if delta < -1e-6` (two_opt.py:25); uniform
    const int p = (int)((g >> 16) & 0xffff), q = (int)(g & 0xffff);
    // ---- reverse t[p..q], refresh what depends on it
    const int L = q - p + 1;
    for (int k = tid; k < (L >> 1); k += NT) { const uint16_t u = t[p + k], v = t[q - k]; t[p + k] = v; t[q - k] = u; }
    __syncthreads();
    for (int k = tid; k < L; k += NT) pos[t[p + k]] = (uint16_t)(p + k);
    for (int m = p - 1 + tid; m <= q; m += NT) refresh_edge(m);   // (q <= n-1: edge n-1 ends at t[n] = t[0], unchanged as p >= 1)
    __syncthreads();
    lap(4);                                                   // reversal, positions, records and ranks of edges p-1 .. q
  }
  for (int k = tid; k < n; k += NT) tour[k] = t[k];
  if (sweeps_out && tid == 0) sweeps_out[blk] = (int32_t)it;
  if (state && tid == 0) state[blk] = (int32_t)it | ((handed_over || final_pass) ? 0 : TWO_OPT_DONE);
}

}  // namespace daco

using namespace daco;

extern "C" size_t daco_two_opt_tables_bytes(int B, int n) {
  if (B <= 0 || n < 4) return 0;
  return (size_t)B * nbr_instance_bytes(n);
}

extern "C" int daco_two_opt_prepare(void *stream, int B, int n, const float *dist, long dist_bstride, void *tables,
                                    size_t tables_bytes) {
  if (B <= 0 || n < 4 || !dist || !tables) { set_error("daco_two_opt_prepare: bad argument (B=%d n=%d)", B, n); return DACO_E_BADARG; }
  if (n > 1024) { set_error("daco_two_opt_prepare: n=%d above 1024 (row sort in LDS)", n); return DACO_E_TOOLARGE; }
  if (tables_bytes < daco_two_opt_tables_bytes(B, n)) { set_error("daco_two_opt_prepare: tables too small"); return DACO_E_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  const size_t stride = nbr_instance_bytes(n);
  unsigned char *tabs = (unsigned char *)tables;
  if (zero_async(tabs, NBR_HEADER, s, B, stride) != hipSuccess) { set_error("daco_two_opt_prepare: clearing the headers failed"); return DACO_E_HIP; }
  int P2 = 4;
  while (P2 < n) P2 <<= 1;
  hipLaunchKernelGGL(nbr_maxabs_kernel, dim3(16, B), dim3(256), 0, s, n, dist, dist_bstride, tabs, stride);
  hipLaunchKernelGGL(nbr_sort_rows_kernel, dim3((unsigned)B * n), dim3(256), (size_t)P2 * 8, s, n, P2, dist, dist_bstride, tabs, stride);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("two_opt table kernels launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}

namespace daco {
int launch_two_opt_nbr(hipStream_t s, int B, int T, int n, const float *dist, long dist_bstride, const void *tables,
                       const void *tables_T, uint16_t *tours, long max_iterations, int32_t *sweeps, int32_t *state,
                       uint32_t w_switch, int final_pass) {
  const int np2 = (n + 2) & ~1;
  const size_t lds = (size_t)np2 * 8 + (size_t)np2 * 2 * 4 + (size_t)(2 * np2 + 2) * 4 + (size_t)NBR_QUEUE * 4 + 16 * 8 + 16 * 4 + 32;
  unsigned long long *prof = nullptr;
  if (getenv("DACO_TWO_OPT_PROFILE")) {                       // debugging aid: synchronises and prints
    if (hipMalloc((void **)&prof, 8 * sizeof(unsigned long long)) != hipSuccess) prof = nullptr;
    else (void)hipMemsetAsync(prof, 0, 8 * sizeof(unsigned long long), s);
  }
  // fewer tours than two per CU: 1024 threads each
  int wide = (long)B * T <= 512;
  if (const char *ev = getenv("DACO_TWO_OPT_WIDE")) wide = atoi(ev);
#define DACO_NBR_LAUNCH(SYM_, NT_)                                                                                              \
  hipLaunchKernelGGL((two_opt_nbr_kernel<SYM_, NT_>), dim3((unsigned)B * T), dim3(NT_), lds, s, n, T, dist, dist_bstride,      \
                     (const unsigned char *)tables, (const unsigned char *)tables_T, nbr_instance_bytes(n), tours,              \
                     max_iterations, sweeps, state, w_switch, final_pass, prof)
  if (tables == tables_T) { if (wide) DACO_NBR_LAUNCH(true, 1024); else DACO_NBR_LAUNCH(true, 256); }
  else { if (wide) DACO_NBR_LAUNCH(false, 1024); else DACO_NBR_LAUNCH(false, 256); }
#undef DACO_NBR_LAUNCH
  hipError_t e = hipGetLastError();
  if (prof) {
    unsigned long long h[8] = {0};
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(prof);
    fprintf(stderr, "[two_opt_nbr profile] tours %d sweeps %llu candidates/sweep %.0f | cycles per sweep: prefix %.0f eval %.0f reduce %.0f "
            "apply %.0f | set-up per tour %.0f\n", B * T, h[6], h[6] ? (double)h[7] / h[6] : 0.0, h[6] ? (double)h[1] / h[6] : 0.0,
            h[6] ? (double)h[2] / h[6] : 0.0, h[6] ? (double)h[3] / h[6] : 0.0, h[6] ? (double)h[4] / h[6] : 0.0, (double)h[0] / (B * T));
  }
  if (e != hipSuccess) { set_error("two_opt_nbr_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}
}  // namespace daco

extern "C" int daco_two_opt_nbr(void *stream, int B, int T, int n, const float *dist, long dist_bstride, const void *tables,
                                const void *tables_T, uint16_t *tours, long max_iterations, int32_t *sweeps) {
  if (B <= 0 || T <= 0 || n < 4 || !dist || !tours || !tables || !tables_T || max_iterations < 0) {
    set_error("daco_two_opt_nbr: bad argument (B=%d T=%d n=%d)", B, T, n);
    return DACO_E_BADARG;
  }
  if (n > 1024) { set_error("daco_two_opt_nbr: n=%d above 1024", n); return DACO_E_TOOLARGE; }
  return launch_two_opt_nbr((hipStream_t)stream, B, T, n, dist, dist_bstride, tables, tables_T, tours, max_iterations, sweeps,
                            nullptr, 0xffffffffu, 0);
}
