// daco_nls.hip -- the whole neural-guided local search of a tour in ONE launch, on candidate lists whose per-list minima
// are cached from sweep to sweep.
//
// Reference behaviour replaced: tsp_nls/aco.py:241-258 (ACO.nls: 2-opt, then T_nls x { T_p perturbation sweeps on the
// heuristic-derived matrix, 2-opt repair, keep the tour if it got shorter }) over tsp_nls/two_opt.py:6-39 (two_opt_once /
// _two_opt_python).  Round 2 ran this as 21 host-driven passes (63 launches + the cost / keep-best glue); a pass of the
// candidate-list kernel (daco_two_opt_nbr.hip, whose tables, tolerance rule and pair arithmetic this file shares) re-evaluated
// every list in every sweep.
//
// Two changes, neither of which touches a result:
//  (1) One workgroup owns a tour for all passes: first pass, the perturbation / repair rounds, the f32 tour lengths
//      (daco_tour_costs' summation order) and the `new < best` comparison all happen here; a tour's search depends on no
//      other tour, so there is nothing to wait for and the slowest tour of a pass no longer holds up the next pass.
//  (2) Dirty lists.  change(i,j) reads the nodes at positions i-1, i, j, j+1 and nothing else.  After the move (p,q)
//      (reverse t[p..q]) the nodes at positions outside [p,q] have not moved and the tour edges outside [p-1,q] are what
//      they were, so the candidates of a list -- same prefix of the same sorted neighbour list, same positions, same four
//      loads -- evaluate to the same bits unless the list's own position lies in [p-1,q+1] or one of the entries it walks
//      does.  Every list therefore keeps the minimum key of its candidates (ordered change | i | j) and a 64-bit signature
//      of the position buckets its walked entries fall into; a sweep re-walks only the lists whose position is in the
//      range or whose signature meets the range's buckets (conservative: a false positive is re-evaluated to the same
//      value), and takes the minimum over all cached keys.  Moves near a local optimum mostly reverse short segments
//      (median 3-9 positions at n = 500 in the NLS's repair passes, measured with the oracle): a sweep then walks a few
//      hundred list entries instead of a few thousand.
// The incumbent filter of daco_two_opt_nbr.hip (skip the matrix gather of a candidate whose lower bound exceeds what
// another candidate has already achieved in this sweep) is not used here: a cached list minimum has to be the minimum of
// the whole list, whatever the other lists held when it was computed.
#include <cstdio>
#include <cstdlib>

#include "daco_device.h"
#include "../../include/deepaco_hip.h"

namespace daco {

__device__ inline uint32_t nls_ord_f32(float x) {
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float nls_unord_f32(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

// minimum of an unsigned value over the wave (in lane 63, read back as a wave-uniform value): six v_min_u32 on the DPP network
__device__ inline uint32_t nls_wave_min_u32(uint32_t x) {
  // (lanes without a source keep their value: the instruction is disabled there; a DPP read needs two wait states after
  // the VALU write of its source)
  asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
               "s_nop 1"
               : "+v"(x));
  return (uint32_t)readlane_i((int)x, 63);
}
// minimum of 64-bit keys over the wave, lexicographically: the smallest high word first, then the smallest low word among the
// lanes that hold it (two 32-bit reductions instead of one with a 64-bit compare-and-select per step)
__device__ inline uint64_t nls_wave_min(uint64_t v) {
  const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
  const uint32_t mh = nls_wave_min_u32(hi);
  const uint32_t ml = nls_wave_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
  return ((uint64_t)mh << 32) | ml;
}

// inclusive integer add-scan over the wave: rows of 16 (row_shr), then the rows' totals (row_bcast 15 / 31)
__device__ inline uint32_t nls_wave_scan_u32(uint32_t v) {
  int x = (int)v;
  x += dpp_i<DPP_ROW_SHR(1), 0xF, true>(0, x);
  x += dpp_i<DPP_ROW_SHR(2), 0xF, true>(0, x);
  x += dpp_i<DPP_ROW_SHR(4), 0xF, true>(0, x);
  x += dpp_i<DPP_ROW_SHR(8), 0xF, true>(0, x);
  x += dpp_i<DPP_ROW_BCAST15, 0xA, false>(0, x);
  x += dpp_i<DPP_ROW_BCAST31, 0xC, false>(0, x);
  return (uint32_t)x;
}

// row * n + column with the 24-bit multiplier (n <= 1024: every index is below 2^21; v_mad_u32_u24 runs at full rate, the 32-bit
// v_mul_lo_u32 / v_mad_u64_u32 the compiler picks otherwise at a quarter of it -- four of them per walked entry)
__device__ inline uint32_t nls_idx(uint32_t row, uint32_t n, uint32_t col) { return __umul24(row, n) + col; }

// loads at 32-bit byte offsets from a wave-uniform base (n <= 1024: every table is below 2^24 bytes)
template <typename T>
__device__ inline T nls_ld(const T *base, uint32_t idx) {
  return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (uint32_t)(idx * (uint32_t)sizeof(T)));
}

constexpr uint64_t NLS_NONE = ~(uint64_t)0;
typedef short nls_s2 __attribute__((ext_vector_type(2)));
// the owner of a flat entry without a search: the compacted lists' first entries as a bitmap of 64 words, one word and its
// prefix count per lane (a sweep near a local optimum walks a few hundred entries; above NLS_OB_MAX the binary search stays)
constexpr uint32_t NLS_OB_WORDS = 64, NLS_OB_MAX = NLS_OB_WORDS * 32;
#ifndef NLS_WAVES_256
#define NLS_WAVES_256 6                       // waves per SIMD the 256-thread variant is compiled for (80 VGPRs)
#endif

struct NlsMatrix {                            // one matrix of the search: distances or the perturbation matrix
  const float *d;
  const NbrEntry *nb, *nbT;
  const uint16_t *rk, *rkT;
};

// LDS of one tour
struct NlsLds {
  int2 *rec;                                  // position k: {t[k] | t[k+1] << 16, bits of e[k] = d[t[k]][t[k+1]]}
  uint16_t *t, *pos, *rA, *rB;                // t[n] = t[0]; rA / rB: list lengths of the two sides of edge m
  uint32_t *pre;                              // offsets of the (compacted) non-empty dirty lists' entries, D + 1 values
  uint16_t *ditem;                            // their items
  uint64_t *ckey, *sig;                       // per list: minimum key of its candidates, position buckets of its walked entries
  uint64_t *red;
  uint32_t *wsum;
  uint32_t *obits;                            // NLS_OB_WORDS words: bit w set = a compacted dirty list starts at flat entry w (sweeps of <= NLS_OB_MAX entries)
  float *scal;                                // [0] tour length broadcast
  unsigned long long *lap;                    // profile: cycles per phase of this tour (thread 0)
};

// One 2-opt search (two_opt.py:31-39) of the tour in LDS on matrix M: at most max_it sweeps; returns the sweeps done
// (the final non-improving sweep counts, as in the reference's loop).
// Lists.  SYM (M equals its transpose, one table set): item m = the sorted list of node t[m], m = 0 .. n (t[n] = t[0]: the
// closing edge's side B), serving side A of edge m (pairs (m+1, pos v)) and side B of edge m-1 (pairs (pos u, m-1)) in one
// walk up to the longer of the two ranks.  General: item m = edge m, its side A entries (from nb[t[m]]) followed by its side
// B entries (from nbT[t[m+1]]).
template <bool SYM, int NT, int MAXIPT, int NLS_G>
__device__ inline int nls_search(const NlsLds &L, const NlsMatrix &M, const int n, const long max_it, const int sh,
                                 unsigned long long &walked, unsigned long long *prof, const bool owner_bits) {
  constexpr int NW = NT / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t un = (uint32_t)n;
  int2 *rec = L.rec;
  uint16_t *t = L.t, *pos = L.pos;
  // the two list lengths of the edges, interleaved: rAB[2 m] = side A of edge m, rAB[2 m + 3] = side B of edge m -- so that the word
  // at 2 m holds what item m of a symmetric search walks: side A of edge m | side B of edge m - 1 << 16
  uint16_t *rAB = L.rA;
#define NLS_RA(m) rAB[2 * (m)]
#define NLS_RB(m) rAB[2 * (m) + 3]
  auto refresh_edge = [&](int m) {
    const int x = t[m], y = t[m + 1];
    const uint32_t xy = nls_idx((uint32_t)x, un, (uint32_t)y), yx = nls_idx((uint32_t)y, un, (uint32_t)x);
    const float e = nls_ld(M.d, xy);                          // three independent loads, one trip to memory
    const uint16_t ra = nls_ld(M.rk, xy), rb = nls_ld(M.rkT, yx);
    __builtin_amdgcn_sched_barrier(0);
    rec[m] = make_int2(x | (y << 16), __float_as_int(e));
    NLS_RA(m) = ra;
    NLS_RB(m) = rb;
  };
  // prof (DACO_NLS_PROFILE=1, a debugging aid): shader-clock cycles per phase as thread 0 sees them, summed over the launch
  unsigned long long tmark = prof ? clock64() : 0;
  auto lap = [&](int slot) {
    if (prof && tid == 0) { const unsigned long long now = clock64(); L.lap[slot] += now - tmark; tmark = clock64(); }
  };
  for (int m = tid; m < n; m += NT) refresh_edge(m);
  __syncthreads();
  lap(0);
  const int items = SYM ? n + 1 : n, ipt = (items + NT - 1) / NT;
  auto count_a = [&](int m) -> uint32_t { return m <= n - 3 ? NLS_RA(m) : 0; };
  auto count_of = [&](int item) -> uint32_t {
    if (SYM) {
      const uint32_t pk = *reinterpret_cast<const uint32_t *>(&rAB[2 * item]);      // side A of edge item | side B of edge item - 1
      const uint32_t ca = item <= n - 3 ? pk & 0xffffu : 0, cb = item >= 3 ? pk >> 16 : 0;
      return ca > cb ? ca : cb;
    }
    return count_a(item) + (item >= 2 ? NLS_RB(item) : 0);
  };
  int dlo = 0, dhi = n;                                       // positions whose lists are dirty: everything at first
  int it = 0;
  while (it < max_it) {
    ++it;
    // ---- which lists to re-walk; prefix sum of their lengths (thread tid owns items tid*ipt ..)
    const int blo = dlo >> sh, bhi = (dhi < n ? dhi : n - 1) >> sh;
    const uint64_t span = (bhi - blo >= 63) ? ~(uint64_t)0 : ((((uint64_t)1 << (bhi - blo + 1)) - 1) << blo);
    const int i0 = tid * ipt;
    uint32_t cq[MAXIPT];
    uint32_t local = 0;                                       // entries of my dirty lists | (non-empty dirty lists) << 22
    if (tid < (int)NLS_OB_WORDS) L.obits[tid] = 0;            // (last read right after the previous sweep's compaction barrier)
#pragma unroll
    for (int q = 0; q < MAXIPT; ++q) {
      cq[q] = 0;
      const int item = i0 + q;
      if (q < ipt && item < items) {
        const bool dirty = (item >= dlo && item <= dhi) || (L.sig[item] & span) != 0;
        if (dirty) { cq[q] = count_of(item); L.ckey[item] = NLS_NONE; L.sig[item] = 0; }
        local += cq[q] + (cq[q] ? 1u << 22 : 0u);             // (n <= 1024: fewer than 2^22 entries, fewer than 2^10 lists)
      }
    }
    const uint32_t inc = nls_wave_scan_u32(local);            // inclusive, on the DPP network (no LDS traffic)
    if (lane == 63) L.wsum[wave] = inc;
    __syncthreads();
    lap(1);
    uint32_t base = inc - local;
    for (int w = 0; w < wave; ++w) base += L.wsum[w];
    // the non-empty dirty lists, compacted: list j = item ditem[j], entries pre[j] .. pre[j+1]
    uint32_t at = base & 0x3fffff, j = base >> 22;
#pragma unroll
    for (int q = 0; q < MAXIPT; ++q) {
      if (cq[q]) {
        L.pre[j] = at; L.ditem[j] = (uint16_t)(i0 + q);
        if (at < NLS_OB_MAX) atomicOr(&L.obits[at >> 5], 1u << (at & 31));
        at += cq[q]; ++j;
      }
    }
    if (tid == NT - 1) { L.pre[j] = at; L.wsum[NW] = j; }
    __syncthreads();
    const uint32_t D = L.wsum[NW], W = L.pre[D];
    walked += W;
    lap(2);
    {
      const uint32_t cnt = W;
      // A thread takes NLS_G entries per round and keeps their memory accesses in flight together: first every table entry,
      // then every matrix gather (a sweep near a local optimum walks a few hundred entries: one round, two memory round
      // trips; entry by entry it was a chain of two dependent trips per entry, which is what a sweep's time was made of).
      // Both sides evaluate ((table + gathered) - c) - e: f32 addition commutes, so one expression serves the two.
      // (A thread's NLS_G entries are neighbours in the flat order: one binary search over the compacted lists finds the first
      // one's list, each following entry is in the same list or the next -- every compacted list has at least one entry.)
      const int steps = D > 1 ? 32 - __builtin_clz(D - 1) : 0;            // ceil(log2 D), uniform
      // Round 6: the list of a thread's first entry from the bitmap of list starts -- lane l keeps the number of
      // starts before it; entry w belongs to list (starts up to and including w) - 1: one cross-lane read, one LDS read and a population
      // count instead of ceil(log2 D) dependent LDS reads (eight at the ~150 dirty lists of a steady-state sweep, per round).
      const bool by_bits = owner_bits && cnt <= NLS_OB_MAX;               // uniform
      uint32_t obp = 0;                                                   // lane l: list starts before word l
      if (by_bits) { const uint32_t pc = (uint32_t)__popc(L.obits[lane]); obp = nls_wave_scan_u32(pc) - pc; }
      // (the trip count is the wavefront's: every lane of a wavefront that has an entry in this round takes part in the
      // cross-lane reads; a lane past the end evaluates the last entry's list's first entry and drops it)
      for (uint32_t r0 = (uint32_t)(__builtin_amdgcn_readfirstlane(wave) * 64 * NLS_G); r0 < cnt; r0 += NT * NLS_G) {
        if (prof && tid == 0) L.lap[7] += 1;
        const uint32_t w0r = r0 + (uint32_t)lane * NLS_G;
        const bool ok0 = w0r < cnt;
        const uint32_t w0 = ok0 ? w0r : cnt - 1;
        uint32_t mm[NLS_G], lo[NLS_G], ga[NLS_G], ww[NLS_G];
        bool ok[NLS_G];
        {
          uint32_t a = 0;
          if (by_bits) {
            const int wi = (int)(w0 >> 5);
            const uint32_t bw = L.obits[wi], bp = (uint32_t)__shfl((int)obp, wi);      // (independent: one LDS round trip)
            a = bp + (uint32_t)__popc(bw & ((2u << (w0 & 31)) - 1u)) - 1u;
          } else {
            uint32_t b = D;                                   // pre[a] <= w0 < pre[b]
            for (int st = 0; st < steps; ++st) {
              const uint32_t mid = (a + b) >> 1;
              const bool ge = L.pre[mid] <= w0;
              a = ge ? mid : a;
              b = ge ? b : mid;
            }
          }
          uint32_t start = L.pre[a], next = L.pre[a + 1];
#pragma unroll
          for (int j = 0; j < NLS_G; ++j) {
            const uint32_t w = w0 + j;
            ok[j] = ok0 && w < cnt;
            if (j > 0 && ok[j] && w >= next) { ++a; start = next; next = L.pre[a + 1]; }
            lo[j] = L.ditem[a];
            ww[j] = ok[j] ? w - start : 0;                    // (entries past the end read their list's first entry and drop it)
          }
        }
        float g[NLS_G];
        NbrEntry en[NLS_G];
#pragma unroll
        for (int j = 0; j < NLS_G; ++j) {
          const uint32_t m = lo[j], k = ww[j];
          mm[j] = m;
          if (SYM) {
            en[j] = nls_ld(M.nb, nls_idx((uint32_t)t[m], un, k));
            // whether entry k is inside the item's two lists (side A of edge m, side B of edge m - 1) is known from m and k alone:
            // the two lengths are one LDS word, read here while the table entry is on its way, and what stays is the pair of
            // 16-bit differences length - k (v_pk_sub_i16; ranks and k are below 2^11).  (A rank read under the side's branch was
            // a third dependent LDS round trip per entry, and the branches kept the entries' round trips from overlapping.)
            const nls_s2 lens = __builtin_bit_cast(nls_s2, *reinterpret_cast<const uint32_t *>(&rAB[2 * m]));
            const nls_s2 kk = {(short)k, (short)k};
            lo[j] = __builtin_bit_cast(uint32_t, (nls_s2)(lens - kk));
          } else {
            const int2 r1 = rec[m];
            const uint32_t ca = m + 3 <= un ? NLS_RA(m) : 0;
            const bool side_a = k < ca;
            // (one multiply: the row is selected first, then row * n + entry)
            en[j] = nls_ld(side_a ? M.nb : M.nbT, nls_idx(side_a ? (uint32_t)(r1.x & 0xffff) : (uint32_t)r1.x >> 16, un, side_a ? k : k - ca));
            lo[j] = side_a;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NLS_G; ++j) {
          const uint32_t m = mm[j], pw = pos[en[j].id];
          if (ok[j]) atomicOr((unsigned long long *)&L.sig[m], (unsigned long long)1 << (pw >> sh));
          if (SYM) {
            const nls_s2 room = __builtin_bit_cast(nls_s2, lo[j]);      // (list length - k) of side A, of side B
            const bool side_a = pw > m + 1, side_b = pw >= 1 && pw + 1 < m;
            // side A of edge m: i = m + 1, j = pw: a = table, b = d[t[i]][t[j+1]] gathered
            // side B of edge m - 1: i = pw, j = m - 1: b = table (by symmetry), a = d[t[i-1]][t[j]] gathered
            const bool take = side_a ? (m + 3 <= un && room.x > 0) : (side_b && room.y > 0);
            ok[j] = ok[j] && take;
            const uint32_t e1 = side_a ? m : (side_b ? pw - 1 : 0), e2 = side_a ? pw : (side_b ? m - 1 : 0);
            const uint32_t x1 = (uint32_t)rec[e1].x, x2 = (uint32_t)rec[e2].x;
            ga[j] = nls_idx(side_a ? x1 >> 16 : (x1 & 0xffff), un, side_a ? x2 >> 16 : (x2 & 0xffff));
            lo[j] = ((e1 + 1) << 16) | e2;                    // = (i << 16) | j of the pair: (m + 1, pw) on side A, (pw, m - 1) on side B
          } else {
            const bool side_a = lo[j] != 0;
            // side A: i = m + 1, j = pos[v]: a = d[t[m]][v] from the table, b = d[t[i]][t[j+1]] gathered
            // side B: j = m, i = pos[u]: b = d[u][t[m+1]] from the transposed table, a = d[t[i-1]][t[j]] gathered
            ok[j] = ok[j] && (side_a ? pw > m + 1 : (pw >= 1 && pw < m));
            const uint32_t e1 = side_a ? m : (pw >= 1 ? pw - 1 : 0), e2 = side_a ? pw : m;
            const uint32_t x1 = (uint32_t)rec[e1].x, x2 = (uint32_t)rec[e2].x;
            ga[j] = nls_idx(side_a ? x1 >> 16 : (x1 & 0xffff), un, side_a ? x2 >> 16 : (x2 & 0xffff));
            lo[j] = ((e1 + 1) << 16) | e2;                    // = (i << 16) | j: (m + 1, pw) on side A, (pw, m) on side B
          }
        }
        // (vmcnt counts in order: a gather issued between two table entries' uses would have to return before the next one
        // could be issued -- keep the phases apart)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NLS_G; ++j) g[j] = ok[j] ? nls_ld(M.d, ga[j]) : 0.0f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NLS_G; ++j) {
          // c = e[i - 1], e = e[j]: the two edge lengths are read when the gather is back (the key's low word names the edges), so
          // that a thread holds four words per entry across the second trip to memory
          const float cj = __int_as_float(rec[(lo[j] >> 16) - 1].y), ejj = __int_as_float(rec[lo[j] & 0xffff].y);
          const uint32_t oc = nls_ord_f32(((en[j].d + g[j]) - cj) - ejj);
          if (ok[j] && oc < 0x80000000u) atomicMin((unsigned long long *)&L.ckey[mm[j]], ((uint64_t)oc << 32) | lo[j]);
        }
      }
    }
    lap(3);
    __syncthreads();
    lap(4);
    // ---- minimum over the lists' keys (cached and fresh alike)
    uint64_t best = NLS_NONE;
#pragma unroll
    for (int q = 0; q < MAXIPT; ++q) {
      const int item = i0 + q;
      if (q < ipt && item < items) { const uint64_t c = L.ckey[item]; best = c < best ? c : best; }
    }
    best = nls_wave_min(best);
    if (lane == 0) L.red[wave] = best;
    __syncthreads();
    uint64_t g = L.red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) g = L.red[w] < g ? L.red[w] : g;
    lap(5);
    const float delta = g == NLS_NONE ? 0.0f : nls_unord_f32((uint32_t)(g >> 32));
    if (!((double)delta < -1e-6)) break;                      // `if delta < -1e-6` (two_opt.py:25); uniform
    const int p = (int)((g >> 16) & 0xffff), q = (int)(g & 0xffff);
    const int len = q - p + 1;
    for (int k = tid; k < (len >> 1); k += NT) { const uint16_t u = t[p + k], v = t[q - k]; t[p + k] = v; t[q - k] = u; }
    __syncthreads();
    for (int k = tid; k < len; k += NT) pos[t[p + k]] = (uint16_t)(p + k);
    for (int m = p - 1 + tid; m <= q; m += NT) refresh_edge(m);   // (q <= n-1: edge n-1 ends at t[n] = t[0], unchanged as p >= 1)
    dlo = p - 1; dhi = q + 1;
    __syncthreads();
    lap(6);
  }
  return it;
#undef NLS_RA
#undef NLS_RB
}

// NT threads per tour.  tours [B][T][n] u16 in/out; sweeps_out / costs_out [B][T] or null; counters: [0] sweeps, [1] list
// entries walked, summed over the launch (or null).
// MAXIPT: lists per thread the prefix phase is unrolled for (ceil((n + 1) / NT) must not exceed it)
// NLS_G: list entries a thread evaluates together (their loads in flight at once)
template <int NT, int MAXIPT, int NLS_G>
__global__ void __launch_bounds__(NT, (NT == 192 || (NT == 256 && MAXIPT == 2)) ? 6 : 4)     // (second argument: wavefronts per SIMD -- eight 192-thread or six 256-thread tours of n <= 512 per CU are six, i.e. at most 80 registers)
nls_kernel(int n, int T, const float *dist, long dist_bs, const unsigned char *tabs, const unsigned char *tabsT,
           const float *hdist, long hdist_bs, const unsigned char *htabs, const unsigned char *htabsT, size_t tab_stride,
           uint16_t *tours, long maxt, int T_nls, long T_p, int32_t *sweeps_out, float *costs_out,
           unsigned long long *counters, unsigned long long *prof, int owner_bits) {
  const int blk = xcd_remap(blockIdx.x, gridDim.x);           // an XCD walks consecutive tours: few instances in its L2 at a time
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int np2 = (n + 3) & ~1;                               // >= n + 2, even (the interleaved list lengths reach 2 n + 3)
  NlsLds L;
  L.rec = reinterpret_cast<int2 *>(smem);
  L.t = reinterpret_cast<uint16_t *>(L.rec + np2);
  L.pos = L.t + np2;
  L.rA = L.pos + np2;
  L.rB = L.rA + np2;
  L.ckey = reinterpret_cast<uint64_t *>(L.rB + np2);
  L.sig = L.ckey + np2;
  L.red = L.sig + np2;
  L.lap = reinterpret_cast<unsigned long long *>(L.red + 16);
  L.pre = reinterpret_cast<uint32_t *>(L.lap + 8);
  L.wsum = L.pre + np2 + 2;
  L.ditem = reinterpret_cast<uint16_t *>(L.wsum + 32);
  L.scal = reinterpret_cast<float *>(L.ditem + np2);
  L.obits = reinterpret_cast<uint32_t *>(L.scal + 4);
  const int tid = threadIdx.x, lane = tid & 63;
  const int b = blk / T;
  uint16_t *tour = tours + (size_t)blk * n;
  const bool dsym = tabs == tabsT, hsym = htabs == htabsT;
  const float *dmat = dist + (size_t)b * dist_bs;
  int sh = 0;
  while (((n - 1) >> sh) > 63) ++sh;

  for (int k = tid; k < n; k += NT) { const uint16_t v = tour[k]; L.t[k] = v; L.pos[v] = (uint16_t)k; }
  __syncthreads();
  if (tid == 0) L.t[n] = L.t[0];
  __syncthreads();
  unsigned long long walked = 0;
  long sweeps = 0;
  if (prof && tid == 0) for (int k = 0; k < 8; ++k) L.lap[k] = 0;
  float best_cost = 0.0f;
  bool stale = true;                                          // the tour in memory is not the best one yet
  float *ev = reinterpret_cast<float *>(L.ckey);              // (the lists' keys are dead between two searches)
  // passes: 0 = first 2-opt; then per round an odd pass (perturbation on hdist) and an even one (repair on dist)
  for (int ps = 0; ps <= 2 * T_nls; ++ps) {
    const bool pert = ps & 1;
    if (pert && stale) { for (int k = tid; k < n; k += NT) tour[k] = L.t[k]; stale = false; }
    const unsigned char *tb = (pert ? htabs : tabs) + (size_t)b * tab_stride, *tbT = (pert ? htabsT : tabsT) + (size_t)b * tab_stride;
    NlsMatrix M;
    M.d = pert ? hdist + (size_t)b * hdist_bs : dmat;
    M.nb = nbr_nb(tb); M.nbT = nbr_nb(tbT);
    M.rk = nbr_rk(tb, n); M.rkT = nbr_rk(tbT, n);
    const long cap = pert ? T_p : maxt;
    if (pert ? hsym : dsym) sweeps += nls_search<true, NT, MAXIPT, NLS_G>(L, M, n, cap, sh, walked, prof, owner_bits != 0);
    else sweeps += nls_search<false, NT, MAXIPT, NLS_G>(L, M, n, cap, sh, walked, prof, owner_bits != 0);
    if (pert || (T_nls == 0 && !costs_out)) continue;
    // tour length in daco_tour_costs' order: edges (t[k-1], t[k]) read as d[t[k]][t[k-1]], k = 1 .. n-1, closing edge last
    for (int k = tid; k < n; k += NT) {
      const int u = L.t[k + 1], v = L.t[k];                   // edge k = (t[k], t[k+1]); k = n-1 closes the tour
      ev[k] = dsym ? __int_as_float(L.rec[k].y) : dmat[(size_t)u * n + v];
    }
    __syncthreads();
    if (tid < 64) {
      float sum = 0.0f;
      for (int k0 = 0; k0 < n; k0 += 64) {
        const float e = k0 + lane < n ? ev[k0 + lane] : 0.0f;
        const int cnt = n - k0 < 64 ? n - k0 : 64;
        for (int j = 0; j < cnt; ++j) sum = sum + readlane_f(e, j);
      }
      if (lane == 0) L.scal[0] = sum;
    }
    __syncthreads();
    const float c = L.scal[0];
    if (ps == 0 || c < best_cost) { best_cost = c; stale = true; }   // `improved = new_costs < best_costs` (engine.nls_)
  }
  if (stale) for (int k = tid; k < n; k += NT) tour[k] = L.t[k];
  if (tid == 0) {
    if (sweeps_out) sweeps_out[blk] = (int32_t)sweeps;
    if (costs_out) costs_out[blk] = best_cost;
    if (counters) { atomicAdd(counters, (unsigned long long)sweeps); atomicAdd(counters + 1, walked); }
    if (prof) for (int k = 0; k < 8; ++k) atomicAdd(prof + k, L.lap[k]);
  }
}

}  // namespace daco

using namespace daco;

extern "C" int daco_tsp_nls(void *stream, int B, int T, int n, const float *dist, long dist_bstride, const void *tables,
                            const void *tables_T, const float *hdist, long hdist_bstride, const void *htables,
                            const void *htables_T, uint16_t *tours, long max_iterations, int T_nls, long T_p,
                            int32_t *sweeps, float *costs, unsigned long long *counters) {
  if (B <= 0 || T <= 0 || n < 4 || !dist || !tours || !tables || !tables_T || max_iterations < 0 || T_nls < 0 || T_p < 0 ||
      (T_nls > 0 && (!hdist || !htables || !htables_T))) {
    set_error("daco_tsp_nls: bad argument (B=%d T=%d n=%d T_nls=%d)", B, T, n, T_nls);
    return DACO_E_BADARG;
  }
  if (n > 1024) { set_error("daco_tsp_nls: n=%d above 1024", n); return DACO_E_TOOLARGE; }
  if (max_iterations > 0x3fffffff) max_iterations = 0x3fffffff;
  if (T_p > 0x3fffffff) T_p = 0x3fffffff;
  const int np2 = (n + 3) & ~1;                               // >= n + 2, even (the interleaved list lengths reach 2 n + 3)
  // threads per tour: 256 when there are tours to fill the device several times over (a CU then interleaves several of
  // them), more when there are fewer tours than the device holds (a sweep is a latency chain: more threads shorten it)
  // (the sweeps are bound by instruction issue once the device is full: three wavefronts per tour walk a sweep's ~850 entries in
  // two rounds like four do, with a quarter less of the per-wave work -- scans, reductions, the dirty tests' set-up; measured on
  // config 3: 128 / 192 / 256 / 320 / 384 threads -> 48.1 / 42.1 / 45.8 / 70.0 / 76.8 ms)
  // Round 6 (tools/sweep_nls_threads.py, profiles/r06_sweep_nls_threads.txt): as many threads as a sweep has lists to walk -- about
  // n -- while the launch's threads fit the device (256 CUs x 1024), else fewer: 48 tours of TSP-200 / 500 / 1000 -> 256 / 512 /
  // 1024 threads (1.70 / 4.33 / 16.0 ms; the old rule's 1024 everywhere: 2.21 / 4.56 / 16.0), 400 tours of TSP-500 -> 512 (5.45 ms
  // against 8.95 with 1024), 600 tours of TSP-100 -> 256 (1.45 against 1.55 with 512), 1 024 tours of TSP-500 -> 256 (6.98 against
  // 10.2 with 512); from a few thousand tours on 192 / 256 as before.
  const long ntours = (long)B * T, cap = 262144 / ntours;
  int nt = n <= 256 ? 256 : (n <= 512 ? 512 : 1024);
  while (nt > 256 && nt > cap) nt >>= 1;
  if (nt == 256 && cap < 128 && n + 1 <= 576) nt = 192;
  if (const char *ev = getenv("DACO_NLS_THREADS")) nt = atoi(ev);
  const size_t lds = (size_t)np2 * 8 + (size_t)np2 * 2 * 4 + (size_t)np2 * 8 * 2 + 24 * 8 + (size_t)(np2 + 2) * 4 + 32 * 4 + (size_t)np2 * 2 + 16 + NLS_OB_WORDS * 4;
  // threads per tour: 256 when there are tours to fill the device several times over (a CU then interleaves six of them),
  // more when there are fewer tours than the device holds (a sweep is a latency chain: more threads shorten its evaluation)
  hipStream_t s = (hipStream_t)stream;
  unsigned long long *prof = nullptr;
  if (getenv("DACO_NLS_PROFILE")) {                           // debugging aid: synchronises and prints
    if (hipMalloc((void **)&prof, 8 * sizeof(unsigned long long)) != hipSuccess) prof = nullptr;
    else (void)hipMemsetAsync(prof, 0, 8 * sizeof(unsigned long long), s);
  }
#define DACO_NLS_LAUNCH(NT_, IPT_, G_)                                                                                        \
  hipLaunchKernelGGL((nls_kernel<NT_, IPT_, G_>)    , dim3((unsigned)B * T), dim3(NT_), lds, s, n, T, dist, dist_bstride,     \
                     (const unsigned char *)tables, (const unsigned char *)tables_T, hdist, hdist_bstride,                     \
                     (const unsigned char *)htables, (const unsigned char *)htables_T, nbr_instance_bytes(n), tours,          \
                     max_iterations, T_nls, T_p, sweeps, costs, counters, prof, owner_bits)
  int owner_bits = 1;                                       // (DACO_NLS_OWNER_BITS=0: the binary search over the compacted lists, round 3's form)
  if (const char *ev = getenv("DACO_NLS_OWNER_BITS")) owner_bits = atoi(ev) != 0;
  // entries per thread and round (round 3, 256 threads: 1 / 2 / 3 / 4 -> 58.9 / 48.4 / 45.8 / 50.8 ms on config 3, four entries needing
  // 96 registers; round 6: the two edge lengths of an entry are read after its gather, so four entries fit the 80 registers of six
  // wavefronts per SIMD).  Four when the launch fills the device (the sweeps are issue-bound: fewer rounds), three when it does
  // not (a sweep is a latency chain: more lanes per round).  profiles/r06_nls_owner_bits.txt: config 3 42.8 -> 42.2 ms,
  // 1 024 tours of TSP-1000 53.5 -> 47.4 (from two), 600 tours of TSP-100 1.43 -> 1.57 (hence three there).
  int group = ntours * nt >= 262144 ? 4 : 3;
  if (const char *ev = getenv("DACO_NLS_GROUP")) group = atoi(ev);
  if (nt == 64 && n + 1 <= 128) DACO_NLS_LAUNCH(64, 2, 2);            // one wavefront per tour: the barriers of a sweep cost nothing
  else if (nt == 128 && n + 1 <= 256) DACO_NLS_LAUNCH(128, 2, 2);
  else if (nt >= 1024) DACO_NLS_LAUNCH(1024, 2, 2);
  else if (nt >= 512) DACO_NLS_LAUNCH(512, 3, 2);
  else if (nt == 192 && n + 1 <= 576 && group >= 4) DACO_NLS_LAUNCH(192, 3, 4);
  else if (nt == 192 && n + 1 <= 576) DACO_NLS_LAUNCH(192, 3, 3);
  else if (n + 1 > 512 && group >= 4) DACO_NLS_LAUNCH(256, 5, 4);
  else if (n + 1 > 512) DACO_NLS_LAUNCH(256, 5, 2);
  else if (group <= 1) DACO_NLS_LAUNCH(256, 2, 1);
  else if (group >= 4) DACO_NLS_LAUNCH(256, 2, 4);
  else if (group == 2) DACO_NLS_LAUNCH(256, 2, 2);
  else DACO_NLS_LAUNCH(256, 2, 3);
#undef DACO_NLS_LAUNCH
  hipError_t e = hipGetLastError();
  if (prof) {
    unsigned long long h[8] = {0};
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(prof);
    const double tot = (double)(h[0] + h[1] + h[2] + h[3] + h[4] + h[5] + h[6]);
    fprintf(stderr, "[nls profile] tours %d, share of thread 0's cycles: pass set-up %.3f | dirty+scan %.3f expand %.3f evaluate %.3f "
            "wait %.3f reduce %.3f apply %.3f | cycles per tour %.0f, evaluation rounds per tour %.1f\n", B * T, h[0] / tot, h[1] / tot,
            h[2] / tot, h[3] / tot, h[4] / tot, h[5] / tot, h[6] / tot, tot / (B * T), (double)h[7] / (B * T));
  }
  if (e != hipSuccess) { set_error("nls_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}
