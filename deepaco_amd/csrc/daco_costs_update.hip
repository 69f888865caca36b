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
#include "daco_head_rows.h"
#include <cstdlib>
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

// The same sums, one wavefront per tour: the thread-per-tour kernel above is a chain of `len` dependent-latency gathers
// (152 us for 240 tours of 500 nodes, 113 us for 16 384); here 64 edges are gathered at once and added in the same order
// (lane values read back one by one), so a tour costs len/64 memory latencies instead of len/8.
__global__ void __launch_bounds__(256)
tour_costs_wave_kernel(int B, int n, int len, int A, const float *dist, long dist_bs, const int64_t *paths, int closed,
                       float *costs) {
  const int lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (idx >= B * A) return;
  const int b = idx / A, a = idx - b * A;
  const int64_t *p = paths + (size_t)b * len * A + a;
  const float *d = dist + b * dist_bs;
  const int first = (int)p[0];
  int carry = first;                                      // node before this chunk's first edge
  float s = 0.0f;
  for (int k0 = 1; k0 < len; k0 += 64) {
    const int k = k0 + lane, cnt = min(64, len - k0);
    const int u = k < len ? (int)p[(size_t)k * A] : 0;
    int before = __shfl_up(u, 1, 64);
    if (lane == 0) before = carry;
    // closed tours: d[u_k][u_{k-1}] (the fused sampler's orientation); open routes: d[u_{k-1}][u_k]
    const float e = k < len ? (closed ? d[(size_t)u * n + before] : d[(size_t)before * n + u]) : 0.0f;
    for (int j = 0; j < cnt; ++j) s = s + readlane_f(e, j);
    carry = readlane_i(u, cnt - 1);
  }
  if (closed) s = s + d[(size_t)first * n + carry];
  if (lane == 0) costs[idx] = s;
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

// best-so-far bookkeeping of ACO.run (tsp/aco.py:78-88): first minimum of the costs, and if it
// beats the colony's record, the record and its tour are replaced (device-side: no host branch)
// (tours16 != null: the tours as u16 rows [B][A][ld] -- daco_tsp_sparse_tours_offset -- instead of int64 paths [B][len][A])
__global__ void __launch_bounds__(256)
track_best_kernel(int len, int A, const float *costs, const int64_t *paths, float *lowest, int64_t *shortest,
                  int32_t *best_idx, float *mmas_max, float mmas_scale, const uint16_t *tours16 = nullptr, int ld = 0) {
  __shared__ float rk[4];
  __shared__ int ri[4];
  __shared__ int take;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float bk = __builtin_inff();
  int bi = 0x7fffffff;
  for (int a = tid; a < A; a += 256) {
    const float c = costs[(size_t)b * A + a];
    if (c < bk) { bk = c; bi = a; }
  }
  const KeyIdx r = wave_arg<false>(bk, bi);
  if (lane == 0) { rk[wave] = r.key; ri[wave] = r.idx; }
  __syncthreads();
  if (tid == 0) {
    float k = rk[0];
    int i = ri[0];
    for (int w = 1; w < 4; ++w)
      if (rk[w] < k || (rk[w] == k && ri[w] < i)) { k = rk[w]; i = ri[w]; }
    if (i == 0x7fffffff) i = 0;
    if (best_idx) best_idx[b] = i;
    float low = lowest[b];
    const bool improved = k < low;                      // `if best_cost < self.lowest_cost`
    if (improved) { low = k; lowest[b] = k; }
    if (mmas_max) mmas_max[b] = (1.0f / low) * mmas_scale;   // n / lowest_cost as rtruediv computes it: reciprocal, then * n
    take = improved ? i : -1;
  }
  __syncthreads();
  const int t = take;
  if (t >= 0 && shortest) {
    if (tours16) { for (int k = tid; k < len; k += 256) shortest[(size_t)b * len + k] = (int64_t)tours16[((size_t)b * A + t) * ld + k]; }
    else for (int k = tid; k < len; k += 256) shortest[(size_t)b * len + k] = paths[((size_t)b * len + k) * A + t];
  }
}

// Row owners: a workgroup keeps R rows of tau in LDS.  Row i receives, per ant and in ant order,
// +w at column prev_a(i) and +w at column next_a(i) (tsp/aco.py:95-118: each ant's forward and
// backward edges).  Adds to one element must stay in ant order, so each row is a chain of
// dependent LDS read-modify-writes: two lanes per row (prev side / next side; they never meet on
// a column within one ant) walk the ants in lock step.  The chain is LDS-latency bound, so the
// rest is kept off it: the [node][ant] table arrives in coalesced chunks of DEP_CHUNK ants,
// double-buffered in LDS by all four waves, and the chain reads it four ants at a time.
// (Measured, round 4: the adds as LDS atomics -- ds_add_f32 without return, a lane's adds still execute in the order it issues
// them, results bit-identical -- make the launch 2.5 x slower, 205 us against 82 at the headline shape: the LDS atomic unit
// serialises what the read / add / write chains of 2 x R lanes overlap.  Four ants per LDS round trip -- the four elements
// read together, sums forwarded inside the group -- would also need the PARTNER lane's columns (the prev side of ant a + 1 can
// meet the next side of ant a) and bought 82 -> 74 us without them: the chain is not what the launch waits for.)
constexpr int DEP_CHUNK = 64;

// The head rows of a workgroup's finished rows (HEADS): wavefront w takes the rows w, w + 4, ...  What a row needs from global
// memory -- its row of eta, eta at the head's ids, the ids -- is fetched for up to PF rows at once, so the wavefront waits for
// memory once and then forms the rows out of LDS and registers (one row at a time the fetches were four round trips in a row on
// a kernel that is a chain of latencies already: 64 us on top of the update's 80 at the headline shape).
template <bool RACE, int CH, bool VEC4>
__device__ __forceinline__ void emit_rows_of_wave(int n, int b, int i0, int Rv, const float *rows, const HeadEmit &he, uint32_t *bm,
                                                  int lane, int wave) {
  constexpr int PF = VEC4 ? 2 : 1;     // (unaligned rows: sixteen scalar loads per row, one row at a time)
  const float *eb = he.eta + (size_t)b * he.eta_bs;
  const int ch = he.ch > 0 ? he.ch : -he.ch;
  for (int r0 = wave; r0 < Rv; r0 += 4 * PF) {
    HeadEta<CH> pre[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int rr = r0 + 4 * j;
      if (rr < Rv) {
        const size_t row = (size_t)b * n + i0 + rr;
        head_eta_fetch<CH, VEC4>(pre[j], n, ch, eb + (size_t)(i0 + rr) * n, he.hid + row * (16 * he.spl), he.spl, lane);
      }
    }
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int rr = r0 + 4 * j;
      if (rr < Rv) {
        const size_t row = (size_t)b * n + i0 + rr;
        emit_head_row_pre<RACE, CH, VEC4>(n, ch, rows + rr * n, pre[j], bm, he.hrow + row * sp_head_row_bytes(he.spl), he.P + row * (256 * ch),
                                          he.spl, he.dead, lane);
      }
    }
  }
}

// SYM: symmetric deposit, two lanes per row (prev / next side).  !SYM: directed deposit, one lane
// per row adds at next_a(i) (0xFFFF = ant a does not leave node i); the hub row is skipped (it
// belongs to deposit_hub_kernel).  LDS: rows[R][n] | stage[2][R][DEP_CHUNK] | wts[2][DEP_CHUNK].
// HEADS (symmetric only, round 6): the workgroup still holds its R finished rows of tau in LDS when they have been written back;
// one wavefront per row then forms the NEXT iteration's head row of sampler "scan_sparse" (HEADS = 1) or of the race on head rows
// (HEADS = 2) from them and the row of eta (daco_head_rows.h emit_head_row: the very code of sparse_prepass_kernel), so that
// iteration neither launches the pre-pass nor reads tau again.
// GT (symmetric only): the table in the grouped layout [B][ceil(A/8)][n][8] (daco_tsp_sample_heads(nbr_grouped = 1)): the
// entries of a workgroup's R rows and eight ants are one run of R x 32 bytes.
template <bool SYM, int HEADS, bool GT = false>
__global__ void __launch_bounds__(256, 4)
deposit_rows_kernel(int n, int A, int R, int hub, float *tau, const uint32_t *nbr, const float *costs, const float *weights,
                    float decay, const int *best, const float *clamp_min, const float *clamp_max, float floor_val, const HeadEmit he) {
  extern __shared__ __attribute__((aligned(16))) float rows[];
  uint32_t *stage = (uint32_t *)(rows + (((size_t)R * n + 3) & ~(size_t)3));
  float *wts = (float *)(stage + 2 * R * DEP_CHUNK);
  const int bpi = (n + R - 1) / R;
  const int b = blockIdx.x / bpi;
  const int i0 = (blockIdx.x - b * bpi) * R;
  const int Rv = min(R, n - i0);
  float *g = tau + ((size_t)b * n + i0) * n;
  const int cnt = Rv * n;
  const int hub_lo = SYM ? -1 : (hub - i0) * n, hub_hi = SYM ? -1 : hub_lo + n;
  int alo = 0, ahi = A;
  if (best) { alo = best[b]; ahi = alo + 1; }
  const uint32_t *tab = GT ? nbr + (size_t)b * ((A + 7) >> 3) * n * 8 + (size_t)i0 * 8     // group g, row r, ant a8 at ((g n) + r) 8 + a8
                           : nbr + ((size_t)b * n + i0) * A;        // rows i0.. of this instance's [n][A] table
  const float *cs = costs + (size_t)b * A, *wt = weights ? weights + (size_t)b * A : nullptr;
  // chunk loader: consecutive threads on consecutive ants of one row (256-byte segments).  Wave 0 runs
  // the chains, so after the first chunk only waves 1..3 fetch: a chain never waits for a prefetch.
  // (Round 6, last session: a loader thread's entries of a chunk are all in flight before the first is waited for.  The loop used
  // to be load -> s_waitcnt vmcnt(0) -> LDS store per entry -- five or six dependent memory round trips per chunk for each of the
  // 192 loader threads, more than the 2.9 us the chain spends on a chunk: the chain wave stood at the chunk barrier waiting for
  // its loaders, and the launch was bound by their latency, not by the chain.)
  constexpr int LDU = 6;                                     // entries per loader thread in flight (6 x 192 >= 16 rows x 64 ants)
  auto load_chunk = [&](int c0, int buf, int t0, int nt) {
    const int m = min(DEP_CHUNK, ahi - c0);
    if constexpr (GT) {
      // consecutive threads on the eight ants of a row, then on the rows, then on the chunk's groups: runs of Rv x 32 bytes
      const int per_g = Rv * 8, total = per_g * (DEP_CHUNK / 8);
      for (int ib = t0; ib < total; ib += nt * LDU) {
        uint32_t v[LDU];
        int dstx[LDU];
#pragma unroll
        for (int u = 0; u < LDU; ++u) {
          const int i = ib + u * nt, ic = i < total ? i : t0;
          const int gq = ic / per_g, rem = ic - gq * per_g;
          const int r = rem >> 3, a8 = rem & 7;
          const int ant = c0 + gq * 8 + a8;                 // (c0: a multiple of 8 except for the elitist's single ant)
          const int j = ant - c0;
          const int antc = j < m ? ant : c0;                  // (an ant past the chunk's end: reads the first one's entry, stores ~0)
          v[u] = tab[((size_t)(antc >> 3) * n + r) * 8 + (antc & 7)];
          dstx[u] = (i < total && j < DEP_CHUNK) ? ((buf * R + r) * DEP_CHUNK + j) | (j < m ? 0 : (int)0x80000000) : -1;
        }
#pragma unroll
        for (int u = 0; u < LDU; ++u)
          if (dstx[u] != -1) stage[dstx[u] & 0x7fffffff] = dstx[u] < 0 ? 0xFFFFFFFFu : v[u];
      }
    } else {
      const int total = Rv * DEP_CHUNK;
      for (int ib = t0; ib < total; ib += nt * LDU) {
        uint32_t v[LDU];
        int dstx[LDU];
#pragma unroll
        for (int u = 0; u < LDU; ++u) {
          const int i = ib + u * nt, ic = i < total ? i : t0;
          const int r = ic / DEP_CHUNK, j = ic - r * DEP_CHUNK;
          v[u] = tab[(size_t)r * A + c0 + (j < m ? j : 0)];
          dstx[u] = i < total ? ((buf * R + r) * DEP_CHUNK + j) | (j < m ? 0 : (int)0x80000000) : -1;
        }
#pragma unroll
        for (int u = 0; u < LDU; ++u)
          if (dstx[u] != -1) stage[dstx[u] & 0x7fffffff] = dstx[u] < 0 ? 0xFFFFFFFFu : v[u];
      }
    }
    if (t0 < DEP_CHUNK)
      wts[buf * DEP_CHUNK + t0] = t0 < m ? (wt ? wt[c0 + t0] : 1.0f / cs[c0 + t0]) : 0.0f;
  };
  load_chunk(alo, 0, threadIdx.x, 256);
  // the rows themselves: 16-byte vectors when the slab is aligned (n % 4 == 0), else element-wise
  const bool vec_ok = SYM && (n & 3) == 0 && ((uintptr_t)g & 15) == 0;
  if (vec_ok) {
    const float4 *g4 = (const float4 *)g;
    float4 *r4 = (float4 *)rows;
    for (int i = threadIdx.x; i < cnt / 4; i += blockDim.x) {
      float4 x = g4[i];
      x.x *= decay; x.y *= decay; x.z *= decay; x.w *= decay;
      r4[i] = x;
    }
  } else {
    for (int i = threadIdx.x; i < cnt; i += blockDim.x)
      if (SYM || i < hub_lo || i >= hub_hi) rows[i] = g[i] * decay;
  }
  __syncthreads();
  const int r = SYM ? threadIdx.x >> 1 : threadIdx.x;
  const int sh = SYM ? (threadIdx.x & 1) << 4 : 16;                // prev side: low half-word, next side: high
  const bool chain = r < Rv && (SYM || i0 + r != hub);
  float *row = rows + (chain ? r : 0) * n;
  int buf = 0;
  for (int c0 = alo; c0 < ahi; c0 += DEP_CHUNK, buf ^= 1) {
    if (c0 + DEP_CHUNK < ahi && threadIdx.x >= 64)                  // next chunk lands while this one is applied
      load_chunk(c0 + DEP_CHUNK, buf ^ 1, threadIdx.x - 64, 192);
    if (chain) {
      const int m = min(DEP_CHUNK, ahi - c0);
      const uint32_t *st = stage + (buf * R + r) * DEP_CHUNK;
      const float *w = wts + buf * DEP_CHUNK;
      int j = 0;
      // the next four ants' columns and weights are fetched before this group's adds: only the
      // read-modify-writes themselves stay on the dependent chain
      uint4 v = *(const uint4 *)st;
      float4 w4 = *(const float4 *)w;
      for (; j + 4 <= m; j += 4) {
        const int jn = j + 8 <= m ? j + 4 : j;            // (the staging rows hold DEP_CHUNK entries: in bounds)
        const uint4 vn = *(const uint4 *)(st + jn);
        const float4 wn = *(const float4 *)(w + jn);
        const uint32_t c0_ = (v.x >> sh) & 0xFFFFu, c1_ = (v.y >> sh) & 0xFFFFu, c2_ = (v.z >> sh) & 0xFFFFu, c3_ = (v.w >> sh) & 0xFFFFu;
        if (SYM || c0_ != 0xFFFFu) row[c0_] = row[c0_] + w4.x;
        if (SYM || c1_ != 0xFFFFu) row[c1_] = row[c1_] + w4.y;
        if (SYM || c2_ != 0xFFFFu) row[c2_] = row[c2_] + w4.z;
        if (SYM || c3_ != 0xFFFFu) row[c3_] = row[c3_] + w4.w;
        v = vn;
        w4 = wn;
      }
      for (; j < m; ++j) {
        const uint32_t col = (st[j] >> sh) & 0xFFFFu;
        if (SYM || col != 0xFFFFu) row[col] = row[col] + w[j];
      }
    }
    __syncthreads();
  }
  const bool clamp = clamp_max != nullptr;
  const float cmin = clamp ? clamp_min[b] : 0.0f, cmax = clamp ? clamp_max[b] : 0.0f;
  auto finish = [&](float x) {
    if (clamp) { x = x < cmin ? cmin : x; x = x > cmax ? cmax : x; }
    if (floor_val > 0.0f) x = x < floor_val ? floor_val : x;
    return x;
  };
  if (vec_ok) {
    float4 *g4 = (float4 *)g;
    float4 *r4 = (float4 *)rows;
    for (int i = threadIdx.x; i < cnt / 4; i += blockDim.x) {
      float4 x = r4[i];
      x.x = finish(x.x); x.y = finish(x.y); x.z = finish(x.z); x.w = finish(x.w);
      g4[i] = x;
      if constexpr (HEADS != 0) r4[i] = x;                 // (the head rows are formed from the values tau now holds)
    }
  } else {
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
      if (!SYM && i >= hub_lo && i < hub_hi) continue;
      const float x = finish(rows[i]);
      g[i] = x;
      if constexpr (HEADS != 0) rows[i] = x;
    }
  }
  if constexpr (SYM && HEADS != 0) {
    // (the wavefronts' bitmaps live in the staging area, which the chains no longer need: 512 bytes of static LDS on top of the
    // 40 KB slab were the difference between four and three workgroups per CU -- the update took 143 us instead of ~90)
    uint32_t (*head_bm)[32] = reinterpret_cast<uint32_t (*)[32]>(stage);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (16-byte row vectors: the LDS rows are aligned whenever n % 4 == 0; eta's rows when its base and stride are -- he.ch < 0 says no)
    const bool vec4 = he.ch > 0;
    const int ch = vec4 ? he.ch : -he.ch;
    if (ch <= 2) { if (vec4) emit_rows_of_wave<HEADS == 2, 2, true>(n, b, i0, Rv, rows, he, head_bm[wave], lane, wave);
                   else emit_rows_of_wave<HEADS == 2, 2, false>(n, b, i0, Rv, rows, he, head_bm[wave], lane, wave); }
    else { if (vec4) emit_rows_of_wave<HEADS == 2, 4, true>(n, b, i0, Rv, rows, he, head_bm[wave], lane, wave);
           else emit_rows_of_wave<HEADS == 2, 4, false>(n, b, i0, Rv, rows, he, head_bm[wave], lane, wave); }
  }
}

// ------------------------------------------------------------------ directed deposit (CVRP and siblings)
// cvrp/aco.py:107-130: tau[path[:-1], path[1:]] += 1/cost per ant, duplicates of an index pair
// within one ant (the padding edge (0,0)) collapse to ONE add.  Row i >= 1 (a customer) is
// left exactly once per ant -> one lane per row walks the ants in order.  Row 0 (the depot) is
// left once per route: build_next_kernel also collects, per ant, the SET of nodes that follow
// the depot as a bitmap (W words per ant; a set, so the (0,0) padding edge counts once); the depot
// kernel gives every column one thread that walks the ants in order with its sum in a register.
__global__ void __launch_bounds__(256)
build_next_kernel(int B, int n, int len, int A, int hub, int W, const int64_t *paths, uint32_t *nbr, uint32_t *hubmask) {
  const long total = (long)B * (len - 1) * A;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int a = (int)(i % A);
    const long r = i / A;
    const int k = (int)(r % (len - 1)), b = (int)(r / (len - 1));
    const int64_t *p = paths + (size_t)b * len * A + a;
    const uint32_t u = (uint32_t)p[(size_t)k * A], v = (uint32_t)p[(size_t)(k + 1) * A];
    if ((int)u != hub) nbr[((size_t)b * n + u) * A + a] = v << 16;
    else atomicOr(hubmask + ((size_t)b * A + a) * W + (v >> 5), 1u << (v & 31));
  }
}

__device__ inline float ant_weight(const float *weights, const float *costs, size_t idx) {
  return weights ? weights[idx] : 1.0f / costs[idx];
}

// the hub row (depot / dummy node): thread per column, ants in order, the running value stays in a
// register.  The per-ant bitmap words and weights are staged through LDS in chunks of HUB_CHUNK
// ants (coalesced loads, 1/cost computed once per ant), so the ant loop itself touches no memory
// but LDS and has no dependent loads.
constexpr int HUB_CHUNK = 256;

__global__ void __launch_bounds__(256)
deposit_hub_kernel(int n, int A, int W, int hub, float *tau, const uint32_t *hubmask, const int32_t *tab_lens,
                   const float *costs, const float *weights, float decay, const int *best, const float *clamp_min,
                   const float *clamp_max, float floor_val) {
  extern __shared__ __attribute__((aligned(16))) uint32_t hub_lds[];   // [HUB_CHUNK][W] bitmap words | [HUB_CHUNK] weights | [HUB_CHUNK] padded?
  uint32_t *mw = hub_lds;
  float *wl = (float *)(hub_lds + (size_t)HUB_CHUNK * W);
  uint32_t *padl = (uint32_t *)(wl + HUB_CHUNK);
  __shared__ int longest_s;
  const int b = blockIdx.x;
  float *g = tau + ((size_t)b * n + hub) * n;
  int alo = 0, ahi = A;
  if (best) { alo = best[b]; ahi = alo + 1; }
  // a table written by the sampler holds each ant's own route only; the reference pads the shorter routes
  // with the depot up to the longest one of the colony, i.e. one (hub,hub) edge for every shorter ant
  if (threadIdx.x == 0) longest_s = 0;
  __syncthreads();
  if (tab_lens) {
    int mx = 0;
    for (int a = threadIdx.x; a < A; a += blockDim.x) mx = max(mx, tab_lens[(size_t)b * A + a]);
    atomicMax(&longest_s, mx);
  }
  __syncthreads();
  const int longest = longest_s;
  const bool clamp = clamp_max != nullptr;
  const float cmin = clamp ? clamp_min[b] : 0.0f, cmax = clamp ? clamp_max[b] : 0.0f;
  // 256 columns per pass (one pass up to n = 256); the chunks are staged again for every pass
  for (int cb = 0; cb < n; cb += blockDim.x) {
    const int c = cb + threadIdx.x;
    const uint32_t *mc = mw + (c >> 5);
    const uint32_t bit = 1u << (c & 31);
    const bool is_hub = c == hub;
    float v = c < n ? g[c] * decay : 0.0f;
    for (int c0 = alo; c0 < ahi; c0 += HUB_CHUNK) {
      const int m = min(HUB_CHUNK, ahi - c0);
      __syncthreads();
      for (int i = threadIdx.x; i < m * W; i += blockDim.x) mw[i] = hubmask[((size_t)b * A + c0) * W + i];
      for (int j = threadIdx.x; j < m; j += blockDim.x) {
        wl[j] = ant_weight(weights, costs, (size_t)b * A + c0 + j);
        padl[j] = (tab_lens && tab_lens[(size_t)b * A + c0 + j] < longest) ? 1u : 0u;
      }
      __syncthreads();
      if (c < n) {
        int j = 0;
        for (; j + 8 <= m; j += 8) {                      // the eight ants' words and weights are read up front
          uint32_t hit[8];
          float wv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            hit[u] = (mc[(size_t)(j + u) * W] & bit) | (is_hub ? padl[j + u] : 0u);
            wv[u] = wl[j + u];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) v = hit[u] ? v + wv[u] : v;
        }
        for (; j < m; ++j)
          if ((mc[(size_t)j * W] & bit) || (is_hub && padl[j])) v = v + wl[j];
      }
    }
    if (c < n) {
      if (clamp) { v = v < cmin ? cmin : v; v = v > cmax ? cmax : v; }
      if (floor_val > 0.0f) v = v < floor_val ? floor_val : v;
      g[c] = v;
    }
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
  if (len >= 64)
    hipLaunchKernelGGL(tour_costs_wave_kernel, dim3((total + 3) / 4), dim3(256), 0, (hipStream_t)stream, B, n, len, A, dist,
                       dist_bstride, paths, closed, costs);
  else
    hipLaunchKernelGGL(tour_costs_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, n,
                       len, A, dist, dist_bstride, paths, closed, costs);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("tour_costs_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}

extern "C" int daco_track_best(void *stream, int B, int len, int A, const float *costs, const int64_t *paths,
                               float *lowest, int64_t *shortest, int32_t *best_idx, float *mmas_max, float mmas_scale) {
  if (B <= 0 || len <= 0 || A <= 0 || !costs || !lowest || (shortest && !paths)) {
    set_error("daco_track_best: bad argument (B=%d len=%d A=%d)", B, len, A);
    return DACO_E_BADARG;
  }
  hipLaunchKernelGGL(track_best_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, len, A, costs, paths, lowest, shortest,
                     best_idx, mmas_max, mmas_scale);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("track_best_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}

extern "C" int daco_track_best_tours16(void *stream, int B, int len, int A, int ld, const float *costs, const uint16_t *tours16,
                                       float *lowest, int64_t *shortest, int32_t *best_idx, float *mmas_max, float mmas_scale) {
  if (B <= 0 || len <= 0 || A <= 0 || ld < len || !costs || !lowest || !tours16) {
    set_error("daco_track_best_tours16: bad argument (B=%d len=%d A=%d ld=%d)", B, len, A, ld);
    return DACO_E_BADARG;
  }
  hipLaunchKernelGGL(track_best_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, len, A, costs, (const int64_t *)nullptr, lowest, shortest,
                     best_idx, mmas_max, mmas_scale, tours16, ld);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("track_best_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}

extern "C" size_t daco_directed_table_bytes(int B, int n, int A) {
  if (B <= 0 || n <= 0 || A <= 0) return 0;
  // successor per (node, ant) | per-ant set of depot successors | per-ant route length
  return align256((size_t)B * n * A * sizeof(uint32_t)) + align256((size_t)B * A * ((n + 31) / 32) * sizeof(uint32_t)) +
         align256((size_t)B * A * sizeof(int32_t));
}

extern "C" size_t daco_pheromone_update_workspace_bytes(int B, int n, int len, int A) {
  if (B <= 0 || n <= 0 || A <= 0 || len <= 0) return 0;
  // nbr table | best-ant index | (directed) per-ant bitmap of the nodes that follow the hub
  return align256((size_t)B * A * n * sizeof(uint32_t)) + align256((size_t)B * sizeof(int)) +
         align256((size_t)B * A * ((n + 31) / 32) * sizeof(uint32_t));
}

// rows per workgroup from an LDS budget: smaller slabs mean more workgroups per CU moving tau while others run
// their add chains.  Measured (DACO_DEPOSIT_LDS_KB sweep, TSP-500 x 512 x 64 / CVRP-100 x 512 x 256): symmetric
// 80 / 53 / 40 / 32 KiB -> 119 / 91 / 84 / 105 us; directed 80 / 40 / 26 / 20 KiB -> 92 / 70 / 54 / 53 us.
static int rows_per_block(int n, bool symmetric, int B = 1 << 20) {
  static const int override_kb = getenv("DACO_DEPOSIT_LDS_KB") ? atoi(getenv("DACO_DEPOSIT_LDS_KB")) : 0;
  const int budget_kb = override_kb ? override_kb : (symmetric ? 40 : 24);
  int R = (budget_kb * 1024 - 2 * DEP_CHUNK * 4 - 16) / (4 * n + 2 * DEP_CHUNK * 4);
  const int cap = symmetric ? 32 : 64;
  if (R > cap) R = cap;
  if (R < 1) R = 1;
  // few instances (round 6: one colony per instance is the reference's call pattern): the slab size above would leave most CUs
  // without a workgroup, and a workgroup's load / store / head-row phases are serial per wavefront -- fewer rows per workgroup
  // until the launch has two workgroups per CU (the chains are as long either way: one per row)
  while (R > 1 && (long)B * ((n + R - 1) / R) < 512) R = (R + 1) / 2;
  return R;
}
static size_t deposit_lds_bytes(int R, int n) {
  return (((size_t)R * n + 3) & ~(size_t)3) * sizeof(float) + (size_t)2 * R * DEP_CHUNK * sizeof(uint32_t) + 2 * DEP_CHUNK * sizeof(float);
}

static int pheromone_update_impl(void *stream, int B, int n, int len, int A, float *tau,
                                 const int64_t *paths, const float *costs, float decay, int elitist,
                                 int symmetric, const float *clamp_min, const float *clamp_max,
                                 float floor_val, const uint32_t *nbr_in, const float *weights, int hub,
                                 void *workspace, size_t workspace_bytes, const HeadEmit &he) {
  if (B <= 0 || n < 3 || A <= 0 || !tau || (!paths && !nbr_in) || !costs || !workspace) {     // (the table replaces the paths)
    set_error("daco_pheromone_update: bad argument (B=%d n=%d A=%d)", B, n, A);
    return DACO_E_BADARG;
  }
  if ((clamp_min == nullptr) != (clamp_max == nullptr)) { set_error("daco_pheromone_update: clamp_min/clamp_max must both be given"); return DACO_E_BADARG; }
  if (n > DACO_MAX_NODES) { set_error("daco_pheromone_update: n=%d exceeds DACO_MAX_NODES", n); return DACO_E_TOOLARGE; }
  if (symmetric && len != n) { set_error("daco_pheromone_update: symmetric deposit needs len == n"); return DACO_E_BADARG; }
  if (hub >= n) { set_error("daco_pheromone_update: hub %d >= n %d", hub, n); return DACO_E_BADARG; }
  if (!symmetric && len < 2) { set_error("daco_pheromone_update: directed deposit needs len >= 2"); return DACO_E_BADARG; }
  if (!symmetric && nbr_in && hub != 0) { set_error("daco_pheromone_update: a directed next_table implies hub = 0"); return DACO_E_BADARG; }
  const size_t need = daco_pheromone_update_workspace_bytes(B, n, len, A);
  if (workspace_bytes < need) { set_error("daco_pheromone_update: workspace %zu < %zu", workspace_bytes, need); return DACO_E_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  const uint32_t *nbr = nbr_in ? nbr_in : (const uint32_t *)workspace;
  int *best = (int *)((char *)workspace + align256((size_t)B * A * n * sizeof(uint32_t)));
  if (!symmetric) {
    const int W = (n + 31) / 32;
    const uint32_t *hubmask = (const uint32_t *)((char *)best + align256((size_t)B * sizeof(int)));
    const int32_t *tab_lens = nullptr;
    if (nbr_in) {
      hubmask = (const uint32_t *)((const char *)nbr_in + align256((size_t)B * n * A * sizeof(uint32_t)));
      tab_lens = (const int32_t *)((const char *)hubmask + align256((size_t)B * A * W * sizeof(uint32_t)));
    } else {
      if (hipMemsetAsync((void *)hubmask, 0, (size_t)B * A * W * sizeof(uint32_t), s) != hipSuccess) { set_error("hipMemsetAsync failed"); return DACO_E_HIP; }
      if (hipMemsetAsync(workspace, 0xFF, (size_t)B * A * n * sizeof(uint32_t), s) != hipSuccess) { set_error("hipMemsetAsync failed"); return DACO_E_HIP; }
      const long total = (long)B * (len - 1) * A;
      int blocks = (int)((total + 255) / 256);
      if (blocks > 16384) blocks = 16384;
      hipLaunchKernelGGL(build_next_kernel, dim3(blocks), dim3(256), 0, s, B, n, len, A, hub, W, paths, (uint32_t *)workspace, (uint32_t *)hubmask);
    }
    if (elitist) hipLaunchKernelGGL(argmin_cost_kernel, dim3(B), dim3(64), 0, s, A, costs, best);
    const int R = rows_per_block(n, false, B);
    const int bpi = (n + R - 1) / R;
    if (hub >= 0)
      hipLaunchKernelGGL(deposit_hub_kernel, dim3(B), dim3(256), (size_t)HUB_CHUNK * (W + 2) * sizeof(uint32_t), s, n, A, W, hub, tau, hubmask, tab_lens,
                         costs, weights, decay, elitist ? best : nullptr, clamp_min, clamp_max, floor_val);
    hipLaunchKernelGGL((deposit_rows_kernel<false, 0>), dim3(B * bpi), dim3(256), deposit_lds_bytes(R, n), s, n, A, R, hub, tau,
                       nbr, costs, weights, decay, elitist ? best : nullptr, clamp_min, clamp_max,
                       floor_val, he);
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
  const int R = rows_per_block(n, true, B);
  const int bpi = (n + R - 1) / R;
#define DACO_DEPOSIT_SYM_G(H, G) hipLaunchKernelGGL((deposit_rows_kernel<true, H, G>), dim3(B * bpi), dim3(256), deposit_lds_bytes(R, n), s, n, A, R, 0, tau, \
                                                    nbr, costs, weights, decay, elitist ? best : nullptr, clamp_min, clamp_max, floor_val, he)
#define DACO_DEPOSIT_SYM(H) do { if (grouped) DACO_DEPOSIT_SYM_G(H, true); else DACO_DEPOSIT_SYM_G(H, false); } while (0)
  const bool grouped = nbr_in != nullptr && he.nbr_grouped != 0;        // (a table rebuilt here from `paths` is [n][A])
  if (!he.eta) DACO_DEPOSIT_SYM(0);
  else if (!he.race) DACO_DEPOSIT_SYM(1);
  else DACO_DEPOSIT_SYM(2);
#undef DACO_DEPOSIT_SYM
#undef DACO_DEPOSIT_SYM_G
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("pheromone update launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}

extern "C" int daco_pheromone_update(void *stream, int B, int n, int len, int A, float *tau,
                                     const int64_t *paths, const float *costs, float decay, int elitist,
                                     int symmetric, const float *clamp_min, const float *clamp_max,
                                     float floor_val, const uint32_t *nbr_in, const float *weights, int hub,
                                     void *workspace, size_t workspace_bytes) {
  return pheromone_update_impl(stream, B, n, len, A, tau, paths, costs, decay, elitist, symmetric, clamp_min, clamp_max, floor_val, nbr_in,
                               weights, hub, workspace, workspace_bytes, HeadEmit{});
}

extern "C" size_t daco_tsp_sparse_workspace_bytes(int B, int n, int A);

extern "C" int daco_pheromone_update_heads(void *stream, int B, int n, int A, float *tau, const int64_t *paths, const float *costs,
                                           float decay, int elitist, const float *clamp_min, const float *clamp_max, float floor_val,
                                           const uint32_t *nbr_in, const float *weights, void *workspace, size_t workspace_bytes,
                                           const float *eta, long eta_bstride, float alpha, float beta, const uint16_t *head_id,
                                           int head_slots, int race, int nbr_grouped, void *sparse_workspace, size_t sparse_workspace_bytes) {
  if (!eta || !head_id || !sparse_workspace) { set_error("daco_pheromone_update_heads: bad argument"); return DACO_E_BADARG; }
  if (head_slots != 64 && head_slots != 128) { set_error("daco_pheromone_update_heads: head_slots = %d (64 or 128)", head_slots); return DACO_E_BADARG; }
  if (n <= 128 || n > 1024) { set_error("daco_pheromone_update_heads: n=%d outside 129..1024 (the sizes daco_tsp_sample_heads serves)", n); return DACO_E_TOOLARGE; }
  if (alpha != 1.0f || beta != 1.0f) { set_error("daco_pheromone_update_heads: alpha = beta = 1 only (the rows of tau are formed here; other exponents take the sampler's own pass)"); return DACO_E_BADARG; }
  const size_t need = daco_tsp_sparse_workspace_bytes(B, n, A);
  if (sparse_workspace_bytes < need) { set_error("daco_pheromone_update_heads: sparse workspace %zu < %zu bytes", sparse_workspace_bytes, need); return DACO_E_WORKSPACE; }
  HeadEmit he;
  he.eta = eta; he.eta_bs = eta_bstride; he.hid = head_id;
  he.P = (float *)sparse_workspace;                                     // (the sampler's layout: dense rows, then head rows)
  he.hrow = (char *)sparse_workspace + align256((size_t)B * n * (n <= 512 ? 512 : 1024) * sizeof(float));
  he.spl = head_slots / 16; he.race = race ? 1 : 0; he.nbr_grouped = nbr_grouped ? 1 : 0;
  const int ld = n <= 512 ? 512 : 1024;
  he.dead = ld;
  const bool vec4 = (n & 3) == 0 && (eta_bstride & 3) == 0 && (((uintptr_t)eta | (uintptr_t)tau) & 15) == 0;
  he.ch = vec4 ? ld / 256 : -(ld / 256);
  return pheromone_update_impl(stream, B, n, n, A, tau, paths, costs, decay, elitist, 1, clamp_min, clamp_max, floor_val, nbr_in, weights, 0,
                               workspace, workspace_bytes, he);
}
