// daco_head_rows.h -- the HEAD ROW of one transition row (sampler "scan_sparse" / the race on head rows), formed by one
// wavefront from the row of tau (global memory, or the copy the pheromone update holds in LDS) and the row of eta.
//
// Reference behaviour: the row is tau[i]^alpha * eta[i]^beta of tsp/aco.py:165-172, alpha = beta = 1 here (tsp_nls/aco.py:195 forms it once per
// iteration); the head / tail split is this library's (restated in the CPU checker as orc_sparse_head_values).  Two callers
// share this code so that their head rows agree bit for bit: sparse_prepass_kernel (daco_scan_sparse.hip: the first iteration,
// and every caller that updates tau some other way) and deposit_rows_kernel (daco_costs_update.hip: the update that just wrote
// the row emits the NEXT iteration's head row from the registers / LDS it still holds -- tau is not read a second time).
#pragma once
#include "daco_device.h"

namespace daco {

// head slots per row: 64 or 128 (SPL = 4 or 8 per lane, 16 lanes); the last slot holds the tail total (its id field: the live
// count in the caller's table).  Bytes per lane of a head row: SPL f32 values, SPL u16 ids -- 24 or 48.
constexpr int SP_KH_MAX = 128;
__host__ __device__ constexpr int sp_lane_bytes(int spl) { return spl * 6; }
__host__ __device__ constexpr size_t sp_head_row_bytes(int spl) { return (size_t)16 * sp_lane_bytes(spl); }

// element k0..k0+3 of the row tau * eta (0 past n), no fused multiply-add.  The head-row kernels take alpha = beta = 1 -- every
// script of the reference (tsp/aco.py:10-11) -- and nothing else: other exponents are applied to the two matrices BEFORE these
// kernels run (pow_pair_kernel, daco_scan_sparse.hip: tau^alpha and eta^beta into the workspace), the products are the same.
// (Round 6 first formed tau^alpha * eta^beta here with pw(): sixteen powf bodies per vector kept the inliner from inlining this
// function at all, and the rare ways of scan_sparse_kernel and every row of emit_head_row made real calls -- +65 us on the
// headline launch, and the update's head rows cost 64 us instead of ~10; profiles/r06_fused_head_rows.txt.)
// The loads are UNCONDITIONAL (a vector past the row reads the row's first entries and is then replaced by zeros): a load under
// `if (k0 < n)` sits in its own exec-masked block, the first use waits there, and the next chunk's loads are issued only after
// that wait -- two chunks were two memory round trips on the critical path of every row walk (measured: +55 us on the headline
// launch, whose wavefronts stop for a walk some thirty times per tour).
template <bool VEC4>
__device__ __forceinline__ void sp_load4(const float *row, int n, int k0, float4 &out) {
  if constexpr (VEC4) {
    out = *reinterpret_cast<const float4 *>(row + (k0 < n ? k0 : 0));      // (n % 4 == 0: a vector is inside the row or past it)
  } else {
    out.x = row[k0 + 0 < n ? k0 + 0 : 0]; out.y = row[k0 + 1 < n ? k0 + 1 : 0];
    out.z = row[k0 + 2 < n ? k0 + 2 : 0]; out.w = row[k0 + 3 < n ? k0 + 3 : 0];
  }
}
__device__ __forceinline__ float4 sp_mul4_masked(const float4 &t, const float4 &e, int n, int k0) {
  float4 v;
  v.x = k0 + 0 < n ? t.x * e.x : 0.0f; v.y = k0 + 1 < n ? t.y * e.y : 0.0f;
  v.z = k0 + 2 < n ? t.z * e.z : 0.0f; v.w = k0 + 3 < n ? t.w * e.w : 0.0f;
  return v;
}
template <bool VEC4>
__device__ __forceinline__ float4 sp_prob4(const float *tr, const float *er, int n, int k0) {
  float4 t, e;
  sp_load4<VEC4>(tr, n, k0, t);
  sp_load4<VEC4>(er, n, k0, e);
  return sp_mul4_masked(t, e, n, k0);
}

// What a head row needs of eta (and of the caller's head table), fetched from global memory up front: the row itself in up to CH
// chunks, eta at the head's ids, the ids.  Several rows' worth can be in flight before the first is used.
template <int CH>
struct HeadEta {
  float4 e[CH];
  float eh[2];
  int id[2];
  int cnt;
};
template <int CH, bool VEC4>
__device__ __forceinline__ void head_eta_fetch(HeadEta<CH> &h, int n, int ch, const float *er, const uint16_t *ids, int spl, int lane) {
  const int kh = 16 * spl;
  h.cnt = ids[kh - 1] < kh - 1 ? ids[kh - 1] : kh - 1;      // (a malformed table must not reach past the row: count clamped, id >= n = empty)
#pragma unroll
  for (int c = 0; c < CH; ++c) sp_load4<VEC4>(er, n, c < ch ? (c * 64 + lane) * 4 : 0, h.e[c]);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = lane + 64 * j;
    h.id[j] = m < kh ? (int)ids[m] : n;
    h.eh[j] = er[h.id[j] < n ? h.id[j] : 0];
  }
}

// One wavefront, one row.  tr: the row of tau (global memory or LDS); h: what head_eta_fetch brought of the row of eta and of the
// caller's head table row (kh = 16 spl slots, the last one = the live count); bm: 32 words of LDS private to this wavefront; hl:
// where the head row goes (16 lanes x sp_lane_bytes(spl)); ch: 256-candidate chunks of the 64-lane walk (ld / 256 -- the
// summation order of the tail total is the 64-lane scan's, restated in the CPU checker as sparse_tail_scan).  Value of slot m =
// P[id_m] for the live slots, +0 for the others; the last slot = the tail total (the 64-lane scan total of the row's non-head
// entries).  The id of an empty slot and of the last slot is `dead` (>= n): its visited flag is never set, so the scan needs no
// "is a candidate" select.
// RACE (the exponential race on head rows): value = 1 / P[id_m] (+inf for the other slots), last slot = the smallest 1 / P of
// the tail, i.e. the reciprocal of its largest entry.
// pr: the row of the dense matrix P = tau * eta (ld = 256 ch floats, zeros past n) that the rare ways of the scan walk, written
// from the same registers.  (Round 6 tried to do without P -- 65 MB written per iteration at the headline shape -- and let the
// rare ways read tau and eta instead: the headline launch got 70 us SLOWER on the same box.  A wavefront stops for a row walk
// some thirty times per tour, the walk is a bare memory round trip, and a row written a moment ago is found in the L2 / Infinity
// Cache where the two rows of tau and eta are not.  profiles/r06_fused_head_rows.txt.)
template <bool RACE, int CH, bool VEC4>
__device__ __forceinline__ void emit_head_row_pre(int n, int ch, const float *tr, const HeadEta<CH> &h, uint32_t *bm, char *hl,
                                                  float *pr, int spl, int dead, int lane) {
  const int kh = 16 * spl, ls = sp_lane_bytes(spl);
  const int cnt = h.cnt;
  if (lane < 32) bm[lane] = 0u;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int m = lane + 64 * j, id = h.id[j]; if (m < cnt && id < n) atomicOr(&bm[id >> 5], 1u << (id & 31)); }
  __builtin_amdgcn_wave_barrier();
  float part = RACE ? __builtin_inff() : 0.0f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if (c < ch) {
      const int k0 = (c * 64 + lane) * 4;
      float4 t;
      sp_load4<VEC4>(tr, n, k0, t);
      const float4 v = sp_mul4_masked(t, h.e[c], n, k0);
      *reinterpret_cast<float4 *>(pr + k0) = v;
      const uint32_t w = bm[(k0 >> 5) & 31] >> (k0 & 31);          // the four candidates share a word
      if constexpr (RACE) {
        part = fminf(part, (w & 1u) ? __builtin_inff() : 1.0f / v.x);
        part = fminf(part, (w & 2u) ? __builtin_inff() : 1.0f / v.y);
        part = fminf(part, (w & 4u) ? __builtin_inff() : 1.0f / v.z);
        part = fminf(part, (w & 8u) ? __builtin_inff() : 1.0f / v.w);
      } else {
        part = part + ((w & 1u) ? 0.0f : v.x);
        part = part + ((w & 2u) ? 0.0f : v.y);
        part = part + ((w & 4u) ? 0.0f : v.z);
        part = part + ((w & 8u) ? 0.0f : v.w);
      }
    }
  }
  float T;
  if constexpr (RACE) {
    for (int o = 32; o >= 1; o >>= 1) part = fminf(part, __shfl_xor(part, o));
    T = part;
  } else {
    T = readlane_f(wave_scan_add(part), 63);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {                                  // slot m: lane m / spl of the row, element m % spl
    const int m = lane + 64 * j;
    if (m < kh) {
      const int id = h.id[j];
      const bool live = m < cnt && id < n;
      const float pid = live ? tr[id] * h.eh[j] : 0.0f;
      float val;
      if constexpr (RACE) val = m == kh - 1 ? T : (live ? 1.0f / pid : __builtin_inff());
      else val = m == kh - 1 ? T : (live ? pid : 0.0f);
      char *hs = hl + (m / spl) * ls;
      *reinterpret_cast<float *>(hs + (m % spl) * 4) = val;
      *reinterpret_cast<uint16_t *>(hs + spl * 4 + (m % spl) * 2) = (uint16_t)(live && m != kh - 1 ? id : dead);
    }
  }
  __builtin_amdgcn_wave_barrier();                              // (bm is reused by the wavefront's next row)
}

template <bool RACE, int CH, bool VEC4>
__device__ __forceinline__ void emit_head_row(int n, int ch, const float *tr, const float *er,
                                              const uint16_t *ids, uint32_t *bm, char *hl, float *pr, int spl, int dead, int lane) {
  HeadEta<CH> h;
  head_eta_fetch<CH, VEC4>(h, n, ch, er, ids, spl, lane);
  emit_head_row_pre<RACE, CH, VEC4>(n, ch, tr, h, bm, hl, pr, spl, dead, lane);
}

// what the update needs to emit head rows (null eta: no emission)
struct HeadEmit {
  const float *eta = nullptr;      // [B][n][n] or one shared [n][n] (eta_bs = 0)
  long eta_bs = 0;
  const uint16_t *hid = nullptr;   // [B][n][16 spl]
  char *hrow = nullptr;            // [B][n][16 sp_lane_bytes(spl)]
  float *P = nullptr;              // [B][n][256 ch]: the dense rows the scan's rare ways walk
  int spl = 4, ch = 2, dead = 512, race = 0;
  int nbr_grouped = 0;             // the update's table arrives as [B][ceil(A/8)][n][8] (written by daco_tsp_sample_heads(nbr_grouped = 1))
};

}  // namespace daco
