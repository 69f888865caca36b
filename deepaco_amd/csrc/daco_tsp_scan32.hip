// daco_tsp_scan32.hip -- TSP tour construction, prefix-scan draw, TWO ants per wavefront.
//
// Same reference behaviour as daco_tsp_sample.hip in DACO_SCAN mode (tsp/aco.py:134-177 with the
// roulette draw of tsp_nls/aco.py:260-275), for 256 < n <= 512.  The one-ant-per-wave kernel is
// instruction-issue bound and two thirds of its instructions are per-STEP overhead (DPP scan,
// ballot, lane picks, stores, loop) rather than per-candidate work.  Here each 32-lane half of a
// wave builds one tour, so that overhead is paid once for two ants:
//   * candidate k of an ant sits in lane s = (k/4) % 32 of its half, chunk c = k/128
//     (16-byte loads, 512 contiguous bytes per half-wave load);
//   * the inclusive scan needs only the in-row DPP steps plus row_bcast:15 (halves never mix);
//   * per-ant values travel over the LDS crossbar (ds_swizzle / ds_bpermute, no VALU slot) or as
//     two 16-bit fields of one scalar register picked per half with a single v_bfe;
//   * picking the candidate inside the chosen lane reuses the scan: that lane deals its masked
//     values to the 32 lanes of its half through LDS and the same scan + first-lane pick runs
//     across candidates (instead of a compare-and-count chain every lane would have to execute).
// Draw semantics are the 32-lane variant of the scan specification; the GPU
// tests hold it bit-exact against the CPU restatement of that specification.
#include "daco_sample_kernel.h"
#include <cstdlib>

namespace daco {

// LLVM floating-point compare predicates for __builtin_amdgcn_fcmpf (wave-wide result mask)
constexpr int FCMP_OGT = 2, FCMP_OGE = 3;

typedef float f32x2 __attribute__((ext_vector_type(2)));

// inclusive add-scan inside each 32-lane half: Kogge-Stone in rows of 16, then (ROWS2) the second
// row of each half adds the first row's total (lanes 16..31 += lane 15, lanes 48..63 += lane 47)
template <bool ROWS2>
__device__ inline float half_scan_add(float x) {
  x = x + dpp_f<DPP_ROW_SHR(1), 0xF, true>(0.0f, x);
  x = x + dpp_f<DPP_ROW_SHR(2), 0xF, true>(0.0f, x);
  x = x + dpp_f<DPP_ROW_SHR(4), 0xF, true>(0.0f, x);
  x = x + dpp_f<DPP_ROW_SHR(8), 0xF, true>(0.0f, x);
  // rows 1 and 3 only (row_mask 0xa): x += lane 15 of the row before; one instruction, rows 0 / 2 keep x.
  // s_nop 1 on both sides: two wait states between a VALU write of x and a DPP read of it -- the
  // compiler's hazard recogniser does not look inside the asm, nor does it know what follows it.
  if constexpr (ROWS2)
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1" : "+v"(x));
  return x;
}
// lane 31 of each half to all of its lanes (LDS crossbar, no memory, no VALU slot)
__device__ inline float half_bcast_last(float x) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x3E0));   // BROADCAST,32,31
}
__device__ inline uint32_t lowest_bit(uint32_t x) { return x & (0u - x); }

// FUSED: tour lengths and the neighbour table are both produced (the colony iteration's call)
template <int CH2, bool LOGP, bool FUSED>
__global__ void __launch_bounds__(256)
tsp_scan32_kernel(const SampleParams p) {
  constexpr int NJ = CH2 * 4;                           // candidates per lane (<= 32)
  constexpr int ROWF = CH2 * 128;                       // padded row length of this layout
  // open[h][k] = 1.0f while node k is unvisited by ant h of the workgroup, else 0.0f
  __shared__ __attribute__((aligned(16))) float open_flags[8][ROWF];
  // per ant: [0..31] candidate slots of the chosen lane, [32] threshold, [33] chosen lane, [34] choice
  __shared__ __attribute__((aligned(16))) float pick[8][40];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int up = lane >> 5, s = lane & 31;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int bpi = (p.A + 7) >> 3;                       // workgroups per instance (8 ants each)
  const int b = w / bpi;
  const int a0 = ((w - b * bpi) * 4 + wave) * 2;        // ants a0 (lower half), a0+1 (upper half)
  if (a0 >= p.A) return;
  const int n = p.n, A = p.A, ld = p.ld;
  const uint64_t iter_now = p.iter + (p.iter_dev ? *p.iter_dev : 0ull);   // a captured graph advances *iter_dev
  // odd A: the last upper half builds ant A-1 a second time (same counters, same tour, same stores)
  const int a = a0 + up < A ? a0 + up : A - 1;
  const bool lead = __builtin_amdgcn_inverse_ballot_w64(0x0000000100000001ull);   // lane 0 of each half
  const uint32_t gid = p.ant_gid0 + (uint32_t)(b * (p.gid_bstride ? p.gid_bstride : A) + a);
  const char *Pb = (const char *)(p.P + (size_t)b * n * ld);           // uniform; lanes add 32-bit offsets
  const uint32_t ldb = (uint32_t)ld * 4u, lane_off = (uint32_t)s * 16u;
  char *path_t = (char *)(p.paths + (size_t)b * n * A);                // row t of this instance's [n][A] block
  const uint32_t a8 = (uint32_t)a * 8u, a4 = (uint32_t)a * 4u;
  char *logp_t = LOGP ? (char *)(p.logp + (size_t)b * (n - 1) * A) : nullptr;
  char *rs_t = (LOGP && p.rowsum) ? (char *)(p.rowsum + (size_t)b * (n - 1) * A) : nullptr;
  const bool want_cost = FUSED || p.costs != nullptr, want_nbr = FUSED || p.nbr != nullptr;
  const char *dist_b = want_cost ? (const char *)(p.dist + (size_t)b * p.dist_bs) : nullptr;
  char *nbr_b = want_nbr ? (char *)(p.nbr + (size_t)b * n * A) : nullptr;        // [n][A] table of this instance
  const uint32_t A4 = (uint32_t)A * 4u;
  float *fl = open_flags[wave * 2 + up], *pk = pick[wave * 2 + up];
#pragma unroll
  for (int c = 0; c < CH2; ++c) *(float4 *)(fl + (c * 32 + s) * 4) = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
  pk[s] = 0.0f;                                         // slots >= NJ stay zero for the whole kernel
  if (s < 8) pk[32 + s] = 0.0f;
  const int ubase = (lane & 32) << 2;                   // ds_bpermute byte address of this half's lane 0
  // slot j = s of the chosen lane L is candidate (j/4)*128 + L*4 + j%4; lanes beyond NJ hold no slot
  const int cbase = s < NJ ? ((s >> 2) << 7) | (s & 3) : 0;

  int prev;
  if (p.start) prev = (int)p.start[(size_t)b * A + a];
  else if (p.fixed_start >= 0) prev = p.fixed_start;
  else {
    const u32x4 r = rng_block(p.seed, iter_now, STREAM_START, gid, 0);
    prev = (int)__umulhi(r.x, (uint32_t)n);
  }
  const int first = prev;
  __builtin_amdgcn_wave_barrier();
  if (lead) {
    fl[prev] = 0.0f;
    *(int64_t *)(path_t + a8) = prev;
  }
  __builtin_amdgcn_wave_barrier();
  int pprev = 0;
  float cost = 0.0f, dpend = 0.0f;
  u32x4 ublk = {0, 0, 0, 0};                            // 128 cached uniforms per ant (lane s: block base+s)
  uint64_t feasible = ~0ull;

  for (int tb = 0; tb < n; tb += 32) {
    // uniform of step t: lane (t&31), component (t>>5)&3 of Philox block ((t>>7)<<5) + lane
    if ((tb & 127) == 0) ublk = rng_block(p.seed, iter_now, STREAM_SCAN, gid, (uint32_t)(((tb >> 7) << 5) + s));
    const float ucur = u01(comp(ublk, (tb >> 5) & 3));
    const int i1 = n - tb < 32 ? n - tb : 32;
    for (int i = tb == 0 ? 1 : 0; i < i1; ++i) {
      path_t += (size_t)A * 8;
      const uint32_t voff = __umul24((uint32_t)prev, ldb) + lane_off;
      float4 row[CH2], fo[CH2];
#pragma unroll
      for (int c = 0; c < CH2; ++c) row[c] = *(const float4 *)(Pb + voff + c * 512);
      const float u = __int_as_float(__builtin_amdgcn_ds_bpermute(ubase | (i << 2), __float_as_int(ucur)));
#pragma unroll
      for (int c = 0; c < CH2; ++c) fo[c] = *(const float4 *)(fl + (c * 32 + s) * 4);

      // ---- level 1: which lane.  Closed candidates contribute p*0 = +0.0f; even and odd slots
      // accumulate separately (packed fma: the products are exact, so each fma is one rounding)
      f32x2 acc = {0.0f, 0.0f};
#pragma unroll
      for (int c = 0; c < CH2; ++c) {
        acc = __builtin_elementwise_fma((f32x2){row[c].x, row[c].y}, (f32x2){fo[c].x, fo[c].y}, acc);
        acc = __builtin_elementwise_fma((f32x2){row[c].z, row[c].w}, (f32x2){fo[c].z, fo[c].w}, acc);
      }
      const float part = acc.x + acc.y;
      const float incl = half_scan_add<true>(part);
      const float S = half_bcast_last(incl);
      const float r = fmaxf(u * S, 1.401298464e-45f);   // keep r > 0 if u*S underflows
      const uint64_t m = __builtin_amdgcn_fcmpf(incl, r, FCMP_OGE) & __builtin_amdgcn_fcmpf(part, 0.0f, FCMP_OGT);
      const uint64_t alive = __builtin_amdgcn_fcmpf(S, 0.0f, FCMP_OGT);    // S > 0 <=> some open candidate has p > 0
      feasible &= alive;
      // what is left to cover inside the chosen lane: r - incl[L-1]; lane L forms its own
      float excl = dpp_f<0x138 /* wave_shr:1 */, 0xF, true>(0.0f, incl);
      excl = s == 0 ? 0.0f : excl;
      const float thr = r - excl;
      // ---- level 2: which candidate of lane L.  Lane L deals its NJ values to the lanes of its
      // half through LDS; the same scan + first-lane pick then runs across candidates.
      const bool mine = __builtin_amdgcn_inverse_ballot_w64((uint64_t)lowest_bit((uint32_t)m) |
                                                            ((uint64_t)lowest_bit((uint32_t)(m >> 32)) << 32));
      if (mine) {
#pragma unroll
        for (int c = 0; c < CH2; ++c) *(float4 *)(pk + 4 * c) = row[c];
        *(float2 *)(pk + 32) = make_float2(thr, __int_as_float(s));
      }
      __builtin_amdgcn_wave_barrier();
      const float cvraw = pk[s];
      const float2 tl = *(const float2 *)(pk + 32);
      const int mychoice = cbase + (__float_as_int(tl.y) << 2);
      const float cv = cvraw * fl[mychoice];
      const float sc = half_scan_add<(NJ > 16)>(cv);
      const uint64_t pos = __builtin_amdgcn_fcmpf(cv, 0.0f, FCMP_OGT) & alive;   // (a dead row holds stale slots)
      const uint64_t k = __builtin_amdgcn_fcmpf(sc, tl.x, FCMP_OGE) & pos;
      uint32_t k0 = (uint32_t)k, k1 = (uint32_t)(k >> 32);
      if (__builtin_expect(k0 == 0 || k1 == 0, 0)) {
        // rounding: no candidate reached thr -> the lane's last open candidate with p > 0
        const uint32_t q0 = (uint32_t)pos, q1 = (uint32_t)(pos >> 32);
        if (k0 == 0 && q0) k0 = 0x80000000u >> __builtin_clz(q0);
        if (k1 == 0 && q1) k1 = 0x80000000u >> __builtin_clz(q1);
      }
      const bool win = __builtin_amdgcn_inverse_ballot_w64((uint64_t)lowest_bit(k0) | ((uint64_t)lowest_bit(k1) << 32));
      if (win) {
        pk[34] = __int_as_float(mychoice);
        fl[mychoice] = 0.0f;                            // visited
      }
      __builtin_amdgcn_wave_barrier();
      // no feasible candidate (flagged; the reference raises): move to node 0 like the one-ant kernel and the oracle
      const int choice = S > 0.0f ? __float_as_int(pk[34]) : 0;
      __builtin_amdgcn_wave_barrier();

      if (lead) {
        if (!(S > 0.0f)) fl[0] = 0.0f;
        *(int64_t *)(path_t + a8) = choice;
        if constexpr (LOGP) {
          const float pc = *(const float *)(Pb + __umul24((uint32_t)prev, ldb) + (uint32_t)choice * 4u);
          *(float *)(logp_t + a4) = clamp_log(pc / S);
          logp_t += (size_t)A * 4;
          if (rs_t) { *(float *)(rs_t + a4) = S; rs_t += (size_t)A * 4; }
        }
        if (want_cost) {                                 // fused tour length, edge added one step late
          cost = cost + dpend;
          dpend = *(const float *)(dist_b + ((__umul24((uint32_t)choice, (uint32_t)n) + (uint32_t)prev) << 2));
        }
        if (want_nbr) *(uint32_t *)(nbr_b + __umul24((uint32_t)prev, A4) + a4) = (uint32_t)pprev | ((uint32_t)choice << 16);
      }
      pprev = prev;
      prev = choice;
    }
  }
  if (lead) {
    if (want_cost) {
      cost = cost + dpend;
      cost = cost + *(const float *)(dist_b + ((__umul24((uint32_t)first, (uint32_t)n) + (uint32_t)prev) << 2));
      p.costs[(size_t)b * A + a] = cost;
    }
    if (want_nbr) {                                     // close the cycle: last -> first -> second
      uint32_t *nbr_a = (uint32_t *)(nbr_b + a4);        // + node * A
      const int second = (int)p.paths[((size_t)b * n + 1) * A + a];
      if (n == 2) { nbr_a[(size_t)first * A] = (uint32_t)prev | ((uint32_t)prev << 16); nbr_a[(size_t)prev * A] = (uint32_t)first | ((uint32_t)first << 16); }
      else { nbr_a[(size_t)prev * A] = (uint32_t)pprev | ((uint32_t)first << 16); nbr_a[(size_t)first * A] = (uint32_t)prev | ((uint32_t)second << 16); }
    }
  }
  if (feasible != ~0ull && p.flags && lane == 0) atomicOr(p.flags + b, 1);
}

template <int CH2>
static hipError_t launch32(const SampleParams &sp, bool logp, hipStream_t s) {
  const int bpi = (sp.A + 7) / 8;
  dim3 grid((unsigned)(sp.B * bpi)), block(256);
  const bool fused = sp.costs && sp.nbr;
  static const int pad = getenv("DACO_SCAN32_LDS_PAD") ? atoi(getenv("DACO_SCAN32_LDS_PAD")) : 0;
#define DACO_L32(L, F) hipLaunchKernelGGL((tsp_scan32_kernel<CH2, L, F>), grid, block, pad, s, sp)
  if (logp) { if (fused) DACO_L32(true, true); else DACO_L32(true, false); }
  else { if (fused) DACO_L32(false, true); else DACO_L32(false, false); }
#undef DACO_L32
  return hipGetLastError();
}

// ------------------------------------------------------------------ CVRP (cvrp/aco.py:138-205)
// Same draw, the closed set now also holds the customers whose demand exceeds the remaining
// capacity (strict, cvrp/aco.py:200) and the depot while the ant stands on it with customers left
// (:179).  Lanes multiply the row by the combined 0/1 factor, the chosen lane deals those masked
// values.  The two ants of a wave finish at different steps: a finished half keeps stepping with
// its stores and state updates switched off until its neighbour is done.
template <int CH2, bool LOGP, bool FUSED>
__global__ void __launch_bounds__(256)
cvrp_scan32_kernel(const SampleParams p) {
  constexpr int NJ = CH2 * 4, ROWF = CH2 * 128;
  __shared__ __attribute__((aligned(16))) float open_flags[8][ROWF];
  __shared__ __attribute__((aligned(16))) float pick[8][40];
  __shared__ __attribute__((aligned(16))) float dem_s[ROWF];          // demand of this instance, +inf padding
  __shared__ uint32_t hub_s[8][CH2 * 4];                               // per ant: set of nodes that follow the depot
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int up = lane >> 5, s = lane & 31;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int bpi = (p.A + 7) >> 3;
  const int b = w / bpi;
  const int a0 = ((w - b * bpi) * 4 + wave) * 2;
  const int n = p.n, A = p.A, ld = p.ld, Lmax = p.Lmax;
  const uint64_t iter_now = p.iter + (p.iter_dev ? *p.iter_dev : 0ull);
  for (int k = threadIdx.x; k < ROWF; k += 256) dem_s[k] = k < n ? p.demand[(size_t)b * n + k] : __builtin_inff();
  __syncthreads();
  if (a0 >= A) return;
  const int a = a0 + up < A ? a0 + up : A - 1;          // odd A: the last upper half repeats ant A-1
  const bool lead = __builtin_amdgcn_inverse_ballot_w64(0x0000000100000001ull);
  const uint32_t gid = p.ant_gid0 + (uint32_t)(b * (p.gid_bstride ? p.gid_bstride : A) + a);
  const char *Pb = (const char *)(p.P + (size_t)b * n * ld);
  const uint32_t ldb = (uint32_t)ld * 4u, lane_off = (uint32_t)s * 16u;
  int64_t *path_a = p.paths + (size_t)b * Lmax * A + a;
  float *logp_a = LOGP ? p.logp + (size_t)b * (Lmax - 1) * A + a : nullptr;
  float *rs_a = (LOGP && p.rowsum) ? p.rowsum + (size_t)b * (Lmax - 1) * A + a : nullptr;
  float *fl = open_flags[wave * 2 + up], *pk = pick[wave * 2 + up];
  const char *dist_b = (FUSED || p.costs) ? (const char *)(p.dist + (size_t)b * p.dist_bs) : nullptr;
  char *next_b = (FUSED || p.nbr) ? (char *)(p.nbr + (size_t)b * n * A) : nullptr;           // [n][A] table of this instance
  const uint32_t A4 = (uint32_t)A * 4u, a4 = (uint32_t)a * 4u;
  uint32_t *hub_l = hub_s[wave * 2 + up];
  if (s < CH2 * 4) hub_l[s] = 0u;
  float cost = 0.0f, dpend = 0.0f;
  float4 dm[CH2];
#pragma unroll
  for (int c = 0; c < CH2; ++c) {
    *(float4 *)(fl + (c * 32 + s) * 4) = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    dm[c] = *(const float4 *)(dem_s + (c * 32 + s) * 4);
  }
  pk[s] = 0.0f;
  if (s < 8) pk[32 + s] = 0.0f;
  const int ubase = (lane & 32) << 2;
  const int cbase = s < NJ ? ((s >> 2) << 7) | (s & 3) : 0;
  __builtin_amdgcn_wave_barrier();
  if (lead) path_a[0] = 0;

  int prev = 0, remaining = n - 1, len = 1;
  float used = 0.0f + dem_s[0];
  bool finished = remaining == 0;
  u32x4 ublk = {0, 0, 0, 0};
  float ucur = 0.0f;
  uint64_t feasible = ~0ull;
  uint64_t act = __builtin_amdgcn_ballot_w64(!finished);               // lanes of the halves still building

  for (int t = 1; t < Lmax && act != 0; ++t) {
    const uint32_t voff = __umul24((uint32_t)prev, ldb) + lane_off;
    float4 row[CH2], fo[CH2];
#pragma unroll
    for (int c = 0; c < CH2; ++c) row[c] = *(const float4 *)(Pb + voff + c * 512);
    if ((t & 31) == 0 || t == 1) {
      if ((t & 127) == 0 || t == 1) ublk = rng_block(p.seed, iter_now, STREAM_SCAN, gid, (uint32_t)(((t >> 7) << 5) + s));
      ucur = u01(comp(ublk, (t >> 5) & 3));
    }
    const float u = __int_as_float(__builtin_amdgcn_ds_bpermute(ubase | ((t & 31) << 2), __float_as_int(ucur)));
#pragma unroll
    for (int c = 0; c < CH2; ++c) fo[c] = *(const float4 *)(fl + (c * 32 + s) * 4);
    const float rem = p.capacity - used;
    f32x2 acc = {0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < CH2; ++c) {
      float4 f = fo[c];
      f.x = dm[c].x > rem ? 0.0f : f.x;  f.y = dm[c].y > rem ? 0.0f : f.y;
      f.z = dm[c].z > rem ? 0.0f : f.z;  f.w = dm[c].w > rem ? 0.0f : f.w;
      if (c == 0) f.x = (s == 0 && prev == 0 && remaining > 0) ? 0.0f : f.x;       // the depot, cvrp/aco.py:179
      const f32x2 lo = (f32x2){row[c].x, row[c].y} * (f32x2){f.x, f.y}, hi = (f32x2){row[c].z, row[c].w} * (f32x2){f.z, f.w};
      row[c] = make_float4(lo.x, lo.y, hi.x, hi.y);
      acc = acc + lo;
      acc = acc + hi;
    }
    const float part = acc.x + acc.y;
    const float incl = half_scan_add<true>(part);
    const float S = half_bcast_last(incl);
    const float r = fmaxf(u * S, 1.401298464e-45f);
    const uint64_t m = __builtin_amdgcn_fcmpf(incl, r, FCMP_OGE) & __builtin_amdgcn_fcmpf(part, 0.0f, FCMP_OGT) & act;
    feasible &= __builtin_amdgcn_fcmpf(S, 0.0f, FCMP_OGT) | ~act;
    float excl = dpp_f<0x138 /* wave_shr:1 */, 0xF, true>(0.0f, incl);
    excl = s == 0 ? 0.0f : excl;
    const float thr = r - excl;
    const bool mine = __builtin_amdgcn_inverse_ballot_w64((uint64_t)lowest_bit((uint32_t)m) |
                                                          ((uint64_t)lowest_bit((uint32_t)(m >> 32)) << 32));
    if (mine) {
#pragma unroll
      for (int c = 0; c < CH2; ++c) *(float4 *)(pk + 4 * c) = row[c];
      *(float2 *)(pk + 32) = make_float2(thr, __int_as_float(s));
    }
    __builtin_amdgcn_wave_barrier();
    const float cv = pk[s];
    const float2 tl = *(const float2 *)(pk + 32);
    const int mychoice = cbase + (__float_as_int(tl.y) << 2);
    const float sc = half_scan_add<(NJ > 16)>(cv);
    const uint64_t pos = __builtin_amdgcn_fcmpf(cv, 0.0f, FCMP_OGT) & act;
    const uint64_t k = __builtin_amdgcn_fcmpf(sc, tl.x, FCMP_OGE) & pos;
    uint32_t k0 = (uint32_t)k, k1 = (uint32_t)(k >> 32);
    if (__builtin_expect(k0 == 0 || k1 == 0, 0)) {
      const uint32_t q0 = (uint32_t)pos, q1 = (uint32_t)(pos >> 32);
      if (k0 == 0 && q0) k0 = 0x80000000u >> __builtin_clz(q0);
      if (k1 == 0 && q1) k1 = 0x80000000u >> __builtin_clz(q1);
    }
    const bool win = __builtin_amdgcn_inverse_ballot_w64((uint64_t)lowest_bit(k0) | ((uint64_t)lowest_bit(k1) << 32));
    if (win) {
      pk[34] = __int_as_float(mychoice);
      if (mychoice != 0) fl[mychoice] = 0.0f;            // customers are visited once, the depot stays open
    }
    __builtin_amdgcn_wave_barrier();
    const int choice = __float_as_int(pk[34]);
    __builtin_amdgcn_wave_barrier();

    // ---- outputs: lane 0 of every half that is still building (one EXEC mask, no nesting)
    const bool writer = __builtin_amdgcn_inverse_ballot_w64(act & 0x0000000100000001ull);
    if (writer) {
      path_a[(size_t)t * A] = choice;
      if constexpr (LOGP) {
        const float pc = *(const float *)(Pb + __umul24((uint32_t)prev, ldb) + (uint32_t)choice * 4u);
        logp_a[(size_t)(t - 1) * A] = clamp_log(pc / S);
        if (rs_a) rs_a[(size_t)(t - 1) * A] = S;
      }
      if (FUSED || dist_b) {                             // fused route length, edge added one step late
        cost = cost + dpend;
        dpend = *(const float *)(dist_b + ((__umul24((uint32_t)prev, (uint32_t)n) + (uint32_t)choice) << 2));
      }
      if (FUSED || next_b) {
        // who follows `prev`.  Row 0 of the table is never read (the depot's successors are a set,
        // kept as a bitmap), so the store needs no branch on prev
        *(uint32_t *)(next_b + __umul24((uint32_t)prev, A4) + a4) = (uint32_t)choice << 16;
        __hip_atomic_fetch_or(hub_l + (choice >> 5), prev == 0 ? 1u << (choice & 31) : 0u, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    }
    // ---- state of each half, as selects (a finished half keeps its values)
    const bool live = !finished;
    const bool moved = live && choice != 0;
    remaining -= moved ? 1 : 0;
    const float load = moved ? used : 0.0f;              // back at the depot the load restarts from 0
    used = live ? load + dem_s[choice] : used;
    finished = finished || (remaining == 0 && choice == 0);
    len = live ? t + 1 : len;
    prev = finished ? 0 : choice;
    act = __builtin_amdgcn_ballot_w64(!finished);
  }
  // the reference steps every ant until the slowest one is done: a done ant keeps drawing the
  // depot (probability 1), so its column is padded with 0 / log(1-eps)
  if (s == 0) {
    if (p.lens) p.lens[(size_t)b * A + a] = len;
    if (p.tab_lens) p.tab_lens[(size_t)b * A + a] = len;
    const float lp1 = clamp_log(1.0f);
    for (int tt = len; tt < Lmax; ++tt) {
      path_a[(size_t)tt * A] = 0;
      if constexpr (LOGP) logp_a[(size_t)(tt - 1) * A] = lp1;
    }
    if (!finished && p.flags) atomicOr(p.flags + b, 2);
    if (dist_b) p.costs[(size_t)b * A + a] = cost + dpend;
    if (next_b) {
      uint32_t *hub_a = p.hubmask + ((size_t)b * A + a) * ((n + 31) >> 5);
      for (int i = 0; i < ((n + 31) >> 5); ++i) hub_a[i] = hub_l[i];
    }
  }
  if (feasible != ~0ull && p.flags && lane == 0) atomicOr(p.flags + b, 1);
}

template <int CH2>
static hipError_t launch_cvrp32(const SampleParams &sp, bool logp, hipStream_t s) {
  const int bpi = (sp.A + 7) / 8;
  dim3 grid((unsigned)(sp.B * bpi)), block(256);
  const bool fused = sp.costs && sp.nbr;
#define DACO_C32(L, F) hipLaunchKernelGGL((cvrp_scan32_kernel<CH2, L, F>), grid, block, 0, s, sp)
  if (logp) { if (fused) DACO_C32(true, true); else DACO_C32(true, false); }
  else { if (fused) DACO_C32(false, true); else DACO_C32(false, false); }
#undef DACO_C32
  return hipGetLastError();
}

// entry used by daco_cvrp_sample when the two-ants-per-wave layout applies
hipError_t launch_cvrp_scan32(const SampleParams &sp, bool logp, hipStream_t s) {
  switch ((sp.n + 127) / 128) {
    case 1: return launch_cvrp32<1>(sp, logp, s);
    case 2: return launch_cvrp32<2>(sp, logp, s);
    case 3: return launch_cvrp32<3>(sp, logp, s);
    case 4: return launch_cvrp32<4>(sp, logp, s);
    case 5: return launch_cvrp32<5>(sp, logp, s);
    case 6: return launch_cvrp32<6>(sp, logp, s);
    case 7: return launch_cvrp32<7>(sp, logp, s);
    default: return launch_cvrp32<8>(sp, logp, s);
  }
}

// entry used by daco_tsp_sample (daco_tsp_sample.hip) when the two-ants-per-wave layout applies
hipError_t launch_tsp_scan32(const SampleParams &sp, bool logp, hipStream_t s) {
  switch ((sp.n + 127) / 128) {
    case 1: return launch32<1>(sp, logp, s);
    case 2: return launch32<2>(sp, logp, s);
    case 3: return launch32<3>(sp, logp, s);
    case 4: return launch32<4>(sp, logp, s);
    case 5: return launch32<5>(sp, logp, s);
    case 6: return launch32<6>(sp, logp, s);
    case 7: return launch32<7>(sp, logp, s);
    default: return launch32<8>(sp, logp, s);
  }
}

}  // namespace daco
