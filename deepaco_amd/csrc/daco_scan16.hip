// daco_scan16.hip -- tour construction for small instances (n <= 256), prefix-scan draw, FOUR ants
// per wavefront: TSP (tsp/aco.py:134-177, tsp_nls/aco.py:184-220) and CVRP (cvrp/aco.py:138-205).
//
// At n <= 256 a row is at most 1 KiB: the step is bound by instruction issue and by the
// latency of its dependent chain, not by bytes, so the per-step overhead is shared by four ants.
// Each 16-lane DPP row of a wave builds one tour:
//   * candidate k of an ant sits in lane s = (k/4) % 16 of its row, chunk c = k/64 (c < CH <= 4);
//   * visited flags are f32 0/1 in LDS, the lane's masked values p*open feed packed adds;
//   * level 1: DPP row scan of the 16 lane sums (no cross-row step at all), S by ds_swizzle, the
//     first lane with incl >= u*S per row from v_mbcnt (bits below me == bits below my row);
//   * level 2: that lane deals its <= 16 masked values to the lanes of its row through LDS and
//     the same scan + first-lane pick runs across candidates; the winner publishes the choice.
// Draw semantics: the 16-lane variant of the scan specification; the GPU
// tests hold it bit-exact against the CPU restatement of that specification.
#include "daco_sample_kernel.h"

namespace daco {

constexpr int FCMP16_OGT = 2, FCMP16_OGE = 3;
typedef float f32x2_16 __attribute__((ext_vector_type(2)));

// inclusive add-scan inside each 16-lane row (Kogge-Stone, DPP row_shr)
template <int STEPS>
__device__ inline float row_scan_add(float x) {
  x = x + dpp_f<DPP_ROW_SHR(1), 0xF, true>(0.0f, x);
  x = x + dpp_f<DPP_ROW_SHR(2), 0xF, true>(0.0f, x);
  if constexpr (STEPS > 2) x = x + dpp_f<DPP_ROW_SHR(4), 0xF, true>(0.0f, x);
  if constexpr (STEPS > 3) x = x + dpp_f<DPP_ROW_SHR(8), 0xF, true>(0.0f, x);
  return x;
}
// lane 15 of each row to all of its lanes (LDS crossbar: lane' = (lane & 0x10) | 0x0F inside each 32)
__device__ inline float row_bcast_last(float x) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x10 | (0x0F << 5)));
}
// of the lanes set in m, the first one of every 16-lane row
__device__ inline uint64_t row_first(uint64_t m) {
  const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  const int base = __builtin_amdgcn_ds_swizzle(below, 0x10);           // the same count at the first lane of my row
  return __builtin_amdgcn_ballot_w64(below == base) & m;
}
// conservative test: does some 16-bit field of x look empty (borrows may add false positives)
__device__ inline bool some_row_empty(uint64_t x) {
  return ((x - 0x0001000100010001ull) & ~x & 0x8000800080008000ull) != 0;
}

// CH: chunks of 64 candidates (n <= 64 * CH, CH <= 4).  FUSED: costs and the update's table too.
template <int CH, bool LOGP, bool FUSED, bool CVRP>
__global__ void __launch_bounds__(256)
scan16_kernel(const SampleParams p) {
  constexpr int NJ = CH * 4;                            // candidates per lane
  constexpr int ROWF = CH * 64;                         // padded row length of this layout
  constexpr int S2 = CH == 1 ? 2 : (CH == 2 ? 3 : 4);   // scan steps that cover NJ slots
  __shared__ __attribute__((aligned(16))) float open_flags[16][ROWF];
  // per ant: [0..15] candidate slots of the chosen lane (NJ used), [16] threshold, [17] chosen lane, [18] choice
  __shared__ __attribute__((aligned(16))) float pick[16][24];
  __shared__ __attribute__((aligned(16))) float dem_s[CVRP ? ROWF : 4];   // CVRP: demand, +inf padding
  __shared__ uint32_t hub_s[16][8];                     // CVRP: per ant, set of nodes that follow the depot (n <= 256)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, s = lane & 15;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int bpi = (p.A + 15) >> 4;                      // workgroups per instance (16 ants each)
  const int b = w / bpi;
  const int a0 = ((w - b * bpi) * 4 + wave) * 4;        // ants a0 .. a0+3, one per row
  const int n = p.n, A = p.A, ld = p.ld;
  const uint64_t iter_now = p.iter + (p.iter_dev ? *p.iter_dev : 0ull);   // a captured graph advances *iter_dev
  if constexpr (CVRP) {
    for (int k = threadIdx.x; k < ROWF; k += 256) dem_s[k] = k < n ? p.demand[(size_t)b * n + k] : __builtin_inff();
    __syncthreads();
  }
  if (a0 >= A) return;
  // A not a multiple of 4: the spare rows build ant A-1 again (same counters, same tour, same stores)
  const int a = a0 + q < A ? a0 + q : A - 1;
  const uint64_t LEAD = 0x0001000100010001ull;          // lane 0 of each row
  const uint32_t gid = p.ant_gid0 + (uint32_t)(b * (p.gid_bstride ? p.gid_bstride : A) + a);
  const char *Pb = (const char *)(p.P + (size_t)b * n * ld);           // uniform; lanes add 32-bit offsets
  const uint32_t ldb = (uint32_t)ld * 4u, lane_off = (uint32_t)s * 16u;
  const int rows = CVRP ? p.Lmax : n;                   // rows of paths for one instance
  int64_t *path_a = p.paths + (size_t)b * rows * A + a;
  float *logp_a = LOGP ? p.logp + (size_t)b * (rows - 1) * A + a : nullptr;
  float *rs_a = (LOGP && p.rowsum) ? p.rowsum + (size_t)b * (rows - 1) * A + a : nullptr;
  const bool want_cost = FUSED || p.costs != nullptr, want_tab = FUSED || p.nbr != nullptr;
  const char *dist_b = want_cost ? (const char *)(p.dist + (size_t)b * p.dist_bs) : nullptr;
  char *tab_b = want_tab ? (char *)(p.nbr + (size_t)b * n * A) : nullptr;           // [n][A] table of this instance
  const uint32_t A4 = (uint32_t)A * 4u, a4 = (uint32_t)a * 4u;
  float *fl = open_flags[wave * 4 + q], *pk = pick[wave * 4 + q];
  uint32_t *hub_l = hub_s[wave * 4 + q];
  float4 dm[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    *(float4 *)(fl + (c * 16 + s) * 4) = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if constexpr (CVRP) dm[c] = *(const float4 *)(dem_s + (c * 16 + s) * 4);
  }
  pk[s] = 0.0f;                                         // slots >= NJ stay zero for the whole kernel
  if (s < 8) pk[16 + s] = 0.0f;
  if (s < 8) hub_l[s] = 0u;
  const int ubase = (lane & 48) << 2;                   // ds_bpermute byte address of this row's lane 0
  // slot j = s of the chosen lane L is candidate (j/4)*64 + L*4 + j%4; lanes beyond NJ hold no slot
  const int cbase = s < NJ ? ((s >> 2) << 6) | (s & 3) : 0;

  int prev;
  if constexpr (CVRP) prev = 0;
  else if (p.start) prev = (int)p.start[(size_t)b * A + a];
  else if (p.fixed_start >= 0) prev = p.fixed_start;
  else {
    const u32x4 r = rng_block(p.seed, iter_now, STREAM_START, gid, 0);
    prev = (int)__umulhi(r.x, (uint32_t)n);
  }
  const int first = prev;
  __builtin_amdgcn_wave_barrier();
  if (s == 0) {
    if constexpr (!CVRP) fl[prev] = 0.0f;               // the depot is never closed for good
    path_a[0] = prev;
  }
  __builtin_amdgcn_wave_barrier();
  int pprev = 0, remaining = n - 1, len = 1;
  float used = CVRP ? 0.0f + dem_s[0] : 0.0f;
  bool finished = CVRP ? remaining == 0 : false;
  float cost = 0.0f, dpend = 0.0f;
  u32x4 ublk = {0, 0, 0, 0};                            // 64 cached uniforms per ant (lane s: block base+s)
  float ucur = 0.0f;
  uint64_t feasible = ~0ull;
  uint64_t act = CVRP ? __builtin_amdgcn_ballot_w64(!finished) : ~0ull;     // lanes of the rows still building
  const int tend = CVRP ? p.Lmax : n;
  const float *uin = (!CVRP && p.noise) ? p.noise + (size_t)b * (n - 1) * A + a : nullptr;

  for (int t = 1; t < tend && act != 0; ++t) {
    const uint32_t voff = __umul24((uint32_t)prev, ldb) + lane_off;
    float4 row[CH], fo[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) row[c] = *(const float4 *)(Pb + voff + c * 256);
    // uniform of step t: lane (t&15), component (t>>4)&3 of Philox block ((t>>6)<<4) + lane
    if ((t & 15) == 0 || t == 1) {
      if ((t & 63) == 0 || t == 1) ublk = rng_block(p.seed, iter_now, STREAM_SCAN, gid, (uint32_t)(((t >> 6) << 4) + s));
      ucur = u01(comp(ublk, (t >> 4) & 3));
    }
    float u = __int_as_float(__builtin_amdgcn_ds_bpermute(ubase | ((t & 15) << 2), __float_as_int(ucur)));
    if constexpr (!CVRP) { if (uin) u = uin[(size_t)(t - 1) * A]; }       // injected uniform stream (tests): [B][n-1][A]
#pragma unroll
    for (int c = 0; c < CH; ++c) fo[c] = *(const float4 *)(fl + (c * 16 + s) * 4);

    // ---- level 1: which lane.  Closed candidates become p*0 = +0.0f; even and odd slots add up separately
    const float rem = CVRP ? p.capacity - used : 0.0f;
    f32x2_16 acc = {0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float4 f = fo[c];
      if constexpr (CVRP) {
        f.x = dm[c].x > rem ? 0.0f : f.x;  f.y = dm[c].y > rem ? 0.0f : f.y;      // strict, cvrp/aco.py:200
        f.z = dm[c].z > rem ? 0.0f : f.z;  f.w = dm[c].w > rem ? 0.0f : f.w;
        if (c == 0) f.x = (s == 0 && prev == 0 && remaining > 0) ? 0.0f : f.x;     // the depot, cvrp/aco.py:179
      }
      const f32x2_16 lo = (f32x2_16){row[c].x, row[c].y} * (f32x2_16){f.x, f.y};
      const f32x2_16 hi = (f32x2_16){row[c].z, row[c].w} * (f32x2_16){f.z, f.w};
      row[c] = make_float4(lo.x, lo.y, hi.x, hi.y);
      acc = acc + lo;
      acc = acc + hi;
    }
    const float part = acc.x + acc.y;
    const float incl = row_scan_add<4>(part);
    const float S = row_bcast_last(incl);
    const float r = fmaxf(u * S, 1.401298464e-45f);     // keep r > 0 if u*S underflows
    const uint64_t m = __builtin_amdgcn_fcmpf(incl, r, FCMP16_OGE) & __builtin_amdgcn_fcmpf(part, 0.0f, FCMP16_OGT) & act;
    const uint64_t alive = __builtin_amdgcn_fcmpf(S, 0.0f, FCMP16_OGT);   // S > 0 <=> some open candidate has p > 0
    feasible &= alive | ~act;
    // what is left to cover inside the chosen lane: r - incl[L-1]; lane L forms its own
    float excl = dpp_f<DPP_ROW_SHR(1), 0xF, true>(0.0f, incl);
    const float thr = r - excl;
    // ---- level 2: which candidate of lane L
    const bool mine = __builtin_amdgcn_inverse_ballot_w64(row_first(m));
    if (mine) {
#pragma unroll
      for (int c = 0; c < CH; ++c) *(float4 *)(pk + 4 * c) = row[c];
      *(float2 *)(pk + 16) = make_float2(thr, __int_as_float(s));
    }
    __builtin_amdgcn_wave_barrier();
    const float cv = pk[s];
    const float2 tl = *(const float2 *)(pk + 16);
    const int mychoice = cbase + (__float_as_int(tl.y) << 2);
    const float sc = row_scan_add<S2>(cv);
    const uint64_t pos = __builtin_amdgcn_fcmpf(cv, 0.0f, FCMP16_OGT) & act & alive;   // (a dead row holds stale slots)
    uint64_t k = __builtin_amdgcn_fcmpf(sc, tl.x, FCMP16_OGE) & pos;
    if (__builtin_expect(some_row_empty(k), 0)) {
      // rounding: no candidate of a row reached thr -> that lane's last open candidate with p > 0
      // (also taken, harmlessly, while a row is finished or infeasible: its fields stay empty)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t ki = (uint32_t)(k >> (16 * i)) & 0xFFFFu, qi = (uint32_t)(pos >> (16 * i)) & 0xFFFFu;
        if (ki == 0 && qi != 0) k |= (uint64_t)(0x80000000u >> __builtin_clz(qi)) << (16 * i);
      }
    }
    const bool win = __builtin_amdgcn_inverse_ballot_w64(row_first(k));
    if (win) {
      pk[18] = __int_as_float(mychoice);
      if (!CVRP || mychoice != 0) fl[mychoice] = 0.0f;  // visited (the CVRP depot stays open)
    }
    __builtin_amdgcn_wave_barrier();
    // no feasible candidate (flagged; the reference raises): move to node 0 like the one-ant kernel and the oracle
    const int choice = S > 0.0f ? __float_as_int(pk[18]) : 0;
    __builtin_amdgcn_wave_barrier();
    if constexpr (!CVRP) { if (s == 0 && !(S > 0.0f)) fl[0] = 0.0f; }

    // ---- outputs: lane 0 of every row that is still building
    const bool writer = __builtin_amdgcn_inverse_ballot_w64(act & LEAD);
    if (writer) {
      path_a[(size_t)t * A] = choice;
      if constexpr (LOGP) {
        const float pc = *(const float *)(Pb + __umul24((uint32_t)prev, ldb) + (uint32_t)choice * 4u);
        logp_a[(size_t)(t - 1) * A] = clamp_log(pc / S);
        if (rs_a) rs_a[(size_t)(t - 1) * A] = S;
      }
      if (want_cost) {                                   // fused length, edge added one step late
        cost = cost + dpend;
        // TSP: d[u_t][u_{t-1}] (tsp/aco.py:127); CVRP: d[u_{t-1}][u_t] (cvrp/aco.py:135)
        const uint32_t e = CVRP ? __umul24((uint32_t)prev, (uint32_t)n) + (uint32_t)choice
                                : __umul24((uint32_t)choice, (uint32_t)n) + (uint32_t)prev;
        dpend = *(const float *)(dist_b + (e << 2));
      }
      if (want_tab) {
        if constexpr (CVRP) {
          // who follows `prev`; row 0 of the table is never read (the depot's successors are a set)
          *(uint32_t *)(tab_b + __umul24((uint32_t)prev, A4) + a4) = (uint32_t)choice << 16;
          __hip_atomic_fetch_or(hub_l + (choice >> 5), prev == 0 ? 1u << (choice & 31) : 0u, __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else {
          *(uint32_t *)(tab_b + __umul24((uint32_t)prev, A4) + a4) = (uint32_t)pprev | ((uint32_t)choice << 16);
        }
      }
    }
    if constexpr (CVRP) {                                // state of each row, as selects
      const bool live = !finished;
      const bool moved = live && choice != 0;
      remaining -= moved ? 1 : 0;
      const float load = moved ? used : 0.0f;            // back at the depot the load restarts from 0
      used = live ? load + dem_s[choice] : used;
      finished = finished || (remaining == 0 && choice == 0);
      len = live ? t + 1 : len;
      prev = finished ? 0 : choice;
      act = __builtin_amdgcn_ballot_w64(!finished);
    } else {
      pprev = prev;
      prev = choice;
    }
  }
  if (s == 0) {
    if constexpr (CVRP) {
      // the reference steps every ant until the slowest one is done: a done ant keeps drawing the
      // depot (probability 1), so its column is padded with 0 / log(1-eps)
      if (p.lens) p.lens[(size_t)b * A + a] = len;
      if (p.tab_lens) p.tab_lens[(size_t)b * A + a] = len;
      const float lp1 = clamp_log(1.0f);
      for (int tt = len; tt < p.Lmax; ++tt) {
        path_a[(size_t)tt * A] = 0;
        if constexpr (LOGP) logp_a[(size_t)(tt - 1) * A] = lp1;
      }
      if (!finished && p.flags) atomicOr(p.flags + b, 2);
      if (want_cost) p.costs[(size_t)b * A + a] = cost + dpend;
      if (want_tab) {
        uint32_t *hub_a = p.hubmask + ((size_t)b * A + a) * ((n + 31) >> 5);
        for (int i = 0; i < ((n + 31) >> 5); ++i) hub_a[i] = hub_l[i];
      }
    } else {
      if (want_cost) {
        cost = cost + dpend;
        cost = cost + *(const float *)(dist_b + ((__umul24((uint32_t)first, (uint32_t)n) + (uint32_t)prev) << 2));
        p.costs[(size_t)b * A + a] = cost;
      }
      if (want_tab) {                                   // close the cycle: last -> first -> second
        uint32_t *nbr_a = (uint32_t *)(tab_b + a4);     // + node * A
        const int second = (int)p.paths[((size_t)b * n + 1) * A + a];
        if (n == 2) { nbr_a[(size_t)first * A] = (uint32_t)prev | ((uint32_t)prev << 16); nbr_a[(size_t)prev * A] = (uint32_t)first | ((uint32_t)first << 16); }
        else { nbr_a[(size_t)prev * A] = (uint32_t)pprev | ((uint32_t)first << 16); nbr_a[(size_t)first * A] = (uint32_t)prev | ((uint32_t)second << 16); }
      }
    }
  }
  if (feasible != ~0ull && p.flags && lane == 0) atomicOr(p.flags + b, 1);
}

template <int CH, bool CVRP>
static hipError_t launch16(const SampleParams &sp, bool logp, hipStream_t s) {
  const int bpi = (sp.A + 15) / 16;
  dim3 grid((unsigned)(sp.B * bpi)), block(256);
  const bool fused = sp.costs && sp.nbr;
#define DACO_L16(L, F) hipLaunchKernelGGL((scan16_kernel<CH, L, F, CVRP>), grid, block, 0, s, sp)
  if (logp) { if (fused) DACO_L16(true, true); else DACO_L16(true, false); }
  else { if (fused) DACO_L16(false, true); else DACO_L16(false, false); }
#undef DACO_L16
  return hipGetLastError();
}

// entries used by daco_tsp_sample / daco_cvrp_sample for n <= DACO_SCAN16_MAX_N in DACO_SCAN mode
hipError_t launch_tsp_scan16(const SampleParams &sp, bool logp, hipStream_t s) {
  switch ((sp.n + 63) / 64) {
    case 1: return launch16<1, false>(sp, logp, s);
    case 2: return launch16<2, false>(sp, logp, s);
    case 3: return launch16<3, false>(sp, logp, s);
    default: return launch16<4, false>(sp, logp, s);
  }
}
hipError_t launch_cvrp_scan16(const SampleParams &sp, bool logp, hipStream_t s) {
  switch ((sp.n + 63) / 64) {
    case 1: return launch16<1, true>(sp, logp, s);
    case 2: return launch16<2, true>(sp, logp, s);
    case 3: return launch16<3, true>(sp, logp, s);
    default: return launch16<4, true>(sp, logp, s);
  }
}

}  // namespace daco
