#pragma once
// daco_scan16_kernel.h -- tour construction for small instances, prefix-scan draw, SEVERAL ants per wavefront:
// TSP (tsp/aco.py:134-177, tsp_nls/aco.py:184-220; n <= 256) and CVRP (cvrp/aco.py:138-205; n <= 512).
//
// scan16_kernel<LPA, CH, LOGP, CVRP>: LPA lanes per ant -- 4 for n <= 128 (sixteen ants per wavefront), 8 for n <= 256 (eight),
// 16 for CVRP with 256 < n <= 512 and behind DACO_SCAN_LAYOUT=16 (four: the layout of rounds 1-2).  At these sizes a row is at
// most 2 KiB and the step is bound by instruction issue and by the latency of its dependent chain, not by bytes: the per-step
// overhead (scan, compare, search, OR-reduce, flag and tour stores: ~55 instructions whatever the row length) is shared by the
// ants of a wavefront, an ant-step costs about 55 * LPA / 64 + 5 instructions.  Candidate k of an ant sits in
// lane s = (k/4) % LPA of its group, chunk c = k / (4 LPA).
// Same structure as tsp_scan32_kernel (daco_tsp_scan32.hip), which an ablation study motivated:
//   * level 1: DPP scan of the group's lane sums (Kogge-Stone; with several groups per 16-lane DPP row the steps that would
//     cross a group boundary add +0.0f), S by quad_perm / row_newbcast, the step's uniform from the group's cached Philox block
//     (LPA = 16: rotated through lane 15; else read through the LDS crossbar, ds_bpermute), the first lane with incl >= u*S per
//     group from the compare mask with bit arithmetic on its LPA-bit fields;
//   * level 2 INSIDE the chosen lane: every lane keeps the running sums of its own <= 32 masked candidates and the
//     chosen one finds its candidate by a binary search over them (count_below32); the choice reaches the lanes of
//     the group by a rotate-OR / quad-perm butterfly all-reduce -- nothing is handed through LDS;
//   * visited flags are f16 0/1 in LDS (the second operand of v_fma_mix_f32; CVRP applies the capacity and depot rules to
//     the f32 row value first), the tours stay in LDS (bytes for n <= 256) and paths / route costs / the update's table
//     leave the workgroup together in an epilogue (runs of 16 / 32 / 64 ants; the edge staging of the costs lives in the
//     flag array, dead by then).
// Draw semantics: the scan specification of DESIGN.md section 4 with `lanes` = LPA; the GPU tests hold every layout
// bit-exact against the CPU restatement of that specification (which honours DACO_SCAN_LAYOUT like the library).
#include <type_traits>
#include "daco_sample_kernel.h"

namespace daco {

constexpr int FCMP16_OGT = 2, FCMP16_OGE = 3;

// inclusive add-scan inside each group of LPA lanes (Kogge-Stone, DPP row_shr).  LPA = 8: lanes 8..15 of a DPP row are a
// group of their own -- a step of d lanes adds +0.0f where it would reach across the boundary (s < d), as the specification
// (oracle: lane_scan restricted to the first 8 lanes of a row) has it
template <int LPA>
__device__ inline float group_scan_add(float x, int s) {
  if constexpr (LPA == 16) {
    x = x + dpp_f<DPP_ROW_SHR(1), 0xF, true>(0.0f, x);
    x = x + dpp_f<DPP_ROW_SHR(2), 0xF, true>(0.0f, x);
    x = x + dpp_f<DPP_ROW_SHR(4), 0xF, true>(0.0f, x);
    x = x + dpp_f<DPP_ROW_SHR(8), 0xF, true>(0.0f, x);
  } else {
    float t = dpp_f<DPP_ROW_SHR(1), 0xF, true>(0.0f, x); x = x + (s >= 1 ? t : 0.0f);
    t = dpp_f<DPP_ROW_SHR(2), 0xF, true>(0.0f, x); x = x + (s >= 2 ? t : 0.0f);
    if constexpr (LPA == 8) { t = dpp_f<DPP_ROW_SHR(4), 0xF, true>(0.0f, x); x = x + (s >= 4 ? t : 0.0f); }
  }
  return x;
}
// lane N of each row to all lanes of the row (gfx90a+ DPP row_newbcast)
template <int N> __device__ inline float row_bcast(float x) { return dpp_f<0x150 + N, 0xF, false>(x, x); }
template <int N> __device__ inline int row_ror(int x) { return dpp_i<0x120 + N, 0xF, false>(x, x); }
// of the lanes set in m, the first one of every group of LPA lanes: per LPA-bit field x, x & ~((x | top) - 1)
template <int LPA>
__device__ inline uint64_t group_first(uint64_t m) {
  constexpr uint64_t TOP = LPA == 16 ? 0x8000800080008000ull : LPA == 8 ? 0x8080808080808080ull : 0x8888888888888888ull;
  constexpr uint64_t ONE = LPA == 16 ? 0x0001000100010001ull : LPA == 8 ? 0x0101010101010101ull : 0x1111111111111111ull;
  return m & ~((m | TOP) - ONE);
}
// last lane of the group to all of its lanes
template <int LPA>
__device__ inline float group_bcast_last(float x, int lane) {
  if constexpr (LPA == 16) return row_bcast<15>(x);
  if constexpr (LPA == 4) return dpp_f<0xFF /* quad_perm [3,3,3,3] */, 0xF, false>(x, x);
  const float lo = row_bcast<7>(x), hi = row_bcast<15>(x);
  return (lane & 8) ? hi : lo;
}
// OR over the lanes of the group, in every lane
template <int LPA>
__device__ inline int group_or(int x) {
  if constexpr (LPA == 16) {
    x |= row_ror<1>(x); x |= row_ror<2>(x); x |= row_ror<4>(x); x |= row_ror<8>(x);
  } else {
    x |= dpp_i<0xB1 /* quad_perm [1,0,3,2] */, 0xF, false>(x, x);
    x |= dpp_i<0x4E /* quad_perm [2,3,0,1] */, 0xF, false>(x, x);
    if constexpr (LPA == 8) x |= dpp_i<0x141 /* row_half_mirror */, 0xF, false>(x, x);
  }
  return x;
}

// CH: chunks of 64 candidates (n <= 64 * CH; CH <= 4 in production, TSP up to 8 = n <= 512 as a measured alternative
// to the two-ants-per-wavefront kernel).
template <int LPA, int CH, bool LOGP, bool CVRP, bool F64 = false>
__global__ void __launch_bounds__(256)   // (asking for five waves per SIMD at CVRP-100 -- 96 registers instead of 102 -- was measured: 0.546 -> 0.59 ms)
scan16_kernel(const SampleParams p, const int TL /* entries per tour buffer */) {
  static_assert(LPA == 16 || LPA == 8 || LPA == 4, "lanes per ant");
  constexpr int LG = LPA == 16 ? 4 : LPA == 8 ? 3 : 2;
  constexpr int APW = 64 / LPA, APB = 4 * APW;          // ants per wavefront / per workgroup
  constexpr int NJ = CH * 4;                            // candidates per lane
  constexpr int NG = (NJ + 7) / 8;                      // 16-byte flag groups per lane
  constexpr int ROWF = CH * LPA * 4;                    // padded row length of this layout
  constexpr int GS = LPA * 8;                           // flags of one 16-byte group across the ant's lanes
  constexpr int FL0 = (CH <= 4 ? 2 : 4) * GS;
  constexpr int FL = FL0 < 128 ? 128 : FL0;             // flag / inverse-table entries per ant (>= n; >= 128: the edge staging below)
  constexpr int CB = LPA * 16;                          // bytes of a row chunk
  static_assert(!CVRP || ROWF <= 512, "CVRP: n <= 512 (hub bitmap, demand row)");
  static_assert(LPA == 16 || NJ <= 32, "eight / sixteen ants per wavefront: up to 32 candidates per lane");
  static_assert(!F64 || CVRP, "float64 load bookkeeping is CVRP's (cvrp_nls/aco.py:254-272)");
  constexpr bool DEM_REGS = LPA == 16 && CH <= 4 && !F64;       // CVRP: the lane's demands stay in registers / are read from LDS every step
  // open[ant][g][lane][8]: f16 1.0 while the node in slot j = 8g + e of that lane is unvisited, else 0.0; slot
  // j = c*4 + v of lane s is node c*(4 LPA) + s*4 + v.  Reused as the inverse-permutation table in the epilogue.
  __shared__ __attribute__((aligned(16))) _Float16 open_flags[APB][FL];
  __shared__ __attribute__((aligned(16))) float dem_s[CVRP ? ROWF : 4];   // CVRP: demand, +inf padding
  __shared__ __attribute__((aligned(16))) double dem64_s[F64 ? ROWF : 2]; // cvrp_nls: the float64 demands (load bookkeeping in double)
  constexpr int HW = (ROWF + 31) / 32;
  __shared__ uint32_t hub_s[APB][HW];                   // CVRP: per ant, set of nodes that follow the depot (bitmap over the padded row)
  __shared__ int len_s[APB];                            // CVRP: rows used by each ant (0: slot holds no ant)
  // the tours: node ids as bytes when every id fits one (n <= 256) -- half the LDS of the workgroup's largest array, i.e. one
  // more workgroup per CU at CVRP-100 (routes of up to 2n + 1 entries)
  using tour_t = std::conditional_t<(ROWF <= 256), uint8_t, uint16_t>;
  extern __shared__ __attribute__((aligned(16))) unsigned char tour_raw[];
  tour_t *tour_s = reinterpret_cast<tour_t *>(tour_raw);   // [APB][TL]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane / LPA, s = lane & (LPA - 1);
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int bpi = (p.A + APB - 1) / APB;                // workgroups per instance (APB ants each)
  const int b = w / bpi;
  const int abase = (w - b * bpi) * APB;                // first ant of the workgroup
  const int a0 = abase + wave * APW;                    // ants a0 .. a0+APW-1, one per group of LPA lanes
  const int n = p.n, A = p.A, ld = p.ld;
  const uint64_t iter_now = p.iter + (p.iter_dev ? *p.iter_dev : 0ull);   // a captured graph advances *iter_dev
  if constexpr (CVRP) {
    for (int k = threadIdx.x; k < ROWF; k += 256) dem_s[k] = k < n ? p.demand[(size_t)b * n + k] : __builtin_inff();
    if constexpr (F64) for (int k = threadIdx.x; k < ROWF; k += 256) dem64_s[k] = k < n ? p.demand64[(size_t)b * n + k] : (double)__builtin_inff();
    for (int k = threadIdx.x; k < APB * HW; k += 256) (&hub_s[0][0])[k] = 0u;
    if (threadIdx.x < APB) len_s[threadIdx.x] = 0;
    __syncthreads();
  }
  const bool active = a0 < A;                           // (a wave without ants still joins the epilogue's barriers)
  // A not a multiple of APW: the spare groups build ant A-1 again (same counters, same tour; their copy is not written)
  const int a = a0 + q < A ? a0 + q : A - 1;
  const uint64_t LEAD = LPA == 16 ? 0x0001000100010001ull : LPA == 8 ? 0x0101010101010101ull : 0x1111111111111111ull;   // lane 0 of each group
  const uint32_t gid = p.ant_gid0 + (uint32_t)(b * (p.gid_bstride ? p.gid_bstride : A) + a);
  const char *Pb = (const char *)(p.P + (size_t)b * n * ld);           // uniform; lanes add 32-bit offsets
  const uint32_t ldb = (uint32_t)ld * 4u, lane_off = (uint32_t)s * 16u;
  const int rows = CVRP ? p.Lmax : n;                   // rows of paths for one instance
  float *logp_a = LOGP ? p.logp + (size_t)b * (rows - 1) * A + a : nullptr;
  float *rs_a = (LOGP && p.rowsum) ? p.rowsum + (size_t)b * (rows - 1) * A + a : nullptr;
  _Float16 *fl = open_flags[wave * APW + q];
  tour_t *tour = tour_s + (size_t)(wave * APW + q) * TL;
  // flag index of node k (chunk c = k / (4 LPA)): 16-byte group c >> 1, lane (k>>2) % LPA, element (c & 1)*4 + (k&3)
  auto flag_index = [](int k) {
    const int c = k / (LPA * 4);
    return (c >> 1) * GS + (((k >> 2) & (LPA - 1)) << 3) + ((c & 1) << 2) + (k & 3);
  };
  uint64_t feasible = ~0ull;
  bool finished = false;
  int len = 1;

  if (active) {
    float4 dm[DEM_REGS ? CH : 1];
    {
      const f16x8 ones = {1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
      for (int g = 0; g < FL / GS; ++g) *(f16x8 *)(fl + g * GS + s * 8) = ones;
#pragma unroll
      for (int c = 0; c < (DEM_REGS ? CH : 1); ++c) { if constexpr (CVRP && DEM_REGS) dm[c] = *(const float4 *)(dem_s + (c * LPA + s) * 4); else dm[c] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
    int prev;
    if constexpr (CVRP) prev = 0;
    else if (p.start) prev = (int)p.start[(size_t)b * A + a];
    else if (p.fixed_start >= 0) prev = p.fixed_start;
    else {
      const u32x4 r = rng_block(p.seed, iter_now, STREAM_START, gid, 0);
      prev = (int)__umulhi(r.x, (uint32_t)n);
    }
    __builtin_amdgcn_wave_barrier();
    if (s == 0) {
      if constexpr (!CVRP) fl[flag_index(prev)] = (_Float16)0.0f;        // the depot is never closed for good
      tour[0] = (tour_t)prev;
    }
    __builtin_amdgcn_wave_barrier();
    int remaining = n - 1;
    float used = CVRP ? 0.0f + dem_s[0] : 0.0f;
    double used64 = F64 ? 0.0 + dem64_s[0] : 0.0;
    finished = CVRP ? remaining == 0 : false;
    u32x4 ublk = {0, 0, 0, 0};                          // 4 LPA cached uniforms per ant
    float ucur = 0.0f;                                  // LPA = 16: rotated once per step, lane 15 holds the current step's uniform
    const int ubase = (lane & ~(LPA - 1)) << 2;         // LPA = 8: byte address of the group's lane 0 for ds_bpermute
    uint64_t act = CVRP ? __builtin_amdgcn_ballot_w64(!finished) : ~0ull;     // lanes of the rows still building
    const int tend = CVRP ? p.Lmax : n;
    const float *uin = (!CVRP && p.noise) ? p.noise + (size_t)b * (n - 1) * A + a : nullptr;

    for (int t = 1; t < tend && act != 0; ++t) {
      const uint32_t rowoff = __umul24((uint32_t)prev, ldb);
      const uint32_t voff = rowoff + lane_off;
      float4 row[CH];
      f16x8 fo[NG];
#pragma unroll
      for (int c = 0; c < CH; ++c) row[c] = *(const float4 *)(Pb + voff + c * CB);
      float u;
      if constexpr (LPA == 16) {
        // uniform of step t: component (t>>4)&3 of Philox block ((t>>6)<<4) + (t&15).  Lane s computes the one of step
        // (t & ~15) + 15 - s; after every step the row is rotated by one lane, so lane 15 always holds the current one
        if ((t & 15) == 0 || t == 1) {
          if ((t & 63) == 0 || t == 1) ublk = rng_block(p.seed, iter_now, STREAM_SCAN, gid, (uint32_t)(((t >> 6) << 4) + (15 - s)));
          ucur = u01(comp(ublk, (t >> 4) & 3));
          if (t == 1) ucur = __int_as_float(row_ror<1>(__float_as_int(ucur)));    // step 1 starts at element 1 of the block
        }
        u = row_bcast<15>(ucur);
        ucur = __int_as_float(row_ror<1>(__float_as_int(ucur)));
      } else {
        // uniform of step t (LPA = 8): component (t>>3)&3 of Philox block ((t>>5)<<3) + (t&7).  Lane s of the group holds the one of
        // step (t & ~7) + s; the group reads lane t & 7 through the LDS crossbar (no VALU slot; issued here, used after the scan)
        if ((t & (LPA - 1)) == 0 || t == 1) {
          if ((t & (4 * LPA - 1)) == 0 || t == 1) ublk = rng_block(p.seed, iter_now, STREAM_SCAN, gid, (uint32_t)(((t >> (LG + 2)) << LG) + s));
          ucur = u01(comp(ublk, (t >> LG) & 3));
        }
        u = __int_as_float(__builtin_amdgcn_ds_bpermute(ubase + ((t & (LPA - 1)) << 2), __float_as_int(ucur)));
      }
      if constexpr (!CVRP) { if (uin) u = uin[(size_t)(t - 1) * A]; }       // injected uniform stream (tests): [B][n-1][A]
#pragma unroll
      for (int g = 0; g < NG; ++g) fo[g] = *(const f16x8 *)(fl + g * GS + s * 8);

      // ---- the lane's running sums in slot order (closed slots add p*0 = +0.0f; the product with a 0/1 factor is exact)
      const float rem = CVRP ? p.capacity - used : 0.0f;
      const double rem64 = F64 ? p.capacity64 - used64 : 0.0;
      float run[32];
      float acc = 0.0f;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int e = (c & 1) * 4;
        const float rv[4] = {row[c].x, row[c].y, row[c].z, row[c].w};
        const float4 dmc = (CVRP && !DEM_REGS && !F64) ? *(const float4 *)(dem_s + (c * LPA + s) * 4) : dm[DEM_REGS ? c : 0];
        const float dv[4] = {dmc.x, dmc.y, dmc.z, dmc.w};
        double dv64[4] = {0.0, 0.0, 0.0, 0.0};
        if constexpr (F64) {
          const double2 d01 = *(const double2 *)(dem64_s + (c * LPA + s) * 4), d23 = *(const double2 *)(dem64_s + (c * LPA + s) * 4 + 2);
          dv64[0] = d01.x; dv64[1] = d01.y; dv64[2] = d23.x; dv64[3] = d23.y;
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          if constexpr (CVRP) {
            // the capacity and depot rules select on the f32 row value, the visited flag stays the f16 operand of the fma
            // (a blocked candidate adds 0*f = +0.0f like a visited one adds p*0)
            float pv = (F64 ? dv64[v] > rem64 : dv[v] > rem) ? 0.0f : rv[v];   // strict, cvrp/aco.py:200 (cvrp_nls: in double, :267-270)
            if (c == 0 && v == 0) pv = (s == 0 && prev == 0 && remaining > 0) ? 0.0f : pv;   // the depot, cvrp/aco.py:179
            acc = __builtin_fmaf(pv, (float)fo[c >> 1][e + v], acc);
          } else {
            acc = __builtin_fmaf(rv[v], (float)fo[c >> 1][e + v], acc);
          }
          run[4 * c + v] = acc;
        }
      }
      // ---- level 1: which lane
      const float part = acc;
      const float incl = group_scan_add<LPA>(part, s);
      const float S = group_bcast_last<LPA>(incl, lane);
      const float r = fmaxf(u * S, 1.401298464e-45f);     // keep r > 0 if u*S underflows
      const uint64_t m = __builtin_amdgcn_fcmpf(incl, r, FCMP16_OGE) & __builtin_amdgcn_fcmpf(part, 0.0f, FCMP16_OGT) & act;
      const uint64_t alive = __builtin_amdgcn_fcmpf(S, 0.0f, FCMP16_OGT);   // S > 0 <=> some open candidate has p > 0
      feasible &= alive | ~act;
      // ---- level 2 in every lane (only the chosen lane's result is used)
      float excl = dpp_f<DPP_ROW_SHR(1), 0xF, true>(0.0f, incl);
      if constexpr (LPA < 16) excl = s == 0 ? 0.0f : excl;      // (lane 8 of a DPP row starts a group)
      const float thr = fmaxf(r - excl, 1.401298464e-45f);
      int cnt = count_below32<NJ>(run, thr);
      const bool mine = __builtin_amdgcn_inverse_ballot_w64(group_first<LPA>(m));
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(mine && cnt >= NJ) != 0, 0)) {
        // rounding: no running sum reached thr -> the lane's last open candidate with p > 0 (rare: the row and the flags
        // are read again; "where the running sum reaches its final value" is not the same -- a term can be absorbed)
        int last = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const float4 rw = *(const float4 *)(Pb + voff + c * CB);
          const f16x8 ff = *(const f16x8 *)(fl + (c >> 1) * GS + s * 8);
          const int e = (c & 1) * 4;
          const float rv[4] = {rw.x, rw.y, rw.z, rw.w};
          const float4 dmc = (CVRP && !DEM_REGS && !F64) ? *(const float4 *)(dem_s + (c * LPA + s) * 4) : dm[DEM_REGS ? c : 0];
          const float dv[4] = {dmc.x, dmc.y, dmc.z, dmc.w};
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            float f = (float)ff[e + v];
            if constexpr (CVRP) {
              f = (F64 ? dem64_s[(c * LPA + s) * 4 + v] > rem64 : dv[v] > rem) ? 0.0f : f;
              if (c == 0 && v == 0) f = (s == 0 && prev == 0 && remaining > 0) ? 0.0f : f;
            }
            last = rv[v] * f > 0.0f ? 4 * c + v : last;
          }
        }
        cnt = cnt >= NJ ? last : cnt;
      }
      const int node = (cnt >> 2) * (LPA * 4) + (s << 2) + (cnt & 3);
      int x = group_or<LPA>(mine ? node + 1 : 0);
      // a row without a winner (no feasible candidate: flagged, the reference raises; or finished) moves to node 0
      const int choice = x ? x - 1 : 0;
      if (mine && (!CVRP || node != 0)) fl[(cnt >> 3) * GS + (s << 3) + (cnt & 7)] = (_Float16)0.0f;   // visited (the CVRP depot stays open)
      if constexpr (!CVRP) { if (s == 0 && !(S > 0.0f)) fl[flag_index(0)] = (_Float16)0.0f; }

      // ---- outputs: lane 0 of every row that is still building
      const bool writer = __builtin_amdgcn_inverse_ballot_w64(act & LEAD);
      if (writer) {
        tour[t] = (tour_t)choice;
        if constexpr (LOGP) {
          const float pc = *(const float *)(Pb + rowoff + (uint32_t)choice * 4u);
          logp_a[(size_t)(t - 1) * A] = clamp_log(pc / S);
          if (rs_a) rs_a[(size_t)(t - 1) * A] = S;
        }
      }
      asm volatile("" ::: "memory");                     // the next step's flag loads follow the stores above
      __builtin_amdgcn_wave_barrier();
      if constexpr (CVRP) {                                // state of each row, as selects
        const bool live = !finished;
        const bool moved = live && choice != 0;
        remaining -= moved ? 1 : 0;
        const float load = moved ? used : 0.0f;            // back at the depot the load restarts from 0
        used = live ? load + dem_s[choice] : used;
        if constexpr (F64) { const double load64 = moved ? used64 : 0.0; used64 = live ? load64 + dem64_s[choice] : used64; }
        finished = finished || (remaining == 0 && choice == 0);
        len = live ? t + 1 : len;
        prev = finished ? 0 : choice;
        act = __builtin_amdgcn_ballot_w64(!finished);
      } else {
        prev = choice;
      }
    }
    if constexpr (CVRP) { if (s == 0 && a0 + q < A) len_s[wave * APW + q] = len; }
  }
  if (feasible != ~0ull && p.flags && lane == 0) atomicOr(p.flags + b, 1);

  // ------------------------------------------------------------------ epilogue: the workgroup's 16 tours leave LDS
  __syncthreads();
  const int nant = A - abase < APB ? A - abase : APB;     // ants of this workgroup (the last one may hold fewer)
  const int k16 = threadIdx.x & (APB - 1);               // this thread's ant in the epilogue: APB lanes = one run per row
  constexpr int TSTEP = 256 / APB;
  {
    // paths[b][t][abase + k]: 16 lanes = one 128-byte run per step row; CVRP pads a finished route with the depot
    int64_t *pb = p.paths + (size_t)b * rows * A + abase;
    if (k16 < nant) {
      const int lk = CVRP ? len_s[k16] : n;
      for (int t = threadIdx.x / APB; t < rows; t += TSTEP) pb[(size_t)t * A + k16] = t < lk ? (int64_t)tour_s[(size_t)k16 * TL + t] : 0;
      if constexpr (CVRP && LOGP) {
        // the reference steps every ant until the slowest one is done: a done ant keeps drawing the depot
        // (probability 1), so its log-prob column is padded with log(1-eps)
        float *lp = p.logp + (size_t)b * (rows - 1) * A + abase;
        const float lp1 = clamp_log(1.0f);
        for (int t = threadIdx.x / APB; t < rows; t += TSTEP) if (t >= lk && t >= 1) lp[(size_t)(t - 1) * A + k16] = lp1;
      }
    }
  }
  if constexpr (CVRP) {
    if (threadIdx.x < nant) {
      const int a_ = abase + threadIdx.x;
      if (p.lens) p.lens[(size_t)b * A + a_] = len_s[threadIdx.x];
      if (p.tab_lens) p.tab_lens[(size_t)b * A + a_] = len_s[threadIdx.x];
    }
    if (active && s == 0 && a0 + q < A && !finished && p.flags) atomicOr(p.flags + b, 2);
  }
  if (p.costs) {
    // route / tour lengths: f32 sum in step order (TSP: d[u_t][u_{t-1}], tsp/aco.py:127, closing edge last; CVRP:
    // d[u_{t-1}][u_t], cvrp/aco.py:135).  64 edges of each of the wave's four ants are gathered with every lane active
    // and staged in LDS; lane 0 of each row adds its ant's 64 values one after the other.
    const float *dist_b = p.dist + (size_t)b * p.dist_bs;
    float (*dstage)[APW][64] = reinterpret_cast<float (*)[APW][64]>(&open_flags[0][0]);   // (the flags are dead; 256 B per ant)
    static_assert(sizeof(open_flags) >= sizeof(float) * 4 * APW * 64, "edge staging inside the flag array");
    if (active) {
      int lmax = n;
      if constexpr (CVRP) {
        lmax = 0;
#pragma unroll
        for (int r4 = 0; r4 < APW; ++r4) lmax = max(lmax, len_s[wave * APW + r4]);
      }
      const int myl = CVRP ? len_s[wave * APW + q] : n;
      float cost = 0.0f;
      const float *mine_d = dstage[wave][q];
      for (int base = 1; base < lmax; base += 64) {
        const int t = base + lane;
#pragma unroll
        for (int r4 = 0; r4 < APW; ++r4) {
          const tour_t *tr = tour_s + (size_t)(wave * APW + r4) * TL;
          const int lr = CVRP ? len_s[wave * APW + r4] : n;
          float dv = 0.0f;
          if (t < lr) dv = CVRP ? dist_b[(uint32_t)tr[t - 1] * (uint32_t)n + tr[t]] : dist_b[(uint32_t)tr[t] * (uint32_t)n + tr[t - 1]];
          dstage[wave][r4][lane] = dv;
        }
        __builtin_amdgcn_wave_barrier();
        if (s == 0 && base < myl) {
#pragma unroll
          for (int v4 = 0; v4 < 16; ++v4) {
            const float4 v = *(const float4 *)(mine_d + 4 * v4);   // (slots past the route's end hold +0.0f)
            cost = cost + v.x; cost = cost + v.y; cost = cost + v.z; cost = cost + v.w;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (s == 0 && a0 + q < A) {
        if constexpr (!CVRP) cost = cost + dist_b[(uint32_t)tour[0] * (uint32_t)n + tour[n - 1]];
        p.costs[(size_t)b * A + a0 + q] = cost;
      }
    }
  }
  if (p.nbr) {
    // the update's table: invert the tours in LDS (the flag array is free now), then 16 lanes write one 64-byte run per
    // node row.  TSP: nbr[node][ant] = prev | next << 16.  CVRP: successor << 16 (customers are visited once; row 0 is
    // never read -- the depot's successors are a SET, kept as a bitmap per ant).
    __syncthreads();
    uint16_t (*inv)[FL] = reinterpret_cast<uint16_t (*)[FL]>(open_flags);
    for (int e = threadIdx.x; e < APB * FL / 8; e += 256) ((uint4 *)&inv[0][0])[e] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (k16 < nant) {
      const int lk = CVRP ? len_s[k16] : n;
      const tour_t *tk = tour_s + (size_t)k16 * TL;
      for (int t = threadIdx.x / APB; t < lk; t += TSTEP) {
        const int v = tk[t];
        if (!CVRP || v != 0) inv[k16][v] = (uint16_t)t;
        if constexpr (CVRP) { if (t >= 1 && tk[t - 1] == 0 && v != 0) atomicOr(&hub_s[k16][v >> 5], 1u << (v & 31)); }
      }
    }
    __syncthreads();
    uint32_t *nb = p.nbr + (size_t)b * n * A + abase;
    if (k16 < nant) {
      const tour_t *tk = tour_s + (size_t)k16 * TL;
      for (int node = threadIdx.x / APB; node < n; node += TSTEP) {
        const int t = inv[k16][node];
        if constexpr (CVRP) {
          if (node != 0) nb[(size_t)node * A + k16] = (uint32_t)tk[t + 1] << 16;
        } else {
          const uint32_t pv = tk[t == 0 ? n - 1 : t - 1], nx = tk[t == n - 1 ? 0 : t + 1];
          nb[(size_t)node * A + k16] = pv | (nx << 16);
        }
      }
      if constexpr (CVRP) {
        const int W32 = (n + 31) >> 5;
        uint32_t *hub_a = p.hubmask + ((size_t)b * A + abase + k16) * W32;
        for (int i = threadIdx.x / APB; i < W32; i += TSTEP) hub_a[i] = hub_s[k16][i];
      }
    }
  }
}

template <int LPA, int CH, bool CVRP, bool F64>
static hipError_t launch16(const SampleParams &sp, bool logp, hipStream_t s) {
  constexpr int APB = 4 * (64 / LPA);
  const int bpi = (sp.A + APB - 1) / APB;
  dim3 grid((unsigned)(sp.B * bpi)), block(256);
  const int rows = CVRP ? sp.Lmax : sp.n;
  const int TL = (rows + 7) & ~7;
  const size_t dyn = (size_t)APB * TL * (CH * LPA * 4 <= 256 ? 1 : 2);
  if (logp) hipLaunchKernelGGL((scan16_kernel<LPA, CH, true, CVRP, F64>), grid, block, dyn, s, sp, TL);
  else hipLaunchKernelGGL((scan16_kernel<LPA, CH, false, CVRP, F64>), grid, block, dyn, s, sp, TL);
  return hipGetLastError();
}

template <int LPA, bool CVRP, bool F64>
static hipError_t launch_by_chunks(const SampleParams &sp, bool logp, hipStream_t s) {
  constexpr int W = LPA * 4;                            // candidates per chunk
  constexpr int MAXCH = 8;
  const int ch = (sp.n + W - 1) / W;
  if (ch > MAXCH) return hipErrorInvalidValue;
  switch (ch) {
    case 1: return launch16<LPA, 1, CVRP, F64>(sp, logp, s);
    case 2: return launch16<LPA, 2, CVRP, F64>(sp, logp, s);
    case 3: return launch16<LPA, 3, CVRP, F64>(sp, logp, s);
    case 4: return launch16<LPA, 4, CVRP, F64>(sp, logp, s);
    default: break;
  }
  if constexpr (MAXCH == 8) {
    switch (ch) {
      case 5: return launch16<LPA, 5, CVRP, F64>(sp, logp, s);
      case 6: return launch16<LPA, 6, CVRP, F64>(sp, logp, s);
      case 7: return launch16<LPA, 7, CVRP, F64>(sp, logp, s);
      default: return launch16<LPA, 8, CVRP, F64>(sp, logp, s);
    }
  }
  return hipErrorInvalidValue;
}

// lanes per ant by scan_small_lanes() -- 4 up to DACO_SCAN4_MAX_N nodes, 8 up to DACO_SCAN16_MAX_N, 16 above (CVRP) and behind
// DACO_SCAN_LAYOUT=16
template <bool CVRP, bool F64>
static hipError_t launch_by_lanes(const SampleParams &sp, bool logp, hipStream_t s) {
  switch (scan_small_lanes(sp.n)) {
    case 4: return launch_by_chunks<4, CVRP, F64>(sp, logp, s);
    case 8: return launch_by_chunks<8, CVRP, F64>(sp, logp, s);
    default: return launch_by_chunks<16, CVRP, F64>(sp, logp, s);
  }
}

}  // namespace daco
