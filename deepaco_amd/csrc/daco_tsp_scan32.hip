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
#include <cstring>

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

// What a step costs was measured piece by piece (tools/scan32_ablate.hip, profiles/r02_scan32_ablation.txt): the
// row stream itself runs at the L2 ceiling, and everything that made the first version 1.9x slower than that was
// per-step small traffic -- an i64 path store, a 4-byte distance gather and a 4-byte neighbour-table store per ant
// and step (2 active lanes per instruction), plus the LDS hand-over of the chosen lane's candidates.  So:
//   * level 2 runs INSIDE the chosen lane: every lane keeps the running sums of its own NJ masked candidates
//     (sequential f32 in slot order: the lane sum is the last of them), and after level 1 the chosen lane finds its
//     candidate by a binary search over those sums.  Per-ant values (row sum, chosen lane, choice) are SGPRs
//     (v_readlane / s_ff1), selected per half with one v_cndmask -- nothing is handed through LDS;
//   * the visited flags are f16 0/1 in LDS (v_fma_mix_f32 multiplies the f32 weight by the f16 flag: still exact,
//     one rounding), laid out so a lane reads its 16 flags with two 16-byte loads;
//   * the tour stays in LDS (u16) while it is built.  When the workgroup's 8 tours are complete it writes them out
//     together: paths as 64-byte runs (8 ants x i64) per step row, the neighbour table as 32-byte runs per node row
//     through an inverse-permutation table in LDS, and the tour lengths from 64-edge gathers with all lanes active
//     (summed in step order by one lane per ant, as the reference's sum).
// The step loop touches memory only for the row.
template <int CH2, bool LOGP>
__global__ void __launch_bounds__(256)
tsp_scan32_kernel(const SampleParams p) {
  constexpr int NJ = CH2 * 4;                           // candidates per lane (<= 32)
  constexpr int NG = (NJ + 7) / 8;                      // 16-byte flag groups per lane
  constexpr int FL = CH2 <= 4 ? 512 : 1024;             // flag / tour / inverse-table entries per ant
  static_assert(CH2 >= 1 && CH2 <= 8, "two ants per wavefront: n <= 1024");
  // open[h][g][lane][8]: f16 1.0 while the node in slot j = 8g + e of that lane is unvisited by ant h, else 0.0.
  // Slot j = c*4 + v of lane s is node c*128 + s*4 + v.  Reused as the inverse-permutation table in the epilogue.
  __shared__ __attribute__((aligned(16))) _Float16 open_flags[8][FL];
  __shared__ __attribute__((aligned(16))) uint16_t tour_s[8][FL];    // tour_s[h][t] = node visited at step t
  __shared__ __attribute__((aligned(16))) float dstage[4][2][64];    // epilogue: edge lengths of one 64-step chunk
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int up = lane >> 5, s = lane & 31;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int bpi = (p.A + 7) >> 3;                       // workgroups per instance (8 ants each)
  const int b = w / bpi;
  const int abase = (w - b * bpi) * 8;                  // first ant of the workgroup
  const int a0 = abase + wave * 2;                      // ants a0 (lower half), a0+1 (upper half)
  const int n = p.n, A = p.A, ld = p.ld;
  const bool active = a0 < A;                           // (a wave without ants still joins the epilogue's barriers)
  const uint64_t iter_now = p.iter + (p.iter_dev ? *p.iter_dev : 0ull);   // a captured graph advances *iter_dev
  // odd A: the last upper half builds ant A-1 a second time (same counters, same tour; its copy is not written)
  const int a = a0 + up < A ? a0 + up : A - 1;
  const bool lead = __builtin_amdgcn_inverse_ballot_w64(0x0000000100000001ull);   // lane 0 of each half
  const bool upper = __builtin_amdgcn_inverse_ballot_w64(0xFFFFFFFF00000000ull);  // per-half select mask
  const uint32_t gid = p.ant_gid0 + (uint32_t)(b * (p.gid_bstride ? p.gid_bstride : A) + a);
  const char *Pb = (const char *)(p.P + (size_t)b * n * ld);           // uniform; lanes add 32-bit offsets
  const uint32_t ldb = (uint32_t)ld * 4u, lane_off = (uint32_t)s * 16u;
  const uint32_t a4 = (uint32_t)a * 4u;
  char *logp_t = LOGP ? (char *)(p.logp + (size_t)b * (n - 1) * A) : nullptr;
  char *rs_t = (LOGP && p.rowsum) ? (char *)(p.rowsum + (size_t)b * (n - 1) * A) : nullptr;
  _Float16 *fl = open_flags[wave * 2 + up];
  uint16_t *tour = tour_s[wave * 2 + up];
  // flag index of node k: group (k>>8), lane (k>>2)&31, element ((k>>7)&1)*4 + (k&3)
  auto flag_index = [](int k) { return ((k >> 8) << 8) | (((k >> 2) & 31) << 3) | (((k >> 7) & 1) << 2) | (k & 3); };
  bool feasible = true;

  if (active) {
    const f16x8 ones = {1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
    for (int g = 0; g < FL / 256; ++g) *(f16x8 *)(fl + g * 256 + s * 8) = ones;
    int prev;
    if (p.start) prev = (int)p.start[(size_t)b * A + a];
    else if (p.fixed_start >= 0) prev = p.fixed_start;
    else {
      const u32x4 r = rng_block(p.seed, iter_now, STREAM_START, gid, 0);
      prev = (int)__umulhi(r.x, (uint32_t)n);
    }
    __builtin_amdgcn_wave_barrier();
    if (lead) {
      fl[flag_index(prev)] = (_Float16)0.0f;
      tour[0] = (uint16_t)prev;
    }
    __builtin_amdgcn_wave_barrier();
    u32x4 ublk = {0, 0, 0, 0};                          // 128 cached uniforms per ant (lane s: block base+s)
    const float *uin = p.noise ? p.noise + (size_t)b * (n - 1) * A + a : nullptr;

    for (int tb = 0; tb < n; tb += 32) {
      // uniform of step t: lane (t&31), component (t>>5)&3 of Philox block ((t>>7)<<5) + lane
      if ((tb & 127) == 0) ublk = rng_block(p.seed, iter_now, STREAM_SCAN, gid, (uint32_t)(((tb >> 7) << 5) + s));
      const int ucur = __float_as_int(u01(comp(ublk, (tb >> 5) & 3)));
      const int i1 = n - tb < 32 ? n - tb : 32;
      for (int i = tb == 0 ? 1 : 0; i < i1; ++i) {
        const uint32_t rowoff = __umul24((uint32_t)prev, ldb);
        const uint32_t voff = rowoff + lane_off;
        float4 row[CH2];
        f16x8 fo[NG];
#pragma unroll
        for (int c = 0; c < CH2; ++c) row[c] = *(const float4 *)(Pb + voff + c * 512);
        const int u_lo = __builtin_amdgcn_readlane(ucur, i), u_hi = __builtin_amdgcn_readlane(ucur, i + 32);
        float u = __int_as_float(upper ? u_hi : u_lo);
        if (uin) u = uin[(size_t)(tb + i - 1) * A];         // injected uniform stream (tests): [B][n-1][A]
#pragma unroll
        for (int g = 0; g < NG; ++g) fo[g] = *(const f16x8 *)(fl + g * 256 + s * 8);

        // ---- the lane's running sums in slot order.  A closed slot adds p*0 = +0.0f; the product with the
        // 0/1 flag is exact, so each fma rounds once like an add.
        float run[32];
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < CH2; ++c) {
          const int e = (c & 1) * 4;
          acc = __builtin_fmaf(row[c].x, (float)fo[c >> 1][e + 0], acc); run[4 * c + 0] = acc;
          acc = __builtin_fmaf(row[c].y, (float)fo[c >> 1][e + 1], acc); run[4 * c + 1] = acc;
          acc = __builtin_fmaf(row[c].z, (float)fo[c >> 1][e + 2], acc); run[4 * c + 2] = acc;
          acc = __builtin_fmaf(row[c].w, (float)fo[c >> 1][e + 3], acc); run[4 * c + 3] = acc;
        }
        // ---- level 1: which lane
        const float part = acc;
        const float incl = half_scan_add<true>(part);
        const int incl_i = __float_as_int(incl);
        const float S0 = __int_as_float(__builtin_amdgcn_readlane(incl_i, 31));
        const float S1 = __int_as_float(__builtin_amdgcn_readlane(incl_i, 63));
        const float S = upper ? S1 : S0;
        const float r = fmaxf(u * S, 1.401298464e-45f);   // keep r > 0 if u*S underflows
        const uint64_t m = __builtin_amdgcn_fcmpf(incl, r, FCMP_OGE) & __builtin_amdgcn_fcmpf(part, 0.0f, FCMP_OGT);
        const bool alive0 = S0 > 0.0f, alive1 = S1 > 0.0f;                 // S > 0 <=> some open candidate has p > 0
        feasible = feasible && alive0 && alive1;
        const uint32_t m0 = (uint32_t)m, m1 = (uint32_t)(m >> 32);
        const int L0 = m0 ? __builtin_ctz(m0) : 0, L1 = m1 ? __builtin_ctz(m1) : 0;     // chosen lane of each half
        // ---- level 2, in every lane (only lane L's result is read): what is left to cover inside the lane is
        // r - incl[L-1]; the candidate is the first slot whose running sum reaches it (such a slot is open with
        // p > 0; the threshold is kept > 0 so that a slot is "reached" only by a positive term)
        float excl = dpp_f<0x138 /* wave_shr:1 */, 0xF, true>(0.0f, incl);
        excl = s == 0 ? 0.0f : excl;
        const float thr = fmaxf(r - excl, 1.401298464e-45f);
        const int cnt = count_below32<NJ>(run, thr);
        int j0 = __builtin_amdgcn_readlane(cnt, L0), j1 = __builtin_amdgcn_readlane(cnt, L1 + 32);
        if (__builtin_expect(j0 >= NJ || j1 >= NJ, 0)) {
          // rounding: no running sum reached thr -> the lane's last open candidate with p > 0.  (Not "where the
          // running sum reaches its final value": a positive term can be absorbed by the sum before it -- a randomised
          // soak found that difference once in 6000 launches.)  Rare: the row and the flags are simply read again.
          int last = 0;
#pragma unroll
          for (int c = 0; c < CH2; ++c) {
            const float4 rw = *(const float4 *)(Pb + voff + c * 512);
            const f16x8 ff = *(const f16x8 *)(fl + (c >> 1) * 256 + s * 8);
            const int e = (c & 1) * 4;
            last = rw.x * (float)ff[e + 0] > 0.0f ? 4 * c + 0 : last;
            last = rw.y * (float)ff[e + 1] > 0.0f ? 4 * c + 1 : last;
            last = rw.z * (float)ff[e + 2] > 0.0f ? 4 * c + 2 : last;
            last = rw.w * (float)ff[e + 3] > 0.0f ? 4 * c + 3 : last;
          }
          if (j0 >= NJ) j0 = __builtin_amdgcn_readlane(last, L0);
          if (j1 >= NJ) j1 = __builtin_amdgcn_readlane(last, L1 + 32);
        }
        // slot j of lane L is node (j>>2)*128 + L*4 + (j&3), flag ((j>>3)<<8) | L<<3 | (j&7).
        // No feasible candidate (flagged; the reference raises): move to node 0 like the one-ant kernel and the oracle
        const int c0 = alive0 ? (((j0 >> 2) << 7) | (j0 & 3)) + (L0 << 2) : 0;
        const int c1 = alive1 ? (((j1 >> 2) << 7) | (j1 & 3)) + (L1 << 2) : 0;
        const int f0 = alive0 ? ((j0 >> 3) << 8) | (L0 << 3) | (j0 & 7) : 0;
        const int f1 = alive1 ? ((j1 >> 3) << 8) | (L1 << 3) | (j1 & 7) : 0;
        const int choice = upper ? c1 : c0;
        if (lead) {
          fl[upper ? f1 : f0] = (_Float16)0.0f;            // visited
          tour[tb + i] = (uint16_t)choice;
          if constexpr (LOGP) {
            const float pc = *(const float *)(Pb + rowoff + (uint32_t)choice * 4u);
            *(float *)(logp_t + a4) = clamp_log(pc / S);
            logp_t += (size_t)A * 4;
            if (rs_t) { *(float *)(rs_t + a4) = S; rs_t += (size_t)A * 4; }
          }
        }
        // the next step's flag loads must follow this store (same wave: the LDS executes them in program order);
        // the compiler barrier keeps the program order
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        prev = choice;
      }
    }
  }
  if (!feasible && p.flags && lane == 0) atomicOr(p.flags + b, 1);

  // ------------------------------------------------------------------ epilogue: the workgroup's 8 tours leave LDS
  __syncthreads();
  const int nant = A - abase < 8 ? A - abase : 8;        // ants of this workgroup (the last one may hold fewer)
  {
    // paths[b][t][abase + k]: 8 lanes = one 64-byte run per step row
    int64_t *pb = p.paths + (size_t)b * n * A + abase;
    const int k = threadIdx.x & 7;
    if (k < nant && p.paths)
      for (int t = threadIdx.x >> 3; t < n; t += 32) pb[(size_t)t * A + k] = (int64_t)tour_s[k][t];
  }
  if (p.costs) {
    // tour lengths (tsp/aco.py:121-132): sum_k d[u_k][u_{k-1}], k = 1..n-1, then the closing edge -- f32, that order.
    // 64 edges of each of the wave's two ants are gathered with every lane active and staged in LDS; lanes 0 and 32
    // add their ant's 64 values one after the other.
    const float *dist_b = p.dist + (size_t)b * p.dist_bs;
    const uint16_t *t0 = tour_s[wave * 2], *t1 = tour_s[wave * 2 + 1];
    const float *mine = dstage[wave][up];
    float cost = 0.0f;
    if (active) {
      // every gather of the two tours is issued before the first one is used (one trip to the L2 instead of one per
      // 64-edge chunk: the chunks' trips in sequence were 0.08 ms of the launch), the closing edges included
      constexpr int NCH = (FL + 63) / 64;
      float d0[NCH], d1[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int t = 1 + c * 64 + lane;
        d0[c] = 0.0f; d1[c] = 0.0f;
        if (t < n) {
          d0[c] = dist_b[(uint32_t)t0[t] * (uint32_t)n + t0[t - 1]];
          d1[c] = dist_b[(uint32_t)t1[t] * (uint32_t)n + t1[t - 1]];
        }
      }
      const uint16_t *tm = up ? t1 : t0;
      const float closing = dist_b[(uint32_t)tm[0] * (uint32_t)n + tm[n - 1]];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (1 + c * 64 < n) {                               // uniform
          dstage[wave][0][lane] = d0[c];
          dstage[wave][1][lane] = d1[c];
          __builtin_amdgcn_wave_barrier();
          if (lead) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const float4 v = *(const float4 *)(mine + 4 * q);      // (slots past the tour's end hold +0.0f)
              cost = cost + v.x; cost = cost + v.y; cost = cost + v.z; cost = cost + v.w;
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
      if (lead && a0 + up < A) {
        cost = cost + closing;
        p.costs[(size_t)b * A + a0 + up] = cost;
      }
    }
  }
  if (p.nbr) {
    // neighbour table nbr[b][node][ant] = prev | next << 16 (what the pheromone update consumes): invert the tours in
    // LDS (the flag array is free now), then 8 lanes write one 32-byte run per node row
    __syncthreads();                                     // every wave is done with its flags
    uint16_t (*inv)[FL] = reinterpret_cast<uint16_t (*)[FL]>(open_flags);
    for (int e = threadIdx.x; e < 8 * FL / 8; e += 256) ((uint4 *)&inv[0][0])[e] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const int k = threadIdx.x & 7;
    if (k < nant)                                         // (the other slots hold no tour: their entries are not nodes)
      for (int t = threadIdx.x >> 3; t < n; t += 32) inv[k][tour_s[k][t]] = (uint16_t)t;
    __syncthreads();
    uint32_t *nb = p.nbr + (size_t)b * n * A + abase;
    if (k < nant)
      for (int node = threadIdx.x >> 3; node < n; node += 32) {
        const int t = inv[k][node];
        const uint32_t pv = tour_s[k][t == 0 ? n - 1 : t - 1], nx = tour_s[k][t == n - 1 ? 0 : t + 1];
        nb[(size_t)node * A + k] = pv | (nx << 16);
      }
  }
}

template <int CH2>
static hipError_t launch32(const SampleParams &sp, bool logp, hipStream_t s) {
  const int bpi = (sp.A + 7) / 8;
  dim3 grid((unsigned)(sp.B * bpi)), block(256);
  static const int pad = getenv("DACO_SCAN32_LDS_PAD") ? atoi(getenv("DACO_SCAN32_LDS_PAD")) : 0;
  if (logp) hipLaunchKernelGGL((tsp_scan32_kernel<CH2, true>), grid, block, pad, s, sp);
  else hipLaunchKernelGGL((tsp_scan32_kernel<CH2, false>), grid, block, pad, s, sp);
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
  const bool upper = __builtin_amdgcn_inverse_ballot_w64(0xFFFFFFFF00000000ull);
  const uint32_t gid = p.ant_gid0 + (uint32_t)(b * (p.gid_bstride ? p.gid_bstride : A) + a);
  const char *Pb = (const char *)(p.P + (size_t)b * n * ld);
  const uint32_t ldb = (uint32_t)ld * 4u, lane_off = (uint32_t)s * 16u;
  int64_t *path_a = p.paths + (size_t)b * Lmax * A + a;
  float *logp_a = LOGP ? p.logp + (size_t)b * (Lmax - 1) * A + a : nullptr;
  float *rs_a = (LOGP && p.rowsum) ? p.rowsum + (size_t)b * (Lmax - 1) * A + a : nullptr;
  float *fl = open_flags[wave * 2 + up];
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
    const int ucur_i = __float_as_int(ucur);
    const int u_lo = __builtin_amdgcn_readlane(ucur_i, t & 31), u_hi = __builtin_amdgcn_readlane(ucur_i, (t & 31) + 32);
    const float u = __int_as_float(upper ? u_hi : u_lo);
#pragma unroll
    for (int c = 0; c < CH2; ++c) fo[c] = *(const float4 *)(fl + (c * 32 + s) * 4);
    const float rem = p.capacity - used;
    // the lane's running sums in slot order; a closed slot (visited, over capacity, or the depot while standing on
    // it) adds p*0 = +0.0f
    float run[16];
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < CH2; ++c) {
      float4 f = fo[c];
      f.x = dm[c].x > rem ? 0.0f : f.x;  f.y = dm[c].y > rem ? 0.0f : f.y;
      f.z = dm[c].z > rem ? 0.0f : f.z;  f.w = dm[c].w > rem ? 0.0f : f.w;
      if (c == 0) f.x = (s == 0 && prev == 0 && remaining > 0) ? 0.0f : f.x;       // the depot, cvrp/aco.py:179
      acc = __builtin_fmaf(row[c].x, f.x, acc); run[4 * c + 0] = acc;
      acc = __builtin_fmaf(row[c].y, f.y, acc); run[4 * c + 1] = acc;
      acc = __builtin_fmaf(row[c].z, f.z, acc); run[4 * c + 2] = acc;
      acc = __builtin_fmaf(row[c].w, f.w, acc); run[4 * c + 3] = acc;
    }
    const float part = acc;
    const float incl = half_scan_add<true>(part);
    const int incl_i = __float_as_int(incl);
    const float S0 = __int_as_float(__builtin_amdgcn_readlane(incl_i, 31)), S1 = __int_as_float(__builtin_amdgcn_readlane(incl_i, 63));
    const float S = upper ? S1 : S0;
    const float r = fmaxf(u * S, 1.401298464e-45f);
    const uint64_t m = __builtin_amdgcn_fcmpf(incl, r, FCMP_OGE) & __builtin_amdgcn_fcmpf(part, 0.0f, FCMP_OGT) & act;
    feasible &= __builtin_amdgcn_fcmpf(S, 0.0f, FCMP_OGT) | ~act;
    float excl = dpp_f<0x138 /* wave_shr:1 */, 0xF, true>(0.0f, incl);
    excl = s == 0 ? 0.0f : excl;
    const float thr = fmaxf(r - excl, 1.401298464e-45f);
    const uint32_t m0 = (uint32_t)m, m1 = (uint32_t)(m >> 32);
    const int L0 = m0 ? __builtin_ctz(m0) : 0, L1 = m1 ? __builtin_ctz(m1) : 0;         // chosen lane of each half
    // level 2 inside the chosen lane: first slot whose running sum reaches thr (see tsp_scan32_kernel)
    const int cnt = count_below<NJ>(run, thr);
    int j0 = __builtin_amdgcn_readlane(cnt, L0), j1 = __builtin_amdgcn_readlane(cnt, L1 + 32);
    if (__builtin_expect(j0 >= NJ || j1 >= NJ, 0)) {
      // rounding: no running sum reached thr -> the lane's last open candidate with p > 0 (row and flags read again)
      int last = 0;
#pragma unroll
      for (int c = 0; c < CH2; ++c) {
        const float4 rw = *(const float4 *)(Pb + voff + c * 512);
        float4 f = *(const float4 *)(fl + (c * 32 + s) * 4);
        f.x = dm[c].x > rem ? 0.0f : f.x;  f.y = dm[c].y > rem ? 0.0f : f.y;
        f.z = dm[c].z > rem ? 0.0f : f.z;  f.w = dm[c].w > rem ? 0.0f : f.w;
        if (c == 0) f.x = (s == 0 && prev == 0 && remaining > 0) ? 0.0f : f.x;
        last = rw.x * f.x > 0.0f ? 4 * c + 0 : last;
        last = rw.y * f.y > 0.0f ? 4 * c + 1 : last;
        last = rw.z * f.z > 0.0f ? 4 * c + 2 : last;
        last = rw.w * f.w > 0.0f ? 4 * c + 3 : last;
      }
      if (j0 >= NJ) j0 = __builtin_amdgcn_readlane(last, L0);
      if (j1 >= NJ) j1 = __builtin_amdgcn_readlane(last, L1 + 32);
    }
    // a half without a feasible candidate (flagged) or already finished moves to the depot
    const int c0 = m0 ? (((j0 >> 2) << 7) | (j0 & 3)) + (L0 << 2) : 0;
    const int c1 = m1 ? (((j1 >> 2) << 7) | (j1 & 3)) + (L1 << 2) : 0;
    const int choice = upper ? c1 : c0;
    if (lead && choice != 0) fl[choice] = 0.0f;          // customers are visited once, the depot stays open
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
    default: return launch32<8>(sp, logp, s);        // (n <= 1024)
  }
}

}  // namespace daco
