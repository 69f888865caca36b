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
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// ---- per-half plumbing of the step loop.  The kernel is bound by VALU issue (profiles/r03_pmc_headline.txt: the
// pipes busy 0.88 of the launch), so a value that is uniform inside each 32-lane half travels as two SGPRs and reaches
// the lanes with the upper half's lanes switched on and off by the scalar unit, instead of two v_mov + v_cndmask.
// Both helpers require EXEC = all lanes (the step loop runs under wave-uniform control flow only) and restore it.
// lanes 0..31: lo, lanes 32..63: hi
__device__ inline int half_pick(int lo, int hi) {
  int v;
  asm volatile("v_mov_b32 %0, %1\n\ts_mov_b32 exec_lo, 0\n\tv_mov_b32 %0, %2\n\ts_mov_b32 exec_lo, -1"
               : "=&v"(v) : "s"(lo), "s"(hi));
  return v;
}
// x * (lanes 0..31: lo, lanes 32..63: hi)
__device__ inline float half_mul(float x, float lo, float hi) {
  float v;
  asm volatile("v_mul_f32 %0, %2, %1\n\ts_mov_b32 exec_lo, 0\n\tv_mul_f32 %0, %3, %1\n\ts_mov_b32 exec_lo, -1"
               : "=&v"(v) : "v"(x), "s"(lo), "s"(hi));
  return v;
}
// r - incl[lane - 1] inside each half (r for the half's first lane): the subtraction reads its operand through DPP
// wave_shr:1 (lane 0 reads 0); lane 32 would see lane 31, the other ant, and is put right by one select.
// s_nop 1: two wait states between a VALU write of incl and the DPP read, whatever the scheduler placed before.
__device__ inline float half_excl_sub(float incl, float r, bool lane32) {
  float t;
  asm("s_nop 1\n\tv_subrev_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(t) : "v"(incl), "v"(r));
  return lane32 ? r : t;
}
// max(x, denorm_min) for x that is not a signalling NaN (fmaxf would spend a second v_max on quieting it)
__device__ inline float at_least_denorm(float x) {
  float y;
  asm("v_max_f32 %0, 1, %1" : "=v"(y) : "v"(x));
  return y;
}
// lowest set bit of a scalar, 0 if there is none (s_ff1 gives -1 there)
__device__ inline int first_bit_or_0(uint32_t x) {
  int l;
  asm("s_ff1_i32_b32 %0, %1" : "=s"(l) : "s"(x));
  return l < 0 ? 0 : l;
}

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
  constexpr int FL = CH2 <= 4 ? 512 : 1024;             // flag / tour / inverse-table entries per ant
  static_assert(CH2 >= 1 && CH2 <= 8, "two ants per wavefront: n <= 1024");
  // open[h][k]: f16 1.0 while node k is unvisited by ant h, else 0.0 (node order: the flag of a chosen node is found
  // without index arithmetic).  Slot j = c*4 + v of lane s is node c*128 + s*4 + v: a lane reads its flags as CH2
  // 8-byte pieces, 256 B apart.  Reused as the inverse-permutation table in the epilogue.
  __shared__ __attribute__((aligned(16))) _Float16 open_flags[8][FL];
  __shared__ __attribute__((aligned(16))) uint16_t tour_s[8][FL];    // tour_s[h][t] = node visited at step t
  __shared__ __attribute__((aligned(16))) float dstage[4][2][64];    // epilogue: edge lengths of one 64-step chunk
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: wave-uniform branches)
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
  const bool lane32 = __builtin_amdgcn_inverse_ballot_w64(0x0000000100000000ull);
  const uint32_t gid = p.ant_gid0 + (uint32_t)(b * (p.gid_bstride ? p.gid_bstride : A) + a);
  const char *Pb = (const char *)(p.P + (size_t)b * n * ld);           // uniform; lanes add 32-bit offsets
  const uint32_t ldb = (uint32_t)ld * 4u, lane_off = (uint32_t)s * 16u;
  const uint32_t a4 = (uint32_t)a * 4u;
  char *logp_t = LOGP ? (char *)(p.logp + (size_t)b * (n - 1) * A) : nullptr;
  char *rs_t = (LOGP && p.rowsum) ? (char *)(p.rowsum + (size_t)b * (n - 1) * A) : nullptr;
  _Float16 *fl = open_flags[wave * 2 + up];
  uint16_t *tour = tour_s[wave * 2 + up];

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
      fl[prev] = (_Float16)0.0f;
      tour[0] = (uint16_t)prev;
    }
    __builtin_amdgcn_wave_barrier();
    u32x4 ublk = {0, 0, 0, 0};                          // 128 cached uniforms per ant (lane s: block base+s)
    const bool has_noise = p.noise != nullptr;
    const float *uin = has_noise ? p.noise + (size_t)b * (n - 1) * A + a : nullptr;

    for (int tb = 0; tb < n; tb += 32) {
      // uniform of step t: lane (t&31), component (t>>5)&3 of Philox block ((t>>7)<<5) + lane
      if ((tb & 127) == 0) ublk = rng_block(p.seed, iter_now, STREAM_SCAN, gid, (uint32_t)(((tb >> 7) << 5) + s));
      const int ucur = __float_as_int(u01(comp(ublk, (tb >> 5) & 3)));
      const int i1 = n - tb < 32 ? n - tb : 32;
      uint32_t uaddr = (uint32_t)(lane - s + (tb == 0 ? 1 : 0)) * 4u;   // ds_bpermute address of the half's lane i
      for (int i = tb == 0 ? 1 : 0; i < i1; ++i) {
        const uint32_t rowoff = __umul24((uint32_t)prev, ldb);
        const uint32_t voff = rowoff + lane_off;
        float4 row[CH2];
        f16x4 fo[CH2];
#pragma unroll
        for (int c = 0; c < CH2; ++c) row[c] = *(const float4 *)(Pb + voff + c * 512);
        // the step's uniform sits in lane i of this half of ucur: fetched over the LDS crossbar (no VALU slot; the
        // row's latency covers it)
        float u = __int_as_float(__builtin_amdgcn_ds_bpermute((int)uaddr, ucur));
        uaddr += 4;
        if (has_noise) u = uin[(size_t)(tb + i - 1) * A];   // injected uniform stream (tests): [B][n-1][A]
#pragma unroll
        for (int c = 0; c < CH2; ++c) fo[c] = *(const f16x4 *)(fl + c * 128 + s * 4);

        // ---- the lane's running sums in slot order.  A closed slot adds p*0 = +0.0f; the product with the
        // 0/1 flag is exact, so each fma rounds once like an add.
        float run[32];
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < CH2; ++c) {
          acc = __builtin_fmaf(row[c].x, (float)fo[c][0], acc); run[4 * c + 0] = acc;
          acc = __builtin_fmaf(row[c].y, (float)fo[c][1], acc); run[4 * c + 1] = acc;
          acc = __builtin_fmaf(row[c].z, (float)fo[c][2], acc); run[4 * c + 2] = acc;
          acc = __builtin_fmaf(row[c].w, (float)fo[c][3], acc); run[4 * c + 3] = acc;
        }
        // ---- level 1: which lane
        const float part = acc;
        const float incl = half_scan_add<true>(part);
        const int incl_i = __float_as_int(incl);
        const int S0i = __builtin_amdgcn_readlane(incl_i, 31), S1i = __builtin_amdgcn_readlane(incl_i, 63);
        const float S0 = __int_as_float(S0i), S1 = __int_as_float(S1i);
        const float r0 = half_mul(u, S0, S1);               // u * S of the lane's half
        // Both row sums in [2^-100, +inf]: u*S cannot underflow (u >= 2^-24) and some open candidate has p > 0 -- decided
        // on the scalar unit.  Anything else (no feasible candidate, a NaN row, sums near the denormals) takes the
        // slow way: r is kept > 0 if u*S underflows, and an ant without candidates is flagged and sent to node 0
        const uint32_t tS0 = (uint32_t)S0i - 0x0D800000u, tS1 = (uint32_t)S1i - 0x0D800000u;
        const bool plain = (tS0 > tS1 ? tS0 : tS1) <= 0x7F800000u - 0x0D800000u;
        // the rest of the step; alive_h: some open candidate of ant h has p > 0 (compile-time true on the plain way)
        auto finish = [&](float r, auto alive0, auto alive1) {
          const uint64_t m = __builtin_amdgcn_fcmpf(incl, r, FCMP_OGE) & __builtin_amdgcn_fcmpf(part, 0.0f, FCMP_OGT);
          const int L0 = first_bit_or_0((uint32_t)m), L1 = first_bit_or_0((uint32_t)(m >> 32));   // chosen lane of each half
          // ---- level 2, in every lane (only lane L's result is read): what is left to cover inside the lane is
          // r - incl[L-1]; the candidate is the first slot whose running sum reaches it (such a slot is open with
          // p > 0; the threshold is kept > 0 so that a slot is "reached" only by a positive term)
          const float thr = at_least_denorm(half_excl_sub(incl, r, lane32));
          const int cnt = count_below32<NJ>(run, thr);
          int j0 = __builtin_amdgcn_readlane(cnt, L0), j1 = __builtin_amdgcn_readlane(cnt, L1 | 32);
          if (__builtin_expect((j0 > j1 ? j0 : j1) >= NJ, 0)) {
            // rounding: no running sum reached thr -> the lane's last open candidate with p > 0.  (Not "where the
            // running sum reaches its final value": a positive term can be absorbed by the sum before it -- a randomised
            // soak found that difference once in 6000 launches.)  Rare: the row and the flags are simply read again.
            // (one chunk at a time: the rare way must not cost the loop registers)
            int last = 0;
#pragma unroll 1
            for (int c = 0; c < CH2; ++c) {
              const float4 rw = *(const float4 *)(Pb + voff + c * 512);
              const f16x4 ff = *(const f16x4 *)(fl + c * 128 + s * 4);
              last = rw.x * (float)ff[0] > 0.0f ? 4 * c + 0 : last;
              last = rw.y * (float)ff[1] > 0.0f ? 4 * c + 1 : last;
              last = rw.z * (float)ff[2] > 0.0f ? 4 * c + 2 : last;
              last = rw.w * (float)ff[3] > 0.0f ? 4 * c + 3 : last;
            }
            if (j0 >= NJ) j0 = __builtin_amdgcn_readlane(last, L0);
            if (j1 >= NJ) j1 = __builtin_amdgcn_readlane(last, L1 | 32);
          }
          // slot j of lane L is node (j>>2)*128 + L*4 + (j&3): j*33 puts j>>2 at bit 7 and keeps j&3 (j < 32).
          // No feasible candidate (flagged; the reference raises): move to node 0 like the one-ant kernel and the oracle
          const int c0 = alive0 ? ((j0 * 33) & 0x383) + (L0 << 2) : 0;
          const int c1 = alive1 ? ((j1 * 33) & 0x383) + (L1 << 2) : 0;
          const int choice = half_pick(c0, c1);
          if (lead) {
            fl[choice] = (_Float16)0.0f;                     // visited
            tour[tb + i] = (uint16_t)choice;
            if constexpr (LOGP) {
              const float S = upper ? S1 : S0;
              const float pc = *(const float *)(Pb + rowoff + (uint32_t)choice * 4u);
              *(float *)(logp_t + a4) = clamp_log(pc / S);
              logp_t += (size_t)A * 4;
              if (rs_t) { *(float *)(rs_t + a4) = S; rs_t += (size_t)A * 4; }
            }
          }
          return choice;
        };
        int choice;
        if (__builtin_expect(plain, 1)) choice = finish(r0, std::true_type{}, std::true_type{});
        else {
          const bool alive0 = S0 > 0.0f, alive1 = S1 > 0.0f;     // S > 0 <=> some open candidate has p > 0
          if (!(alive0 && alive1) && p.flags && lane == 0) atomicOr(p.flags + b, 1);   // (rare: no state carried by the loop)
          choice = finish(at_least_denorm(r0), alive0, alive1);   // r kept > 0 if u*S underflows
        }
        // the next step's flag loads must follow this store (same wave: the LDS executes them in program order);
        // the compiler barrier keeps the program order
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        prev = choice;
      }
    }
  }

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
