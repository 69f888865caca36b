// daco_scan_sparse.hip -- TSP tour construction on HEAD / TAIL rows (sampler "scan_sparse"), four ants per wavefront.
//
// Reference behaviour: tsp/aco.py:134-177 with the roulette draw of tsp_nls/aco.py:260-275 on the k-sparse heuristic of the
// reference's inference (tsp/aco.py:52-67: k live entries per row, 1e-10 elsewhere).  The dense scan kernels stream one
// padded 2 KB row per ant-step and sit on the CU's vector-memory pipe (TA / TD 0.91-0.95 busy, profiles/r04_pmc_headline.txt);
// nothing in the instruction stream moves that, so this sampler moves fewer bytes: a row is split into a head -- up to
// 63 or 127 candidates chosen by the caller (64 / 128 slots), the colony takes the k largest heuristic entries -- and the tail.
// A step draws r = u (H + T), H = the head's live mass, T = the tail's STATIC mass (visited or not): r inside the head is an
// inverse CDF over the head's slots (384 / 768 bytes: values, ids); r past the head walks the whole tail row and, if it lands
// on a visited node, the step draws ONCE more -- the dense masked draw over all open candidates with a second uniform
// (rejection over a superset with an exact fallback: the outcome is the reference's categorical, and a step reads the row at
// most twice); no live head candidate -> the dense masked draw of the 64-lane scan specification with the first uniform.
// The CPU restatement of this draw (draw_scan_sparse in the oracle) is the specification (uniform stream, slot order, summation
// order); the GPU tests hold tours bit-exact against it, the CPU tests hold it against the categorical by chi-square.
//
// Layout: ant = 16 lanes (one DPP row), slot m = SPL * lane + v (SPL = 4 or 8).  Level 1 is the row scan of the 16 lane sums,
// level 2 a compare tree on the winning lane's running sums; the winning lane stores the choice and the ant's lanes read it back.
// The rare ways (tail walk, dense draw) are wave-cooperative: the 64 lanes serve one ant at a time (candidate
// k = (c * 64 + lane) * 4 + v, the 64-lane scan).  Tours (u16) and visited flags (bytes, node order) live in LDS; paths / tour
// lengths / the update's table leave the workgroup in the epilogue of the scan16 family (16 ants = one 128-byte run per row).
#include "daco_sample_kernel.h"
#include "daco_head_rows.h"

namespace daco {

// (head slots per row, bytes per lane: daco_head_rows.h.  32-byte lanes for SPL = 4 -- every 16-byte load aligned, four cache
// lines per row instead of three -- measured 0.61 ms against 0.58 at the headline shape.)
constexpr int SP_FCMP_OGT = 2, SP_FCMP_OGE = 3, SP_FCMP_OLT = 4;

enum : uint32_t { STREAM_SPARSE = 4, STREAM_SPARSE_RETRY = 5 };   // (retry: the second uniform of a step, block t << 8, component 0)

template <int N> __device__ inline float sp_row_bcast(float x) { return dpp_f<0x150 + N, 0xF, false>(x, x); }
template <int N> __device__ inline int sp_row_ror(int x) { return dpp_i<0x120 + N, 0xF, false>(x, x); }
__device__ inline float sp_row_scan(float x) {
  x = x + dpp_f<DPP_ROW_SHR(1), 0xF, true>(0.0f, x);
  x = x + dpp_f<DPP_ROW_SHR(2), 0xF, true>(0.0f, x);
  x = x + dpp_f<DPP_ROW_SHR(4), 0xF, true>(0.0f, x);
  x = x + dpp_f<DPP_ROW_SHR(8), 0xF, true>(0.0f, x);
  return x;
}
// of the lanes set in m, the first one of every row of 16
__device__ inline uint64_t sp_row_first(uint64_t m) { return m & ~((m | 0x8000800080008000ull) - 0x0001000100010001ull); }

// ---------------------------------------------------------------------------------------------------------------
// the pre-pass of one iteration, one wavefront per row: the HEAD ROW the scan reads each step (daco_head_rows.h emit_head_row:
// 16 lanes x ls bytes, lane s = {values of slots 4s..4s+3 (f32), their node ids (u16)}; the last slot = the tail total), from one
// read of the rows of tau and eta.  Round 6: the dense fused row P = tau^alpha * eta^beta is no longer written (65 MB per
// iteration at the headline shape for the 1.4 % of the steps that walked it): the rare ways form tau * eta from the two rows
// themselves (sp_prob4: the same product), and an iteration whose pheromone update emitted the head rows
// (daco_pheromone_update_heads) does not run this kernel at all.
template <bool RACE, bool VEC4, int CH>
__global__ void __launch_bounds__(256)
sparse_prepass_kernel(int B, int n, int ch, const float *tau, long tau_bs, const float *eta, long eta_bs,
                      const uint16_t *hid, float *P, char *hrow, int spl, int dead) {
  __shared__ uint32_t bm[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= (long)B * n) return;
  const int b = (int)(row / n), r = (int)(row - (long)b * n);
  const float *tr = tau + b * tau_bs + (long)r * n, *er = eta + b * eta_bs + (long)r * n;
  emit_head_row<RACE, CH, VEC4>(n, ch, tr, er, hid + row * (16 * spl), bm[wave], hrow + row * sp_head_row_bytes(spl), P + row * (256 * ch), spl, dead, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// one wave-cooperative draw over a whole row for ONE ant (the rare ways).  TAIL = false: the dense masked draw of the
// 64-lane scan specification with uniform `ur` (oracle draw_scan, lanes = 64).  TAIL = true: the walk over the row's
// non-head entries, visited or not, with the threshold `ur` given (oracle draw_scan_sparse, "past the head").
// Returns the node, -1 if no candidate can be drawn (dense: infeasible; tail: a tail without mass).
// (the row itself: the padded dense row of P the pre-pass / the update wrote next to the head row)
struct SpRowSrc { const float *p; };
__device__ __forceinline__ float4 sp_row4(const SpRowSrc &r, int k0) { return *reinterpret_cast<const float4 *>(r.p + k0); }

template <int CHD, bool TAIL>
__device__ __forceinline__ int sparse_row_walk(const SpRowSrc &rowp, const uint8_t *flg, const uint32_t *bm, int lane, float ur) {
  constexpr int NJ = CHD * 4;
  float run[16];
  float acc = 0.0f;
  uint32_t pos = 0;                                      // slots with a positive term (a term can be absorbed by the running sum)
#pragma unroll
  for (int c = 0; c < CHD; ++c) {
    const int k0 = (c * 64 + lane) * 4;
    const float4 rv = sp_row4(rowp, k0);
    float f[4];
    if constexpr (TAIL) {
      const uint32_t w = bm[(k0 >> 5) & 31] >> (k0 & 31);
      f[0] = (w & 1u) ? 0.0f : 1.0f; f[1] = (w & 2u) ? 0.0f : 1.0f; f[2] = (w & 4u) ? 0.0f : 1.0f; f[3] = (w & 8u) ? 0.0f : 1.0f;
    } else {
      const uint32_t ff = *reinterpret_cast<const uint32_t *>(flg + k0);          // four flag bytes (v_cvt_f32_ubyte0..3)
      f[0] = (float)(ff & 0xFFu); f[1] = (float)((ff >> 8) & 0xFFu); f[2] = (float)((ff >> 16) & 0xFFu); f[3] = (float)(ff >> 24);
    }
    acc = __builtin_fmaf(rv.x, f[0], acc); run[4 * c + 0] = acc;
    acc = __builtin_fmaf(rv.y, f[1], acc); run[4 * c + 1] = acc;
    acc = __builtin_fmaf(rv.z, f[2], acc); run[4 * c + 2] = acc;
    acc = __builtin_fmaf(rv.w, f[3], acc); run[4 * c + 3] = acc;
    pos |= (rv.x * f[0] > 0.0f ? 1u : 0u) << (4 * c) | (rv.y * f[1] > 0.0f ? 2u : 0u) << (4 * c) |
           (rv.z * f[2] > 0.0f ? 4u : 0u) << (4 * c) | (rv.w * f[3] > 0.0f ? 8u : 0u) << (4 * c);
  }
#pragma unroll
  for (int j = NJ; j < 16; ++j) run[j] = __builtin_inff();
  const float part = acc;
  const float incl = wave_scan_add(part);
  float r = ur;
  if constexpr (!TAIL) {
    const float S = readlane_f(incl, 63);
    if (!(S > 0.0f)) return -1;
    r = ur * S;
    r = r > 0.0f ? r : 1.401298464e-45f;
  }
  uint64_t m = __ballot(incl >= r && part > 0.0f);
  int L;
  if (m == 0) {
    if constexpr (!TAIL) return -1;
    m = __ballot(part > 0.0f);                           // rounding: the last lane with mass
    if (m == 0) return -1;
    L = 63 - __builtin_clzll(m);
  } else {
    L = __builtin_ctzll(m);
  }
  const float excl = L ? readlane_f(incl, L - 1) : 0.0f;
  const float thr = fmaxf(r - excl, 1.401298464e-45f);   // (> 0: a slot is reached only by a positive term)
  const int cnt = count_below<NJ>(run, thr);
  int jsel = readlane_i(cnt, L);
  if (jsel >= NJ) {                                      // no running sum reached thr: the lane's last positive entry
    const uint32_t pl = (uint32_t)readlane_i((int)pos, L);
    jsel = pl ? 31 - __builtin_clz(pl) : 0;
  }
  return ((jsel >> 2) * 64 + L) * 4 + (jsel & 3);
}

// ---------------------------------------------------------------------------------------------------------------
// CHD: 256-candidate chunks of the dense row (n <= 256 * CHD): 2 or 4.
// RACE: the exponential race of DACO_RACE_PHILOX (the reference's torch.multinomial arithmetic, tsp/aco.py:174-175, with in-kernel
// noise) on the same head rows, with the SAME result as the dense race kernel: the winner over the head's open candidates is the
// winner over the whole row whenever its key is below every key a tail candidate could possibly draw --
// key_j = L_j / p_j >= L_min / max_tail(p), L_min = the smallest value the noise can take (u = 2^-24) -- and that is checked each
// step (with a factor 1/2 of margin); otherwise the step runs the dense race for that ant.  Noise indexed by node id exactly as in
// the dense kernel (Philox block (t << 12) | (k >> 2), component k & 3): bit-identical tours, a head step generates 64 variates
// instead of 512 (the dense race kernel is bound by VALU issue: one Philox block per four candidates and a degree-8 polynomial
// per candidate, profiles/r04_pmc_race.txt).
//
// The step of the scan is bound by instruction issue (profiles/r04_scan_sparse_ablation.txt: the kernel without any memory access
// runs at 0.72 of its time, and twelve more VALU instructions per step cost 13 %), so the loop is written for few instructions:
// one buffer address per step (the row index scales into the 384-byte row), the visited flag of slot 63 and of the empty slots
// comes from a flag byte that is never set (no select for "not a candidate"), lane 15 forms r = u (H + T) from its own registers
// and broadcasts the product, the chosen slot is a three-deep select on the running sums, the WINNING lane of each ant stores the
// flag byte and the tour entry, and the row reads the tour entry back as the next row index (one LDS read instead of a
// four-stage DPP OR network).  All of that leaves the arithmetic (summation order, thresholds, rounding cases) as the oracle has it.
typedef uint32_t sp_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t sp_u32x2 __attribute__((ext_vector_type(2)));
__device__ inline float sp_at_least_denorm(float x) {     // max(x, denorm_min) for x that is not a signalling NaN
  float y;
  asm("v_max_f32 %0, 1, %1" : "=v"(y) : "v"(x));
  return y;
}

// LH (round 6, "LDS heads"; CHD = 2, scan draw, four slots per lane): the launch for FEW ants -- one instance, a few hundred ants:
// the reference's own calling pattern (tsp/test.ipynb:66-68 loops the instances, one colony each).  Such a launch has fewer
// wavefronts than the chip has SIMDs, so a step is a bare latency chain, and a third of it is the L2 round trip of the head row
// (0.41 us per step, 203 us per launch at TSP-500 x 512 ants x 1 instance, profiles/r06_kernel_stats_b1_before.csv).  Here a
// workgroup is ONE wavefront of four ants (three more wavefronts help with the copy and the epilogue) and keeps the instance's
// whole head table in LDS: the rows without their empty lanes (lanes 0 .. kl-1 of 16, kl = ceil((kmax + 1) / 4) for at most kmax
// live slots per row) with the tail total moved into the first slot no row uses (slot kmax, whose id stays `dead`: its term is
// T x 0 = 0 as before) -- 500 rows x 312 bytes at k = 50.  The step then reads its head row from LDS; values, order of the
// additions, uniforms and decisions are those of the kernel above, so the tours are the same bit for bit.
template <int CHD, bool RACE, int SPL, bool LH = false>
__global__ void __launch_bounds__(256, LH ? 1 : (CHD == 2 ? (RACE ? 4 : 6) : 5))
scan_sparse_kernel(const SampleParams p) {
  static_assert(!LH || (CHD == 2 && !RACE && SPL == 4), "the LDS-heads variant: n <= 512, scan draw, 64-slot heads");
  constexpr int APW = 4, APB = LH ? 4 : 16;
  constexpr int LS = sp_lane_bytes(SPL);                 // bytes per lane of a head row
  constexpr int LAST = SPL - 1;                          // (lane 15: the slot of the tail total)
  constexpr int FL = CHD * 256;                          // tour / inverse-table entries per ant (>= n)
  constexpr int FLP = FL + 16;                           // flag bytes per ant: entry FL is never set (the id of slot 63 and of empty slots)
  constexpr int FLT = FL + 2;                            // u16 entries between two ants' tours in LDS (n <= 512): 1028 bytes, so that the
                                                         // four ants of a wavefront (and the sixteen of the epilogue) fall on different banks
  constexpr uint32_t ROWB = 16u * LS;                    // bytes of a head row: lane s holds {SPL f32 values, SPL u16 ids} at s * LS
  // LDS (one dynamic block): visited flags as BYTES (1 while node k is unvisited, node order) and
  //   n <= 512: the u16 tours -- 1.5 KB per ant, six workgroups per CU (all outputs leave in the epilogue below);
  //   n > 512 (TG): a 16-step window of each tour only.  The tours go to global memory 32 bytes per ant every 16 steps and come
  //   back eight at a time for the epilogue, which reuses the block: 3.1 KB per ant would allow three workgroups per CU, 32 KB
  //   per workgroup allow four, and at this size the launch time follows the occupancy (two instead of three: 6.5 -> 8.6 ms at
  //   TSP-1000 x 2048 x 64; its 8 192 workgroups run in many rounds, so the epilogue runs under other workgroups' loops).
  constexpr bool TG = CHD == 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char sparse_dyn[];
  uint8_t *flag_mem = sparse_dyn;                                                   // [APB][FLP]
  uint16_t *tour_mem = reinterpret_cast<uint16_t *>(sparse_dyn + APB * FLP);      // !TG: [APB][FL]; TG: [APB][16], the window
  // tail walk: the head of the row as a bitmap over the nodes, one per wavefront (inside the block at n > 512, so that five
  // workgroups of 32 KB fill a CU's LDS)
  __shared__ uint32_t bm_static[TG ? 1 : (LH ? 1 : 4)][32];
  uint32_t (*bm_s)[32] = TG ? reinterpret_cast<uint32_t (*)[32]>(sparse_dyn + APB * FLP + APB * 32) : bm_static;
  // LH: the head table behind the flags and the tours: [n][kl] lane records of LS bytes, then one empty record
  unsigned char *lh_tab = sparse_dyn + APB * FLP + APB * FLT * 2;
  static_assert(!LH || ((APB * FLP + APB * FLT * 2) % 16 == 0), "LH: table aligned");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, s = lane & 15;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int bpi = (p.A + APB - 1) / APB;
  const int b = w / bpi;
  const int abase = (w - b * bpi) * APB;
  const int a0 = abase + wave * APW;
  const int n = p.n, A = p.A;
  const uint64_t iter_now = p.iter + (p.iter_dev ? *p.iter_dev : 0ull);
  const bool active = a0 < A && (!LH || wave == 0);
  const int a = a0 + q < A ? a0 + q : A - 1;             // spare groups build ant A-1 again (not written)
  const uint32_t gid = p.ant_gid0 + (uint32_t)(b * (p.gid_bstride ? p.gid_bstride : A) + a);
  const float *Pb = p.P + (size_t)b * n * p.ld;
  const char *hrb = (const char *)p.hval + (size_t)b * n * ROWB;
  const __amdgpu_buffer_rsrc_t hres = __builtin_amdgcn_make_buffer_rsrc((void *)hrb, 0, (int)((uint32_t)n * ROWB), 0x00020000);
#define SP_ROW_OF(pv) SpRowSrc{Pb + (size_t)(pv) * p.ld}
  uint32_t sls = (uint32_t)s * LS;
  asm volatile("" : "+v"(sls));                          // (kept in a register: the loop adds it to the row offset)
  uint8_t *fl = flag_mem + (wave * APW + q) * FLP;
  uint16_t *tour = tour_mem + (wave * APW + q) * (TG ? 16 : FLT);
  // TG: tour entry t lives at tour[t & 15] until its chunk is flushed to the workgroup's rows of tours16 [B][A][FL]
  uint16_t *t16b = TG ? p.tours16 + ((size_t)b * A + abase) * FL : nullptr;
  const uint32_t t16o = (uint32_t)((wave * APW + q) * FL + s);
#define SP_T(t) (TG ? ((t) & 15) : (t))
  bool infeasible = false;
  unsigned long long n_dense = 0, n_tail = 0, n_rej = 0;
  // LH: per-lane constants of the LDS row read (lanes past the live records read the empty record), the tail total's place
  uint32_t lh_rb = 0, lh_rbsel = 0, lh_loff = 0, lh_toff = 0;
  if constexpr (LH) {
    const int kl = p.lh_kl, kmax = p.lh_kmax;
    lh_rb = (uint32_t)kl * LS;
    lh_toff = (uint32_t)(kmax / SPL) * LS + (uint32_t)(kmax % SPL) * 4u;
    lh_rbsel = s < kl ? lh_rb : 0u;
    lh_loff = s < kl ? (uint32_t)s * LS : (uint32_t)n * lh_rb;
    // the copy: sixteen threads per row, 8 bytes at a time; then the tail totals (slot 63 of the full row -> slot kmax)
    const char *src = hrb;
    const int l16 = threadIdx.x & 15;
    // (eight rows per thread in flight: one row per iteration the copy was 31 dependent memory round trips -- load, wait, LDS
    // store -- in front of a launch that is a latency chain anyway)
    if (l16 < kl)
      for (int row0 = threadIdx.x >> 4; row0 < n; row0 += 16 * 8) {
        uint2 x[8][3];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int row = row0 + 16 * u;
          const uint2 *g = reinterpret_cast<const uint2 *>(src + (size_t)(row < n ? row : row0) * ROWB + l16 * LS);
          x[u][0] = g[0]; x[u][1] = g[1]; x[u][2] = g[2];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int row = row0 + 16 * u;
          if (row < n) {
            uint2 *d = reinterpret_cast<uint2 *>(lh_tab + (uint32_t)row * lh_rb + l16 * LS);
            d[0] = x[u][0]; d[1] = x[u][1]; d[2] = x[u][2];
          }
        }
      }
    if (threadIdx.x < 4) reinterpret_cast<float *>(lh_tab + (uint32_t)n * lh_rb)[threadIdx.x] = 0.0f;
    if (threadIdx.x >= 4 && threadIdx.x < 6) reinterpret_cast<uint32_t *>(lh_tab + (uint32_t)n * lh_rb)[threadIdx.x] = (uint32_t)p.ld * 0x00010001u;
    __syncthreads();
    bool too_many = false;
    for (int row = threadIdx.x; row < n; row += 256) {
      const char *gr = src + (size_t)row * ROWB;
      // (slot kmax must be empty in every row: its id is `dead`; a table with more live slots than the caller said is an error)
      const uint16_t idk = *reinterpret_cast<const uint16_t *>(gr + (kmax / SPL) * LS + SPL * 4 + (kmax % SPL) * 2);
      too_many |= idk != (uint16_t)p.ld;
      *reinterpret_cast<float *>(lh_tab + (uint32_t)row * lh_rb + lh_toff) = *reinterpret_cast<const float *>(gr + 15 * LS + LAST * 4);
    }
    if (too_many && p.flags) atomicOr(p.flags + b, 4);
    __syncthreads();
  }

  if (active) {
    {
      const uint4 ones = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
#pragma unroll
      for (int g = 0; g < FL / 256; ++g) *(uint4 *)(fl + g * 256 + s * 16) = ones;
      fl[FL + s] = 0;
    }
    int prev;
    if (p.start) prev = (int)p.start[(size_t)b * A + a];
    else if (p.fixed_start >= 0) prev = p.fixed_start;
    else {
      const u32x4 r = rng_block(p.seed, iter_now, STREAM_START, gid, 0);
      prev = (int)__umulhi(r.x, (uint32_t)n);
    }
    __builtin_amdgcn_wave_barrier();
    if (s == 0) { fl[prev] = 0; tour[0] = (uint16_t)prev; }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    u32x4 ublk = {0, 0, 0, 0};
    float ucur = 0.0f;
    // (Round 6 tried to sum the tour lengths while the tours are built -- one 64-lane gather of edge lengths per chunk of sixteen
    // steps, added in step order by row_newbcast: bit-identical, and 150 us SLOWER at the headline shape than the epilogue's
    // 65 us: loads return in order, so every chunk's first head-row load waited behind a gather of 64 different cache lines.
    // profiles/r06_fused_head_rows.txt.)
    for (int t0 = 0; t0 < n; t0 += 16) {
      int t = t0;
      if constexpr (!RACE) {
        // uniform of step t: component (t>>4)&3 of Philox block ((t>>6)<<4) + (t&15); lane s computes the one of step
        // t0 + 15 - s, the row is rotated by one lane per step so that lane 15 holds the current one
        if ((t0 & 63) == 0) ublk = rng_block(p.seed, iter_now, STREAM_SPARSE, gid, (uint32_t)(((t0 >> 6) << 4) + (15 - s)));
        ucur = u01(comp(ublk, (t0 >> 4) & 3));
        if (t0 == 0) ucur = __int_as_float(sp_row_ror<1>(__float_as_int(ucur)));      // (there is no step 0)
      }
      if (t0 == 0) t = 1;
      const int te = t0 + 16 < n ? t0 + 16 : n;
#ifndef DACO_SPARSE_UNROLL
#define DACO_SPARSE_UNROLL 1
#endif
#pragma unroll DACO_SPARSE_UNROLL
      for (; t < te; ++t) {
        // ---- the head of row `prev`: SPL values and SPL ids per lane
        const uint32_t off = LH ? __umul24((uint32_t)prev, lh_rbsel) + lh_loff : __umul24((uint32_t)prev, ROWB) + sls;
        // (the id words stay named scalars, never an array: a select between array elements is turned into an indexed load
        // and the array then lives in scratch memory)
        float h[SPL];
        uint32_t w0, w1, w2 = 0, w3 = 0;                    // the ids, two per word
        if constexpr (LH) {
          const uint2 ra = *reinterpret_cast<const uint2 *>(lh_tab + off), rb_ = *reinterpret_cast<const uint2 *>(lh_tab + off + 8),
                      rc = *reinterpret_cast<const uint2 *>(lh_tab + off + 16);
          const float tv = *reinterpret_cast<const float *>(lh_tab + __umul24((uint32_t)prev, lh_rb) + lh_toff);
          h[0] = __uint_as_float(ra.x); h[1] = __uint_as_float(ra.y); h[2] = __uint_as_float(rb_.x);
          h[3] = s == 15 ? tv : __uint_as_float(rb_.y);      // (lane 15's last slot: the tail total)
          w0 = rc.x; w1 = rc.y;
        } else if constexpr (SPL == 4) {
          const sp_u32x4 hvr = __builtin_amdgcn_raw_buffer_load_b128(hres, off, 0, 0);
          const sp_u32x2 hi2 = __builtin_amdgcn_raw_buffer_load_b64(hres, off + 16u, 0, 0);
          h[0] = __uint_as_float(hvr.x); h[1] = __uint_as_float(hvr.y); h[2] = __uint_as_float(hvr.z); h[3] = __uint_as_float(hvr.w);
          w0 = hi2.x; w1 = hi2.y;
        } else {
          const sp_u32x4 hva = __builtin_amdgcn_raw_buffer_load_b128(hres, off, 0, 0);
          const sp_u32x4 hvb = __builtin_amdgcn_raw_buffer_load_b128(hres, off + 16u, 0, 0);
          const sp_u32x4 hi4 = __builtin_amdgcn_raw_buffer_load_b128(hres, off + 32u, 0, 0);
          h[0] = __uint_as_float(hva.x); h[1] = __uint_as_float(hva.y); h[2] = __uint_as_float(hva.z); h[3] = __uint_as_float(hva.w);
          h[4] = __uint_as_float(hvb.x); h[5] = __uint_as_float(hvb.y); h[6] = __uint_as_float(hvb.z); h[7] = __uint_as_float(hvb.w);
          w0 = hi4.x; w1 = hi4.y; w2 = hi4.z; w3 = hi4.w;
        }
        if constexpr (SPL == 8) asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));   // (no longer elements of one vector: no dynamic extract)
#define SP_IDW(j) ((j) == 0 ? w0 : (j) == 1 ? w1 : (j) == 2 ? w2 : w3)
#define SP_ID(v) (((v) & 1) ? (int)(SP_IDW((v) >> 1) >> 16) : (int)(SP_IDW((v) >> 1) & 0xFFFFu))
        if constexpr (RACE) {
          // ---- the race on the head: 1 / p and node id of SPL slots per lane, one variate each (the last slot and the empty
          // slots: a flag that is never set -> +inf)
          const float rtail = sp_row_bcast<15>(h[LAST]);   // the last slot: the tail's bound
          float bk = __builtin_inff();
          int bi = 0x7fffffff;
#pragma unroll
          for (int v = 0; v < SPL; ++v) {
            const int k = SP_ID(v);
            const u32x4 r4 = rng_block(p.seed, iter_now, STREAM_RACE, gid, ((uint32_t)t << 12) | (uint32_t)(k >> 2));
            const float Lk = neg_log2_1m(u01(comp(r4, k & 3)));
            const float key = fl[k] != 0 ? Lk * h[v] : __builtin_inff();
            if (prefer<false>(key, k, bk, bi)) { bk = key; bi = k; }
          }
          arg_step<false, DPP_ROW_SHR(1), 0xF>(bk, bi);
          arg_step<false, DPP_ROW_SHR(2), 0xF>(bk, bi);
          arg_step<false, DPP_ROW_SHR(4), 0xF>(bk, bi);
          arg_step<false, DPP_ROW_SHR(8), 0xF>(bk, bi);
          const float best = sp_row_bcast<15>(bk);
          int choice = __float_as_int(sp_row_bcast<15>(__int_as_float(bi)));
          // no tail candidate can beat `best` if best < L_min * min_tail(1/p) (half of it: margin for the polynomial's last bits)
          const float lmin = neg_log2_1m(0x1p-24f);
          uint64_t rare = __ballot(!(best < 0.5f * lmin * rtail)) & 0x0001000100010001ull;
          while (rare) {                                    // the dense race for one ant, all 64 lanes
            const int gl = __builtin_ctzll(rare);
            rare &= rare - 1;
            const int g = gl >> 4;
            const int pv = readlane_i(prev, gl);
            const uint32_t gidg = (uint32_t)readlane_i((int)gid, gl);
            const uint8_t *flg = flag_mem + (wave * APW + g) * FLP;
            const SpRowSrc rowp = SP_ROW_OF(pv);
            n_dense += a0 + g < A ? 1ull : 0ull;
            float dk = __builtin_inff();
            int di = 0x7fffffff;
#pragma unroll
            for (int c = 0; c < CHD; ++c) {
              const int k0 = (c * 64 + lane) * 4;
              const float4 pvv = sp_row4(rowp, k0);
              const uint32_t ff = *reinterpret_cast<const uint32_t *>(flg + k0);
              const u32x4 r4 = rng_block(p.seed, iter_now, STREAM_RACE, gidg, ((uint32_t)t << 12) | (uint32_t)(c * 64 + lane));
              const float pp[4] = {pvv.x, pvv.y, pvv.z, pvv.w};
#pragma unroll
              for (int v = 0; v < 4; ++v) {
                const float Lk = neg_log2_1m(u01(comp(r4, v)));
                const float key = ((ff >> (8 * v)) & 0xFFu) ? Lk * (1.0f / pp[v]) : __builtin_inff();     // (padding: p = 0 -> inf)
                if (key < dk) { dk = key; di = k0 + v; }
              }
            }
            const KeyIdx rr = wave_arg<false>(dk, di);
            int cg = rr.idx;
            if (!(rr.key < __builtin_inff())) { infeasible = true; cg = 0; }
            choice = q == g ? cg : choice;
          }
          if (s == 0) { fl[choice] = 0; tour[SP_T(t)] = (uint16_t)choice; }
          asm volatile("" ::: "memory");                    // the next step's flag reads follow these stores
          __builtin_amdgcn_wave_barrier();
          prev = choice;
        } else {
          float f[SPL], run[SPL];
#pragma unroll
          for (int v = 0; v < SPL; ++v) f[v] = (float)fl[SP_ID(v)];
          run[0] = __builtin_fmaf(h[0], f[0], 0.0f);
#pragma unroll
          for (int v = 1; v < SPL; ++v) run[v] = __builtin_fmaf(h[v], f[v], run[v - 1]);      // (lane 15, last slot: the tail total T, its flag is never set)
          const float incl = sp_row_scan(run[LAST]);
          const float excl = dpp_f<DPP_ROW_SHR(1), 0xF, true>(0.0f, incl);
          // r = u (H + T), kept > 0: lane 15 has all three
          const float r = sp_row_bcast<15>(sp_at_least_denorm(ucur * (incl + h[LAST])));
          ucur = __int_as_float(sp_row_ror<1>(__float_as_int(ucur)));

          // (macros, not lambdas: with eight slots per lane the closure of a lambda called from two places was built in scratch memory)
          // the node of the lane's last slot with a positive term (not "where the running sum stops growing": a term can be absorbed)
#define SP_LAST_POSITIVE(out)                                                                                     \
          do {                                                                                                    \
            int lp_ = 0;                                                                                          \
            _Pragma("unroll") for (int v = 1; v < SPL; ++v) lp_ = h[v] * f[v] > 0.0f ? v : lp_;                   \
            const uint64_t lo_ = ((uint64_t)w1 << 32) | w0, hi_ = ((uint64_t)w3 << 32) | w2;                      \
            out = (int)(((lp_ >= 4 ? hi_ : lo_) >> (16 * (lp_ & 3))) & 0xFFFFu);                                  \
          } while (0)
          // level 1 + 2 for a given threshold: the lanes in `first_out` (one per ant, none if rr is past the head) hold the winner
          // in sel_out.  The first slot whose running sum reaches thr (the sums do not decrease, so "sums below thr" is a prefix):
          // the id pair by the odd-numbered sums below, the half of the pair by the parity of the count; rounding (no running sum
          // reached thr): the lane's last positive slot
#define SP_HEAD_DECIDE(rr, first_out, sel_out)                                                                    \
          do {                                                                                                    \
            const float rr_ = (rr);                                                                               \
            const uint64_t m_ = __builtin_amdgcn_fcmpf(incl, rr_, SP_FCMP_OGE) & __builtin_amdgcn_fcmpf(run[LAST], 0.0f, SP_FCMP_OGT); \
            first_out = sp_row_first(m_);                                                                         \
            const float thr_ = sp_at_least_denorm(rr_ - excl);                                                    \
            bool below_[SPL];                                                                                     \
            _Pragma("unroll") for (int v = 0; v < SPL - 1; ++v) below_[v] = run[v] < thr_;                        \
            uint32_t pair_;                                                                                       \
            if constexpr (SPL == 4) pair_ = below_[1] ? w1 : w0;                                                  \
            else pair_ = below_[3] ? (below_[5] ? w3 : w2) : (below_[1] ? w1 : w0);                               \
            bool odd_ = below_[0];                                                                                \
            _Pragma("unroll") for (int v = 1; v < SPL - 1; ++v) odd_ = odd_ != below_[v];                         \
            sel_out = odd_ ? (int)(pair_ >> 16) : (int)(pair_ & 0xFFFFu);                                         \
            const uint64_t bad_ = first_out & __builtin_amdgcn_fcmpf(run[LAST], thr_, SP_FCMP_OLT);               \
            if (__builtin_expect(bad_ != 0, 0)) {                                                                 \
              int lastp_;                                                                                         \
              SP_LAST_POSITIVE(lastp_);                                                                           \
              sel_out = run[LAST] < thr_ ? lastp_ : sel_out;                                                      \
            }                                                                                                     \
          } while (0)
          int sel;
          uint64_t first;
          SP_HEAD_DECIDE(r, first, sel);


          // ---- the rare ways (an ant without a winner), one ant at a time with the whole wavefront
          if (__builtin_expect(__builtin_popcountll(first) != APW, 0)) {
            uint64_t any = first | (first >> 8);
            any |= any >> 4; any |= any >> 2; any |= any >> 1;
            uint64_t rare = ~any & 0x0001000100010001ull;
            while (rare) {
              const int gl = __builtin_ctzll(rare);           // lane 0 of the group
              rare &= rare - 1;
              const int g = gl >> 4;
              const int pv = readlane_i(prev, gl);
              const float Hg = readlane_f(incl, gl + 15);
              const float ug = readlane_f(ucur, gl), rg = readlane_f(r, gl);     // (ucur: already rotated, lane 0 holds this step's)
              const uint32_t gidg = (uint32_t)readlane_i((int)gid, gl);
              const uint8_t *flg = flag_mem + (wave * APW + g) * FLP;
              const SpRowSrc rowp = SP_ROW_OF(pv);
              int choice_g = -1;
              const unsigned long long real = a0 + g < A ? 1ull : 0ull;      // (a spare group repeats ant A-1: not counted)
              if (!(Hg > 0.0f)) {                               // no live head candidate: the dense masked draw with this uniform
                n_dense += real;
                choice_g = sparse_row_walk<CHD, false>(rowp, flg, nullptr, lane, ug);
              } else {
                n_tail += real;
                // the head's nodes as a bitmap, from the ids the ant's own lanes hold (an empty slot holds an id >= n)
                if (lane < 32) bm_s[wave][lane] = 0u;
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                if (q == g) {
#pragma unroll
                  for (int v = 0; v < SPL; ++v) {
                    const int idl = SP_ID(v);
                    if (idl < n) atomicOr(&bm_s[wave][idl >> 5], 1u << (idl & 31));
                  }
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                float rp = rg - Hg;
                rp = rp > 0.0f ? rp : 1.401298464e-45f;
                const int j = sparse_row_walk<CHD, true>(rowp, flg, bm_s[wave], lane, rp);
                if (j < 0) {                                    // a tail without mass: the head's last live candidate
                  const bool live_lane = q == g && run[LAST] > 0.0f;
                  const uint64_t ml = __ballot(live_lane);
                  if (ml != 0) {
                    const int Ll = 63 - __builtin_clzll(ml);
                    int lastp;
                    SP_LAST_POSITIVE(lastp);
                    choice_g = readlane_i(lastp, Ll);
                  }
                } else if (flg[j] != 0) {                       // open: accepted
                  choice_g = j;
                } else {
                  // visited: ONE more draw, the dense masked draw over all open candidates with the step's second uniform --
                  // x then has probability p_x / (H + T) + (T_visited / (H + T)) p_x / (H + T_open) = p_x / (H + T_open)
                  n_rej += real;
                  n_dense += real;
                  const u32x4 rb = rng_block(p.seed, iter_now, STREAM_SPARSE_RETRY, gidg, (uint32_t)t << 8);
                  choice_g = sparse_row_walk<CHD, false>(rowp, flg, nullptr, lane, u01(rb.x));
                }
              }
              if (choice_g < 0) { infeasible = true; choice_g = 0; }
              if (lane == gl) { fl[choice_g] = 0; tour[SP_T(t)] = (uint16_t)choice_g; }
            }
          }
          // the winning lane of every ant marks the node and appends it; the ant's lanes read it back as the next row
          const bool winner = __builtin_amdgcn_inverse_ballot_w64(first);
          if (winner) { fl[sel] = 0; tour[SP_T(t)] = (uint16_t)sel; }
          asm volatile("" ::: "memory");                    // (same wavefront: the LDS executes these in program order)
          // (LH, measured and not kept: the next row index through the ant's sixteen lanes by four DPP rotations instead of this LDS
          // store and load -- 0.2278 against 0.2161 ms per colony iteration at TSP-500 x 512 x 1: the extra branch and five
          // dependent VALU instructions cost more than the LDS round trip they replace)
          prev = tour[SP_T(t)];
          asm volatile("" ::: "memory");
        }
      }
      if constexpr (TG) {
        // the chunk's sixteen entries leave the window: 32 contiguous bytes per ant (entries past n - 1: never read)
        asm volatile("" ::: "memory");
        const uint16_t wv = tour[s];
        if (a0 + q < A) t16b[t16o + (uint32_t)t0] = wv;
        asm volatile("" ::: "memory");
      }
    }
  }
#undef SP_T
#undef SP_ROW_OF
#undef SP_HEAD_DECIDE
#undef SP_LAST_POSITIVE
#undef SP_ID
#undef SP_IDW
  if (infeasible && p.flags && lane == 0) atomicOr(p.flags + b, 1);
  if (p.stats && lane == 0 && (n_dense | n_tail | n_rej)) {
    atomicAdd(p.stats + 0, n_dense); atomicAdd(p.stats + 1, n_tail); atomicAdd(p.stats + 2, n_rej);
  }

  // ------------------------------------------------------------------ epilogue: the workgroup's 16 tours leave
  const int nant = A - abase < APB ? A - abase : APB;
  if constexpr (!TG) {
    uint16_t (*tour_s)[FLT] = reinterpret_cast<uint16_t (*)[FLT]>(tour_mem);
    __syncthreads();
    const int k16 = threadIdx.x & (APB - 1);
    constexpr int TSTEP = 256 / APB;
    if (p.paths && k16 < nant) {
      int64_t *pb = p.paths + (size_t)b * n * A + abase;
      for (int t = threadIdx.x / APB; t < n; t += TSTEP) pb[(size_t)t * A + k16] = (int64_t)tour_s[k16][t];
    }
    if (!p.paths && p.tours16) {
      // no int64 paths asked for: the tours leave as they are, u16 rows of FL entries per ant in the workspace (a quarter of
      // the bytes: 32 MB instead of 131 at the headline shape; daco_track_best_tours16 reads the best one)
      uint32_t *tb = reinterpret_cast<uint32_t *>(p.tours16 + ((size_t)b * A + abase) * FL);
      for (int e = threadIdx.x; e < nant * (FL / 2); e += 256) {
        const int k = e / (FL / 2), j = e - k * (FL / 2);
        tb[(size_t)k * (FL / 2) + j] = reinterpret_cast<const uint32_t *>(tour_s[k])[j];
      }
    }
    if (p.costs) {
      // tour lengths (tsp/aco.py:121-132): sum_k d[u_k][u_{k-1}], then the closing edge -- f32, that order.  64 edges of each
      // of the wave's four ants are gathered with every lane active and staged in the (dead) flag array.
      __syncthreads();
      const float *dist_b = p.dist + (size_t)b * p.dist_bs;
      // (LH: the flag array is four ants' worth; the head table is dead by now and holds the staging rows and the inverse table)
      float (*dstage)[APW][64] = reinterpret_cast<float (*)[APW][64]>(LH ? lh_tab : flag_mem);
      if (active) {
        float cost = 0.0f;
        const float *mine_d = dstage[wave][q];
        // (Round 6, last session, measured and not kept: four chunks' gathers in flight together -- sixteen loads per lane -- made
        // the headline launch SLOWER, 0.522 -> 0.532 ms on one box: the phase is bound by the L2's line rate, as DESIGN 9.1 says,
        // and deeper bursts only disturb the loops of the workgroups still building tours.)
        for (int base = 1; base < n; base += 64) {
          const int t = base + lane;
#pragma unroll
          for (int r4 = 0; r4 < APW; ++r4) {
            const uint16_t *tr = tour_s[wave * APW + r4];
            dstage[wave][r4][lane] = t < n ? dist_b[(uint32_t)tr[t] * (uint32_t)n + tr[t - 1]] : 0.0f;
          }
          __builtin_amdgcn_wave_barrier();
          if (s == 0) {
#pragma unroll
            for (int v4 = 0; v4 < 16; ++v4) {
              const float4 v = *(const float4 *)(mine_d + 4 * v4);
              cost = cost + v.x; cost = cost + v.y; cost = cost + v.z; cost = cost + v.w;
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
        if (s == 0 && a0 + q < A) {
          const uint16_t *tm = tour_s[wave * APW + q];
          cost = cost + dist_b[(uint32_t)tm[0] * (uint32_t)n + tm[n - 1]];
          p.costs[(size_t)b * A + a0 + q] = cost;
        }
      }
    }
    if (p.nbr) {
      // the update's table through an inverse-permutation table in the (dead) flag array, eight ants at a time
      uint16_t (*inv)[FL] = reinterpret_cast<uint16_t (*)[FL]>(LH ? lh_tab + 4096 : flag_mem);
      static_assert(LH || APB * FLP >= 8 * FL * (int)sizeof(uint16_t), "inverse table of eight ants inside the flag array");
      const int k8 = threadIdx.x & 7;
      for (int half = 0; half < 2; ++half) {
        const int nh = nant - half * 8 < 8 ? nant - half * 8 : 8;
        __syncthreads();
        if (nh <= 0) break;
        for (int e = threadIdx.x; e < 8 * FL / 8; e += 256) ((uint4 *)&inv[0][0])[e] = make_uint4(0, 0, 0, 0);
        __syncthreads();
        const uint16_t *tk = tour_s[half * 8 + (k8 < nh ? k8 : 0)];
        if (k8 < nh)
          for (int t = threadIdx.x >> 3; t < n; t += 32) inv[k8][tk[t]] = (uint16_t)t;
        __syncthreads();
        // classic layout [B][n][A]: one 32-byte run per node; grouped (nbr_grouped: [B][ceil(A/8)][n][8]): the eight ants' entries
        // of consecutive nodes are consecutive -- the 256 threads of a pass write 1 KB in one piece
        const int a8 = abase + half * 8;
        uint32_t *nb = p.nbr_grouped ? p.nbr + (((size_t)b * ((A + 7) >> 3) + (a8 >> 3)) * n) * 8 + (a8 & 7) : p.nbr + (size_t)b * n * A + a8;
        const size_t nstride = p.nbr_grouped ? 8 : (size_t)A;
        if (k8 < nh)
          for (int node = threadIdx.x >> 3; node < n; node += 32) {
            const int t = inv[k8][node];
            const uint32_t pv = tk[t == 0 ? n - 1 : t - 1], nx = tk[t == n - 1 ? 0 : t + 1];
            nb[(size_t)node * nstride + k8] = pv | (nx << 16);
          }
      }
    }
  } else {
    // TG: eight tours at a time, global -> LDS (this workgroup wrote them: its stores are visible after the barrier's release /
    // acquire), into the block the loop no longer needs: [8][FL] tours | [8][FL] inverse table (cost staging before it is built)
    uint16_t (*tl)[FL] = reinterpret_cast<uint16_t (*)[FL]>(sparse_dyn);
    uint16_t (*inv)[FL] = reinterpret_cast<uint16_t (*)[FL]>(sparse_dyn + 8 * FL * 2);
    const int k8 = threadIdx.x & 7;
    for (int half = 0; half < 2; ++half) {
      const int nh = nant - half * 8 < 8 ? nant - half * 8 : 8;
      __syncthreads();
      if (nh <= 0) break;
      {
        constexpr int V = FL * 2 / 16;                       // 16-byte pieces per tour
        const uint4 *src = reinterpret_cast<const uint4 *>(t16b + (size_t)half * 8 * FL);
        for (int e = threadIdx.x; e < 8 * V; e += 256)
          if (e / V < nh) reinterpret_cast<uint4 *>(&tl[0][0])[e] = src[e];
      }
      __syncthreads();
      if (p.paths && k8 < nh) {
        int64_t *pb = p.paths + (size_t)b * n * A + abase + half * 8;
        for (int t = threadIdx.x >> 3; t < n; t += 32) pb[(size_t)t * A + k8] = (int64_t)tl[k8][t];
      }
      if (p.costs) {
        // tour lengths (tsp/aco.py:121-132): sum_k d[u_k][u_{k-1}], then the closing edge -- f32, that order.  64 edges of
        // each of the wave's two ants are gathered with every lane active and staged in LDS.
        const float *dist_b = p.dist + (size_t)b * p.dist_bs;
        float (*dstage)[2][64] = reinterpret_cast<float (*)[2][64]>(&inv[0][0]);
        const int hh = lane >> 5;                             // lanes 0 and 32 sum the wave's two ants
        const int mine = wave * 2 + hh;
        float cost = 0.0f;
        for (int base = 1; base < n; base += 64) {
          const int t = base + lane;
#pragma unroll
          for (int r2 = 0; r2 < 2; ++r2) {
            const uint16_t *tr = tl[wave * 2 + r2];
            dstage[wave][r2][lane] = t < n && wave * 2 + r2 < nh ? dist_b[(uint32_t)tr[t] * (uint32_t)n + tr[t - 1]] : 0.0f;
          }
          __builtin_amdgcn_wave_barrier();
          if ((lane & 31) == 0) {
            const float *md = dstage[wave][hh];
#pragma unroll
            for (int v4 = 0; v4 < 16; ++v4) {
              const float4 v = *(const float4 *)(md + 4 * v4);
              cost = cost + v.x; cost = cost + v.y; cost = cost + v.z; cost = cost + v.w;
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
        if ((lane & 31) == 0 && mine < nh) {
          const uint16_t *tm = tl[mine];
          cost = cost + dist_b[(uint32_t)tm[0] * (uint32_t)n + tm[n - 1]];
          p.costs[(size_t)b * A + abase + half * 8 + mine] = cost;
        }
      }
      if (p.nbr) {
        __syncthreads();
        for (int e = threadIdx.x; e < 8 * FL / 8; e += 256) ((uint4 *)&inv[0][0])[e] = make_uint4(0, 0, 0, 0);
        __syncthreads();
        const uint16_t *tk = tl[k8 < nh ? k8 : 0];
        if (k8 < nh)
          for (int t = threadIdx.x >> 3; t < n; t += 32) inv[k8][tk[t]] = (uint16_t)t;
        __syncthreads();
        // classic layout [B][n][A]: one 32-byte run per node; grouped (nbr_grouped: [B][ceil(A/8)][n][8]): the eight ants' entries
        // of consecutive nodes are consecutive -- the 256 threads of a pass write 1 KB in one piece
        const int a8 = abase + half * 8;
        uint32_t *nb = p.nbr_grouped ? p.nbr + (((size_t)b * ((A + 7) >> 3) + (a8 >> 3)) * n) * 8 + (a8 & 7) : p.nbr + (size_t)b * n * A + a8;
        const size_t nstride = p.nbr_grouped ? 8 : (size_t)A;
        if (k8 < nh)
          for (int node = threadIdx.x >> 3; node < n; node += 32) {
            const int t = inv[k8][node];
            const uint32_t pv = tk[t == 0 ? n - 1 : t - 1], nx = tk[t == n - 1 ? 0 : t + 1];
            nb[(size_t)node * nstride + k8] = pv | (nx << 16);
          }
      }
    }
  }
}

}  // namespace daco

using namespace daco;

// tau^alpha and eta^beta for exponents other than 1 (x^2 = x x and x^0 = 1 exactly, as torch computes them; powf otherwise):
// the head-row kernels then run on these two matrices with unit exponents -- the same products tau^alpha * eta^beta
__global__ void __launch_bounds__(256)
pow_pair_kernel(long count_t, const float *tau, float alpha, float *tau_out, long count_e, const float *eta, float beta, float *eta_out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < count_t) tau_out[i] = pw(tau[i], alpha);
  if (i < count_e) eta_out[i] = pw(eta[i], beta);
}

// workspace: the dense rows P [B][n][ld], the head rows (daco_pheromone_update_heads writes both), then the u16 tours [B][A][ld]
// (n > 512: as they are built; n <= 512: written by a call that asks for no int64 paths)
extern "C" size_t daco_tsp_sparse_tours_offset(int B, int n, int A) {
  if (B <= 0 || A <= 0 || n <= 128 || n > 1024) return 0;
  const int ld = n <= 512 ? 512 : 1024;                  // (the row walks of the kernel's two instantiations read 512 / 1024 candidates)
  return align256((size_t)B * n * ld * sizeof(float)) + align256((size_t)B * n * sp_head_row_bytes(SP_KH_MAX / 16));
}
extern "C" size_t daco_tsp_sparse_workspace_bytes(int B, int n, int A) {
  if (B <= 0 || A <= 0 || n <= 128 || n > 1024) return 0;
  const int ld = n <= 512 ? 512 : 1024;
  return daco_tsp_sparse_tours_offset(B, n, A) + align256(((size_t)B * A + 16) * ld * sizeof(uint16_t));
}

// ... and, for exponents other than 1, tau^alpha [B][n][n] and eta^beta [B or 1][n][n] behind it
extern "C" size_t daco_tsp_sparse_workspace_bytes_general(int B, int n, int A) {
  const size_t base = daco_tsp_sparse_workspace_bytes(B, n, A);
  return base ? base + 2 * align256((size_t)B * n * n * sizeof(float)) : 0;
}

static bool sparse_rows_vec4(int n, const float *tau, long tau_bstride, const float *eta, long eta_bstride) {
  return (n & 3) == 0 && (tau_bstride & 3) == 0 && (eta_bstride & 3) == 0 && (((uintptr_t)tau | (uintptr_t)eta) & 15) == 0;
}

static int sample_sparse_impl(bool race, bool heads_ready, int head_live_max, int nbr_grouped, const char *what, void *stream, int B, int n, int A, const float *tau, long tau_bstride, const float *eta,
                                      long eta_bstride, float alpha, float beta, const uint16_t *head_id, int head_slots, const int64_t *start,
                                      int fixed_start, uint64_t seed, uint64_t iter, const uint64_t *iter_offset,
                                      uint32_t ant_gid0, int ant_gid_bstride, int64_t *paths, int32_t *flags, const float *dist,
                                      long dist_bstride, float *costs, uint32_t *nbr, unsigned long long *stats, void *workspace,
                                      size_t workspace_bytes, void *ev_begin, void *ev_end) {
  if (B <= 0 || A <= 0 || !tau || !eta || !head_id || !workspace || (!paths && !nbr)) {
    set_error("%s: bad argument (B=%d n=%d A=%d)", what, B, n, A);
    return DACO_E_BADARG;
  }
  if (head_slots != 64 && head_slots != 128) { set_error("%s: head_slots = %d (64 or 128)", what, head_slots); return DACO_E_BADARG; }
  if (n <= 128 || n > 1024) { set_error("%s: n=%d outside 129..1024 (the dense samplers serve the other sizes)", what, n); return DACO_E_TOOLARGE; }
  if ((size_t)n * A * 8 >= ((size_t)1 << 32)) { set_error("%s: n * A too large for 32-bit offsets", what); return DACO_E_TOOLARGE; }
  if (fixed_start >= n) { set_error("%s: fixed_start %d >= n %d", what, fixed_start, n); return DACO_E_BADARG; }
  if (costs && !dist) { set_error("%s: fused costs need the distance matrix", what); return DACO_E_BADARG; }
  const size_t need = daco_tsp_sparse_workspace_bytes(B, n, A);
  if (workspace_bytes < need) { set_error("%s: workspace %zu < %zu bytes", what, workspace_bytes, need); return DACO_E_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  const int ld = n <= 512 ? 512 : 1024;
  float *P = (float *)workspace;
  char *hrow = (char *)workspace + align256((size_t)B * n * ld * sizeof(float));
  const int spl = head_slots / 16;
  if (alpha != 1.0f || beta != 1.0f) {
    // the kernels take unit exponents: the powers are applied to the matrices first, into the tail of a `general` workspace
    const size_t need_g = daco_tsp_sparse_workspace_bytes_general(B, n, A);
    if (workspace_bytes < need_g) {
      set_error("%s: alpha = %g, beta = %g need daco_tsp_sparse_workspace_bytes_general() = %zu bytes of workspace (got %zu)", what, alpha, beta, need_g, workspace_bytes);
      return DACO_E_WORKSPACE;
    }
    if (heads_ready) { set_error("%s: heads_ready needs alpha = beta = 1 (daco_pheromone_update_heads forms the rows of tau itself)", what); return DACO_E_BADARG; }
    float *tp = (float *)((char *)workspace + need), *ep = tp + align256((size_t)B * n * n * sizeof(float)) / sizeof(float);
    if ((tau_bstride != 0 && tau_bstride != (long)n * n) || (eta_bstride != 0 && eta_bstride != (long)n * n)) {
      set_error("%s: exponents other than 1 need dense matrices (stride n * n between instances, or 0 for a shared one)", what);
      return DACO_E_BADARG;
    }
    const long ct = (tau_bstride ? (long)B : 1L) * n * n, ce = (eta_bstride ? (long)B : 1L) * n * n;
    const long cm = ct > ce ? ct : ce;
    hipLaunchKernelGGL(pow_pair_kernel, dim3((unsigned)((cm + 255) / 256)), dim3(256), 0, s, ct, tau, alpha, tp, ce, eta, beta, ep);
    tau = tp; eta = ep;
    alpha = beta = 1.0f;
  }
  const bool vec4 = sparse_rows_vec4(n, tau, tau_bstride, eta, eta_bstride);
  if (!heads_ready) {
    const dim3 pg((unsigned)(((long)B * n + 3) / 4));
#define DACO_PREPASS_C(R, V, C) hipLaunchKernelGGL((sparse_prepass_kernel<R, V, C>), pg, dim3(256), 0, s, B, n, ld / 256, tau, tau_bstride, eta, eta_bstride, \
                                                   head_id, P, hrow, spl, ld)
#define DACO_PREPASS(R, V) do { if (ld <= 512) DACO_PREPASS_C(R, V, 2); else DACO_PREPASS_C(R, V, 4); } while (0)
    if (race) { if (vec4) DACO_PREPASS(true, true); else DACO_PREPASS(true, false); }
    else { if (vec4) DACO_PREPASS(false, true); else DACO_PREPASS(false, false); }
#undef DACO_PREPASS
#undef DACO_PREPASS_C
  }
  SampleParams sp{};
  sp.B = B; sp.n = n; sp.A = A; sp.ld = ld; sp.CH = ld / 256;
  sp.P = P; sp.start = start; sp.fixed_start = fixed_start;
  sp.tau = tau; sp.tau_bs = tau_bstride; sp.eta = eta; sp.eta_bs = eta_bstride; sp.alpha = alpha; sp.beta = beta; sp.row_vec = vec4 ? 1 : 0;
  sp.seed = seed; sp.iter = iter; sp.iter_dev = iter_offset; sp.ant_gid0 = ant_gid0; sp.gid_bstride = ant_gid_bstride;
  sp.paths = paths; sp.flags = flags; sp.dist = dist; sp.dist_bs = dist_bstride; sp.costs = costs; sp.nbr = nbr; sp.nbr_grouped = nbr_grouped ? 1 : 0;
  sp.hval = (const float *)hrow; sp.hid = head_id; sp.stats = stats;
  sp.tours16 = (uint16_t *)(hrow + align256((size_t)B * n * sp_head_row_bytes(SP_KH_MAX / 16)));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("%s pre-pass: %s", what, hipGetErrorString(e)); return DACO_E_HIP; }
  if (ev_begin && hipEventRecord((hipEvent_t)ev_begin, s) != hipSuccess) { set_error("hipEventRecord(ev_begin) failed"); return DACO_E_HIP; }
  // few ants (no more workgroups of four than the chip has CUs) and a head table that fits a CU's LDS: the LDS-heads variant
  static const int lh_knob = getenv("DACO_SPARSE_LDS_HEADS") ? atoi(getenv("DACO_SPARSE_LDS_HEADS")) : 1;   // (0: never; measurement knob)
  bool lh = false;
  size_t lh_lds = 0;
  if (lh_knob && !race && spl == 4 && ld == 512 && head_live_max > 0 && head_live_max <= 62 && (long)B * ((A + 3) / 4) <= 256) {
    const int kl = (head_live_max + 1 + 3) / 4;
    const size_t table = (size_t)n * kl * sp_lane_bytes(4) + 32;
    lh_lds = (size_t)4 * (512 + 16) + (size_t)4 * (512 + 2) * 2 + (table > 12288 + 1024 ? table : 12288 + 1024);
    lh = lh_lds + 128 <= 160 * 1024;                     // (+ the kernel's static words)
    sp.lh_kl = kl; sp.lh_kmax = head_live_max;
  }
  const int bpi = lh ? (A + 3) / 4 : (A + 15) / 16;
  const dim3 grid((unsigned)(B * bpi));
  // dynamic LDS: flags + tours (n <= 512); the larger of flags + window and eight tours + their inverse table (n > 512)
  const int pad_lds = getenv("DACO_SPARSE_PAD_LDS") ? atoi(getenv("DACO_SPARSE_PAD_LDS")) : 0;   // (measurement knob: fewer workgroups per CU)
#define DACO_SPARSE_LDS(C) ((C) == 2 ? 16 * ((C) * 256 + 16) + 16 * ((C) * 256 + 2) * 2 : 2 * 8 * (C) * 256 * 2)
#define DACO_SPARSE_LAUNCH(C, R, S) hipLaunchKernelGGL((scan_sparse_kernel<C, R, S>), grid, dim3(256), DACO_SPARSE_LDS(C) + pad_lds, s, sp)
#define DACO_SPARSE_PICK(C, R) do { if (spl == 4) DACO_SPARSE_LAUNCH(C, R, 4); else DACO_SPARSE_LAUNCH(C, R, 8); } while (0)
  if (lh) {
    // (more than 64 KB of dynamic LDS has to be asked for; per call: the attribute belongs to the current device's copy of the
    // kernel and a process may drive several devices)
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_sparse_kernel<2, false, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024 - 128) != hipSuccess) { set_error("%s: hipFuncSetAttribute(dynamic LDS) failed", what); return DACO_E_HIP; }
    hipLaunchKernelGGL((scan_sparse_kernel<2, false, 4, true>), grid, dim3(256), lh_lds, s, sp);
  } else if (ld <= 512) { if (race) DACO_SPARSE_PICK(2, true); else DACO_SPARSE_PICK(2, false); }
  else { if (race) DACO_SPARSE_PICK(4, true); else DACO_SPARSE_PICK(4, false); }
#undef DACO_SPARSE_PICK
#undef DACO_SPARSE_LDS
#undef DACO_SPARSE_LAUNCH
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("scan_sparse_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  if (ev_end && hipEventRecord((hipEvent_t)ev_end, s) != hipSuccess) { set_error("hipEventRecord(ev_end) failed"); return DACO_E_HIP; }
  return DACO_OK;
}

#define DACO_SPARSE_ARGS stream, B, n, A, tau, tau_bstride, eta, eta_bstride, alpha, beta, head_id, head_slots, start, fixed_start, seed, iter, \
                         iter_offset, ant_gid0, ant_gid_bstride, paths, flags, dist, dist_bstride, costs, nbr, stats, workspace, \
                         workspace_bytes, ev_begin, ev_end
extern "C" int daco_tsp_sample_sparse(void *stream, int B, int n, int A, const float *tau, long tau_bstride, const float *eta,
                                      long eta_bstride, float alpha, float beta, const uint16_t *head_id, int head_slots, const int64_t *start,
                                      int fixed_start, uint64_t seed, uint64_t iter, const uint64_t *iter_offset,
                                      uint32_t ant_gid0, int ant_gid_bstride, int64_t *paths, int32_t *flags, const float *dist,
                                      long dist_bstride, float *costs, uint32_t *nbr, unsigned long long *stats, void *workspace,
                                      size_t workspace_bytes, void *ev_begin, void *ev_end) {
  return sample_sparse_impl(false, false, 0, 0, "daco_tsp_sample_sparse", DACO_SPARSE_ARGS);
}
extern "C" int daco_tsp_sample_race_head(void *stream, int B, int n, int A, const float *tau, long tau_bstride, const float *eta,
                                         long eta_bstride, float alpha, float beta, const uint16_t *head_id, int head_slots, const int64_t *start,
                                         int fixed_start, uint64_t seed, uint64_t iter, const uint64_t *iter_offset,
                                         uint32_t ant_gid0, int ant_gid_bstride, int64_t *paths, int32_t *flags, const float *dist,
                                         long dist_bstride, float *costs, uint32_t *nbr, unsigned long long *stats, void *workspace,
                                         size_t workspace_bytes, void *ev_begin, void *ev_end) {
  return sample_sparse_impl(true, false, 0, 0, "daco_tsp_sample_race_head", DACO_SPARSE_ARGS);
}
extern "C" int daco_tsp_sample_heads(void *stream, int race, int heads_ready, int head_live_max, int nbr_grouped, int B, int n, int A, const float *tau, long tau_bstride, const float *eta,
                                     long eta_bstride, float alpha, float beta, const uint16_t *head_id, int head_slots, const int64_t *start,
                                     int fixed_start, uint64_t seed, uint64_t iter, const uint64_t *iter_offset,
                                     uint32_t ant_gid0, int ant_gid_bstride, int64_t *paths, int32_t *flags, const float *dist,
                                     long dist_bstride, float *costs, uint32_t *nbr, unsigned long long *stats, void *workspace,
                                     size_t workspace_bytes, void *ev_begin, void *ev_end) {
  return sample_sparse_impl(race != 0, heads_ready != 0, head_live_max, nbr_grouped, "daco_tsp_sample_heads", DACO_SPARSE_ARGS);
}
