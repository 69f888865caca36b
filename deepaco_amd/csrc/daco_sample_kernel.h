// daco_sample_kernel.h -- the tour-construction kernel template shared by daco_tsp_sample.hip (TSP, CVRP,
// step-wise draws) and daco_sib_sample.hip (fused sibling-problem constructions).  See daco_tsp_sample.hip
// for the design notes.
#pragma once
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "daco_device.h"
#include "../../include/deepaco_hip.h"

namespace daco {

struct SampleParams {
  int B, n, A, ld, CH;
  const float *P;      // [B][n][ld] fused transition weights (0 in padding)
  const float *R;      // [B][n][ld] 1/P (+inf in padding) -- RACE_PHILOX only
  int norm_passes;
  const int64_t *start;  // [B][A] or null
  int fixed_start;
  const float *noise;    // [B][n-1][A][n] RACE_NOISE
  uint64_t seed, iter;
  int gid_bstride;           // ant-id stride between instances (0 = A): a rank that builds a slice of a colony's ants keeps the colony's ids
  const uint64_t *iter_dev;  // optional device-side addend to iter (lets a captured HIP graph advance the RNG); or null
  uint32_t ant_gid0;
  int64_t *paths;        // [B][n][A]
  float *logp;           // [B][n-1][A] or null
  float *rowsum;         // [B][n-1][A] or null
  int32_t *flags;        // [B] or null
  const float *dist;     // [B][n][n] (fused costs) or null
  long dist_bs;
  float *costs;          // [B][A] or null
  uint32_t *nbr;         // [B][n][A] prev | next << 16 per (node, ant), for the pheromone update; or null
  uint32_t *hubmask;     // CVRP: [B][A][ceil(n/32)] set of nodes that follow the depot (with nbr); or null
  int32_t *tab_lens;     // CVRP: [B][A] copy of lens inside the successor table (with nbr); or null
  // CVRP (cvrp/aco.py:138-205): node 0 = depot, variable-length routes
  // fused sibling constructions (daco_sib_sample.hip)
  const float *aux_vec;  // [B][n]: SOP predecessor counts, PCTSP prizes, OP distance to the depot d[k][0]
  const float *aux_mat;  // [B][n][ld] padded: SOP "who waits for k" rows, OP distance rows
  float scalar0;         // PCTSP min prize, OP max length, MKP capacity
  const float *wts;      // MKP [B][n][m] item weights
  int m;                 // MKP number of knapsack dimensions (<= 8)
  const float *mask;     // PROB_STEP: [B][A][n] f32, 0 = closed
  int step;              // PROB_STEP: step index (RNG counter word)
  const float *demand;   // [B][n]
  float capacity;
  const double *demand64;  // PROB_CVRP64: [B][n] float64 demands (cvrp_nls/ keeps its data in double) and capacity
  double capacity64;
  int Lmax;              // rows of paths (and Lmax-1 rows of logp)
  int noise_steps;       // rows of the noise tensor
  int32_t *lens;         // [B][A] rows used by each ant
  // head / tail rows (daco_scan_sparse.hip)
  const float *hval = nullptr;       // [B][n] head rows of this iteration: 16 lanes x {SPL f32 values, SPL u16 ids} (sparse_prepass_kernel)
  const uint16_t *hid = nullptr;     // the caller's head table [B][n][slots] (the pre-pass reads it; the scan reads the head rows)
  unsigned long long *stats = nullptr;   // [3] dense steps, tail walks, rejections (tests) or null
  uint16_t *tours16 = nullptr;       // scan_sparse at n > 512: [B][A][ld] the tours as they are built (32 bytes per ant every 16 steps)
  // the rows the rare ways of scan_sparse walk: tau^alpha * eta^beta formed from the caller's tensors (no dense copy since round 6)
  const float *tau = nullptr, *eta = nullptr;
  long tau_bs = 0, eta_bs = 0;
  float alpha = 1.0f, beta = 1.0f;
  int row_vec = 0;                   // rows can be read as aligned 16-byte vectors (n % 4 == 0, aligned bases and strides)
  int nbr_grouped = 0;               // scan_sparse: nbr as [B][ceil(A/8)][n][8] (eight ants' entries of a node together) instead of [B][n][A]
  int lh_kl = 0, lh_kmax = 0;        // scan_sparse, LDS-heads variant: lane records per row kept in LDS, the caller's bound of live slots per row
};

template <class F, int... I>
__device__ inline void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
// compile-time loop: f receives std::integral_constant<int, j>, j = 0..N-1
template <int N, class F>
__device__ inline void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int VEC>
__device__ inline void load_vec(const float *p, float (&out)[VEC]) {
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4 *>(p);
    out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w;
  } else if constexpr (VEC == 2) {
    const float2 t = *reinterpret_cast<const float2 *>(p);
    out[0] = t.x; out[1] = t.y;
  } else {
    out[0] = *p;
  }
}

// P = tau^alpha * eta^beta with zero padding; R = 1/P with +inf padding (optional): launches prob_matrix_kernel
// (daco_tsp_sample.hip)
void launch_prob_matrix(int B, int n, int ld, const float *tau, long tau_bs, const float *eta, long eta_bs, float alpha,
                        float beta, float *P, float *R, hipStream_t s);

// visited bitset: bit (c*VEC+v) of a 64-bit word kept as two 32-bit halves so every test is a
// single 32-bit v_and/v_cmp (the upper half folds away when CH*VEC <= 32)
struct Visited {
  uint32_t lo = 0, hi = 0;
  template <int BIT> __device__ inline bool test() const {
    if constexpr (BIT < 32) return (lo >> BIT) & 1u; else return (hi >> (BIT - 32)) & 1u;
  }
  __device__ inline void set(int bit) {       // bit is wave-uniform
    if (bit < 32) lo |= 1u << bit; else hi |= 1u << (bit - 32);
  }
  // x if candidate BIT is open, +0.0f if closed -- two VALU ops (v_bfe_i32 + v_bfi_b32); written as
  // asm because the optimiser otherwise rewrites it into and + cmp + cndmask
  template <int BIT> __device__ inline float open_only(float x) const {
    int m;
    float r;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(BIT < 32 ? lo : hi), "n"(BIT & 31));
    asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(r) : "v"(m), "v"(x));
    return r;
  }
  template <int BIT> __device__ inline void set_if(bool c) {
    if constexpr (BIT < 32) lo |= (c ? 1u : 0u) << BIT; else hi |= (c ? 1u : 0u) << (BIT - 32);
  }
};

enum { PROB_TSP = 0, PROB_CVRP = 1, PROB_STEP = 2, PROB_SOP = 3, PROB_PCTSP = 4, PROB_OP = 5, PROB_MKP = 6, PROB_CVRP64 = 7 };

// PROB_TSP: whole closed tour; PROB_CVRP: whole capacity-constrained route sequence; PROB_CVRP64: the same with the load
// bookkeeping of cvrp_nls/aco.py:254-272 in float64 (used = used + demand[cur]; demand > capacity - used, all double there:
// with demands k/50 a customer that fits exactly is common, and whether it passes is a matter of the last bit);
// PROB_STEP: ONE draw per ant from an externally maintained mask (ACO.pick_move for the sibling
// problems, whose feasibility logic stays with the caller).
template <int VEC, int CH, int MODE, bool LOGP, int PROB>
__global__ void __launch_bounds__(256)
tsp_sample_kernel(const SampleParams p) {
  constexpr bool CVRP = PROB == PROB_CVRP || PROB == PROB_CVRP64, DEM64 = PROB == PROB_CVRP64, STEP = PROB == PROB_STEP, SOP = PROB == PROB_SOP,
                 PCTSP = PROB == PROB_PCTSP, OP = PROB == PROB_OP, MKP = PROB == PROB_MKP;
  constexpr bool VARLEN = CVRP || PCTSP || OP || MKP;   // solution length differs between ants
  constexpr bool DUMMY = OP || MKP;                     // last node = absorbing dummy, never drawn
  constexpr int NJ = CH * VEC;                          // candidates per lane
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int bpi = (p.A + 3) >> 2;                       // workgroups per instance (4 ants each)
  const int b = w / bpi;
  const int a_raw = (w - b * bpi) * 4 + wave;
  // PROB_TSP ends with a workgroup epilogue (barriers): a wave without an ant builds ant A-1 again and writes nothing
  constexpr bool EPI = PROB == PROB_TSP;                // tour kept in LDS, outputs written by the workgroup at the end
  const bool dup = EPI && a_raw >= p.A;
  if (a_raw >= p.A && !dup) return;                     // (the other problems have no barrier below: safe)
  const int a = dup ? p.A - 1 : a_raw;
  extern __shared__ __attribute__((aligned(16))) uint16_t epi_lds[];   // EPI: [4][TL] tours | [4][TL] inverse | [4][64] f32
  const int TL = (p.n + 7) & ~7;
  uint16_t *tour = epi_lds + (size_t)wave * TL;
  const int n = p.n, A = p.A, ld = p.ld;
  const uint64_t iter_now = p.iter + (p.iter_dev ? *p.iter_dev : 0ull);   // a captured graph advances *iter_dev
  const uint32_t gid = p.ant_gid0 + (uint32_t)(b * (p.gid_bstride ? p.gid_bstride : A) + a);
  const float *Pb = p.P + (size_t)b * n * ld + lane * VEC;
  const float *Rb = (MODE == DACO_RACE_PHILOX) ? p.R + (size_t)b * n * ld + lane * VEC : nullptr;
  const int rows = VARLEN ? p.Lmax : (STEP ? 2 : n);     // rows of paths for one instance (STEP: logp has 1 row)
  int64_t *path_out = p.paths + (STEP ? (size_t)b * A : (size_t)b * rows * A) + a;
  float *logp_out = LOGP ? p.logp + (size_t)b * (rows - 1) * A + a : nullptr;
  float *rs_out = (LOGP && p.rowsum) ? p.rowsum + (size_t)b * (rows - 1) * A + a : nullptr;
  const float *dist_b = (CVRP && p.costs) ? p.dist + (size_t)b * p.dist_bs : nullptr;      // (TSP: in the epilogue)
  uint32_t *nbr_a = nullptr;                            // (TSP: the neighbour table is written by the epilogue)
  int pprev = 0, second = 0;                            // neighbour-table bookkeeping
  const float *demand_b = (CVRP && !DEM64) ? p.demand + (size_t)b * n : nullptr;
  const double *demand64_b = DEM64 ? p.demand64 + (size_t)b * n : nullptr;

  // ---- start node
  int prev;
  if constexpr (CVRP || SOP || PCTSP || OP) prev = 0;
  else if constexpr (MKP) {
    if (p.start) prev = (int)p.start[(size_t)b * A + a];
    else { const u32x4 r = rng_block(p.seed, iter_now, STREAM_START, gid, 0); prev = (int)__umulhi(r.x, (uint32_t)(n - 1)); }
  }
  else if (p.start) prev = (int)p.start[(size_t)b * A + a];
  else if (p.fixed_start >= 0) prev = p.fixed_start;
  else {
    const u32x4 r = rng_block(p.seed, iter_now, STREAM_START, gid, 0);
    prev = (int)__umulhi(r.x, (uint32_t)n);
  }
  prev = __builtin_amdgcn_readfirstlane(prev);
  const int first = prev;

  Visited vis;
  auto mark = [&](int k) {                              // k wave-uniform, >= 0
    const unsigned vi = (unsigned)k / VEC;
    const unsigned bit = (vi >> 6) * VEC + ((unsigned)k % VEC);
    if ((unsigned)lane == (vi & 63u)) vis.set((int)bit);
  };
  int own_lane = -1, own_bit = 0;                       // SCAN: owner of the last choice, known without division
  if constexpr (PROB == PROB_TSP || SOP || OP || MKP) mark(prev);
  if (lane == 0 && !STEP) { if constexpr (EPI) tour[0] = (uint16_t)prev; else path_out[0] = prev; }

  // per-candidate constants / counters of the constrained problems (this lane's candidates):
  //   CVRP demand, SOP number of unvisited predecessors, OP distance back to the depot
  float dem[CH][VEC];
  double dem64[DEM64 ? CH : 1][VEC];                    // PROB_CVRP64: the demands and the load in double
  double used64 = 0.0;
  int remaining = n - 1;                                // CVRP/PCTSP: customers / nodes not yet visited
  float used = 0.0f;                                    // CVRP load on the route; PCTSP prize; OP length
  bool finished = false;
  Visited sticky;                                       // OP/MKP: candidates closed for good
  Visited regular;                                      // DUMMY problems: bits of the real candidates (k < n-1)
  float knap[8] = {0, 0, 0, 0, 0, 0, 0, 0};             // MKP loads
  const float *avec = (SOP || PCTSP || OP) ? p.aux_vec + (size_t)b * n : nullptr;
  const float *amat = (SOP || OP) ? p.aux_mat + (size_t)b * n * ld + lane * VEC : nullptr;
  const float *wts_b = MKP ? p.wts + (size_t)b * n * p.m : nullptr;
  if constexpr (CVRP) {
    static_for<NJ>([&](auto J) {
      constexpr int j = J, c = j / VEC, v = j % VEC;
      const int k = (c * 64 + lane) * VEC + v;
      if constexpr (DEM64) dem64[c][v] = k < n ? demand64_b[k] : (double)__builtin_inff();
      else dem[c][v] = k < n ? demand_b[k] : __builtin_inff();
    });
    if constexpr (DEM64) used64 = used64 + demand64_b[0];
    else used = used + demand_b[0];
  }
  if constexpr (SOP) {                                  // sop/aco.py:118-126: node 0 is visited first
    float r0[CH][VEC];
#pragma unroll
    for (int c = 0; c < CH; ++c) load_vec<VEC>(amat + c * 64 * VEC, r0[c]);
    static_for<NJ>([&](auto J) {
      constexpr int j = J, c = j / VEC, v = j % VEC;
      const int k = (c * 64 + lane) * VEC + v;
      dem[c][v] = k < n ? avec[k] - r0[c][v] : 1.0f;
    });
  }
  if constexpr (OP) {
    static_for<NJ>([&](auto J) {
      constexpr int j = J, c = j / VEC, v = j % VEC;
      const int k = (c * 64 + lane) * VEC + v;
      dem[c][v] = k < n ? avec[k] : 0.0f;
    });
  }
  if constexpr (DUMMY) {
    static_for<NJ>([&](auto J) {
      constexpr int j = J, c = j / VEC, v = j % VEC;
      regular.template set_if<j>((c * 64 + lane) * VEC + v < n - 1);
    });
  }
  if constexpr (MKP) {
#pragma unroll
    for (int dd = 0; dd < 8; ++dd) if (dd < p.m) knap[dd] = wts_b[(size_t)prev * p.m + dd];
  }

  u32x4 ublk = {0, 0, 0, 0};                            // SCAN: 256 cached uniforms per wave
  uint32_t ucur = 0;
  bool infeasible = false, overflow = false;
  float cost = 0.0f, dpend = 0.0f;                      // fused tour length (edge added one step late)

  const int t0 = STEP ? p.step : 1;                     // STEP: the caller's step index keys the RNG
  int t = t0;
  if constexpr (CVRP) finished = remaining == 0;        // (a depot-only instance)
  for (; VARLEN ? (t < p.Lmax && !finished) : (STEP ? t == t0 : t < n); ++t) {
    // ---- candidates closed at this step: visited, plus the problem's own feasibility rules
    Visited blk = vis;
    if constexpr (SOP) {                                 // a node opens when its last predecessor is visited
      static_for<NJ>([&](auto J) {
        constexpr int j = J;
        blk.template set_if<j>(dem[j / VEC][j % VEC] != 0.0f);
      });
    }
    if constexpr (PCTSP) {                               // pctsp/aco.py:166-181: the depot opens once enough
      const bool depot_open = prev != 0 && (used > p.scalar0 || remaining == 0);   // prize is collected
      if (lane == 0 && !depot_open) blk.lo |= 1u;
    }
    if constexpr (OP) {                                  // op/aco.py:195-220: close what cannot get home in time
      float drow[CH][VEC];
#pragma unroll
      for (int c = 0; c < CH; ++c) load_vec<VEC>(amat + (unsigned)prev * (unsigned)ld + c * 64 * VEC, drow[c]);
      static_for<NJ>([&](auto J) {
        constexpr int j = J, c = j / VEC, v = j % VEC;
        const float reach = used + drow[c][v] + dem[c][v];
        sticky.template set_if<j>(reach > p.scalar0);
      });
    }
    if constexpr (MKP) {                                 // mkp/aco.py:163-183: close items that no longer fit
      static_for<NJ>([&](auto J) {
        constexpr int j = J, c = j / VEC, v = j % VEC;
        const int k = (c * 64 + lane) * VEC + v;
        bool over = false;
        if (k < n - 1) {
#pragma unroll
          for (int dd = 0; dd < 8; ++dd)
            if (dd < p.m) over = over || (knap[dd] + wts_b[(size_t)k * p.m + dd] > p.scalar0);
        }
        sticky.template set_if<j>(over);
      });
    }
    if constexpr (DUMMY) {
      blk.lo |= sticky.lo | ~regular.lo;                 // the dummy (and padding) is never a candidate
      blk.hi |= sticky.hi | ~regular.hi;
      // nothing left to add: the ant moves to the dummy node and stays (padding below)
      if (__ballot(((~blk.lo) | (~blk.hi)) != 0u) == 0) { finished = true; break; }
    }
    if constexpr (VARLEN && !CVRP) {
      if (MODE == DACO_RACE_NOISE && t - 1 >= p.noise_steps) { overflow = true; break; }
    }
    if constexpr (STEP) {                                // closed = the caller's mask is 0
      const float *mrow = p.mask + ((size_t)b * A + a) * n;
      static_for<NJ>([&](auto J) {
        constexpr int j = J, c = j / VEC, v = j % VEC;
        const int k = (c * 64 + lane) * VEC + v;
        blk.template set_if<j>(k < n ? mrow[k] == 0.0f : true);
      });
    }
    if constexpr (CVRP) {
      if (MODE == DACO_RACE_NOISE && t - 1 >= p.noise_steps) { overflow = true; break; }
      if constexpr (DEM64) {
        const double rem = p.capacity64 - used64;
        static_for<NJ>([&](auto J) {
          constexpr int j = J;
          blk.template set_if<j>(dem64[j / VEC][j % VEC] > rem);      // strict, in double: cvrp_nls/aco.py:267-270
        });
      } else {
        const float rem = p.capacity - used;
        static_for<NJ>([&](auto J) {
          constexpr int j = J;
          blk.template set_if<j>(dem[j / VEC][j % VEC] > rem);        // strict, cvrp/aco.py:200
        });
      }
      if (lane == 0 && prev == 0 && remaining > 0) blk.lo |= 1u;       // cvrp/aco.py:179
    }
    // ---- stream the row of `prev`
    float row[CH][VEC];
    const float *rp = (MODE == DACO_RACE_PHILOX ? Rb : Pb) + (unsigned)prev * (unsigned)ld;
#pragma unroll
    for (int c = 0; c < CH; ++c) load_vec<VEC>(rp + c * 64 * VEC, row[c]);

    int choice;
    float pchoice = 0.0f, S = 0.0f;

    if constexpr (MODE == DACO_SCAN) {
      // uniform for step t: lane (t&63), component (t>>6)&3 of the Philox block (t>>8)*64 + lane
      if ((t & 63) == 0 || t == t0) {
        if ((t & 255) == 0 || t == t0) ublk = rng_block(p.seed, iter_now, STREAM_SCAN, gid, (uint32_t)(((t >> 8) << 6) + lane));
        ucur = comp(ublk, (t >> 6) & 3);
      }
      const uint32_t ux = (uint32_t)readlane_i((int)ucur, t & 63);
      // masked candidates become +0.0f, so every later add is a no-op for them
      float part = 0.0f;
      float pre[NJ];                                      // running sums inside the lane (non-decreasing)
      static_for<NJ>([&](auto J) {
        constexpr int j = J, c = j / VEC, v = j % VEC;
        row[c][v] = blk.template open_only<j>(row[c][v]);
        part = j == 0 ? row[c][v] : part + row[c][v];     // (+0.0f + x == x: masked values are +0.0f, never -0.0f)
        pre[j] = part;
      });
      const float incl = wave_scan_add(part);
      S = readlane_f(incl, 63);
      float uval = u01(ux);
      if constexpr (PROB == PROB_TSP) { if (p.noise) uval = p.noise[((size_t)b * (p.n - 1) + (t - 1)) * p.A + a]; }   // injected uniforms (tests)
      float r = uval * S;
      r = r > 0.0f ? r : 1.401298464e-45f;               // keep r > 0 if u*S underflows
      const uint64_t m = __ballot(incl >= r && part > 0.0f);
      if (m == 0) { infeasible = true; choice = 0; own_lane = -1; }   // (node 0 is marked by index below)
      else {
        const int L = __builtin_ctzll(m);
        const float excl = L ? readlane_f(incl, L - 1) : 0.0f;
        // what is left to cover inside lane L; the lane's running sums are non-decreasing, so the
        // first index reaching it is the count of those still below it (branch-free)
        const float thr = r - excl;
        int cnt = 0;
        static_for<NJ>([&](auto J) { cnt += pre[J] < thr ? 1 : 0; });
        int jsel = readlane_i(cnt, L);
        if (jsel >= NJ) {
          // rounding: the lane's own sum fell short of r - excl although incl >= r -> last
          // candidate of the lane with p > 0 (rare; wave-uniform branch)
          int last = 0;
#pragma unroll
          for (int c = 0; c < CH; ++c) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) last = row[c][v] > 0.0f ? c * VEC + v : last;
          }
          jsel = readlane_i(last, L);
        }
        choice = (int)((((unsigned)jsel / VEC) * 64u + (unsigned)L) * VEC + ((unsigned)jsel % VEC));
        own_lane = L; own_bit = jsel;
        if constexpr (LOGP) pchoice = p.P[((size_t)b * n + prev) * ld + choice];
      }
    } else if constexpr (MODE == DACO_RACE_PHILOX) {
      float bk = __builtin_inff();
      int bi = 0x7fffffff;
      u32x4 r4{};
      static_for<NJ>([&](auto J) {
        constexpr int j = J, c = j / VEC, v = j % VEC;
        const int k = (c * 64 + lane) * VEC + v;
        // one Philox block serves candidates 4g..4g+3; a lane's VEC candidates share a block
        if (v == 0) r4 = rng_block(p.seed, iter_now, STREAM_RACE, gid, ((uint32_t)t << 12) | (uint32_t)(k >> 2));
        const float Lk = neg_log2_1m(u01(comp(r4, k & 3)));
        const float key = blk.template test<j>() ? __builtin_inff() : Lk * row[c][v];
        if (key < bk) { bk = key; bi = k; }
      });
      const KeyIdx r = wave_arg<false>(bk, bi);
      if (!(r.key < __builtin_inff())) { infeasible = true; choice = 0; }
      else choice = r.idx;
      if constexpr (LOGP) {
        const float *pp = Pb + (size_t)prev * ld;
        float part = 0.0f;
        float pr[CH][VEC];
#pragma unroll
        for (int c = 0; c < CH; ++c) load_vec<VEC>(pp + c * 64 * VEC, pr[c]);
        static_for<NJ>([&](auto J) {
          constexpr int j = J;
          part = part + (blk.template test<j>() ? 0.0f : pr[j / VEC][j % VEC]);
        });
        S = wave_sum(part);
        pchoice = p.P[((size_t)b * n + prev) * ld + choice];
      }
    } else {  // DACO_RACE_NOISE: the arithmetic of torch.multinomial's one-sample path
      const float *q = p.noise + (((size_t)b * (VARLEN ? p.noise_steps : (STEP ? 1 : n - 1)) + (t - t0)) * A + a) * n;
      float part = 0.0f;
      static_for<NJ>([&](auto J) {
        constexpr int j = J, c = j / VEC, v = j % VEC;
        row[c][v] = blk.template open_only<j>(row[c][v]);
        part = j == 0 ? row[c][v] : part + row[c][v];     // (+0.0f + x == x: masked values are +0.0f, never -0.0f)
      });
      float S0 = 0.0f;                                    // un-normalised row sum (for backward)
      if (LOGP && p.norm_passes > 0) S0 = wave_sum(part);
      for (int pass = 0; pass < p.norm_passes; ++pass) {
        S = wave_sum(part);
        part = 0.0f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            row[c][v] = row[c][v] / S;
            part = part + row[c][v];
          }
        }
      }
      float bk = -__builtin_inff(), bp = 0.0f;
      int bi = 0x7fffffff;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const int k = (c * 64 + lane) * VEC + v;
          if (k < n) {
            const float key = row[c][v] / q[k];
            if (key > bk) { bk = key; bi = k; bp = row[c][v]; }
          }
        }
      }
      const KeyIdx r = wave_arg<true>(bk, bi);
      if (!(r.key > 0.0f)) infeasible = true;
      choice = r.idx == 0x7fffffff ? 0 : r.idx;
      if constexpr (LOGP) {
        const int own = (choice / VEC) & 63;
        pchoice = readlane_f(bp, own);
        if (p.norm_passes == 0) S = wave_sum(part); else S = S0;
      }
    }

    choice = __builtin_amdgcn_readfirstlane(choice);
    if constexpr (LOGP) {
      if (lane == 0) {
        // (noise mode with normalisation passes: pchoice is already the normalised probability)
        const float pr = (MODE == DACO_RACE_NOISE && p.norm_passes > 0) ? pchoice : pchoice / S;
        logp_out[(size_t)(t - t0) * A] = clamp_log(pr);
        if (rs_out) rs_out[(size_t)(t - t0) * A] = S;
      }
    }
    if constexpr (CVRP) {
      if (choice != 0) { mark(choice); --remaining; }
      else { used = 0.0f; used64 = 0.0; }
      if constexpr (DEM64) used64 = used64 + demand64_b[choice];
      else used = used + demand_b[choice];             // scalar load
      finished = remaining == 0 && choice == 0;
    } else if constexpr (SOP) {
      mark(choice);
      float rr[CH][VEC];
#pragma unroll
      for (int c = 0; c < CH; ++c) load_vec<VEC>(amat + (unsigned)choice * (unsigned)ld + c * 64 * VEC, rr[c]);
      static_for<NJ>([&](auto J) { constexpr int j = J; dem[j / VEC][j % VEC] = dem[j / VEC][j % VEC] - rr[j / VEC][j % VEC]; });
    } else if constexpr (PCTSP) {
      used = used + avec[choice];
      if (choice != 0) { mark(choice); --remaining; }
      finished = choice == 0;
    } else if constexpr (OP) {
      used = used + p.aux_mat[((size_t)b * n + prev) * ld + choice];
      mark(choice);
    } else if constexpr (MKP) {
      mark(choice);
#pragma unroll
      for (int dd = 0; dd < 8; ++dd) if (dd < p.m) knap[dd] = knap[dd] + wts_b[(size_t)choice * p.m + dd];
    } else if constexpr (PROB == PROB_TSP) {
      if (MODE == DACO_SCAN && own_lane >= 0) { if (lane == own_lane) vis.set(own_bit); }
      else mark(choice);
    }
    if (lane == 0) { if constexpr (EPI) tour[t] = (uint16_t)choice; else path_out[STEP ? 0 : (size_t)t * A] = choice; }
    if (dist_b) {                                        // fused gen_path_costs (wave-uniform)
      cost = cost + dpend;
      // TSP: d[u_t][u_{t-1}] (tsp/aco.py:127); CVRP: d[u_{t-1}][u_t] (cvrp/aco.py:135); scalar load
      dpend = CVRP ? dist_b[(unsigned)prev * (unsigned)n + (unsigned)choice] : dist_b[(unsigned)choice * (unsigned)n + (unsigned)prev];
    }
    if constexpr (CVRP) {
      if (p.nbr && lane == 0) {                          // who follows `prev` (depot: a set, one bit per successor)
        if (prev != 0) p.nbr[((size_t)b * n + prev) * A + a] = (uint32_t)choice << 16;
        else p.hubmask[((size_t)b * A + a) * ((n + 31) >> 5) + (choice >> 5)] |= 1u << (choice & 31);
      }
    }
    if (nbr_a) {                                         // node `prev` now knows both neighbours
      if (lane == 0) nbr_a[(size_t)prev * A] = (uint32_t)pprev | ((uint32_t)choice << 16);
      if (t == 1) second = choice;
      pprev = prev;
    }
    prev = choice;
  }
  if constexpr (VARLEN) {
    // the reference steps every ant until the slowest one is done: a done ant keeps drawing its
    // resting node (depot / dummy, probability 1), so its column is padded with it / log(1-eps)
    if (!finished) overflow = true;
    if (lane == 0) {
      if (p.lens) p.lens[(size_t)b * A + a] = t;
      if constexpr (CVRP) { if (p.tab_lens) p.tab_lens[(size_t)b * A + a] = t; }
      const float lp1 = clamp_log(1.0f);
      const int64_t rest = DUMMY ? n - 1 : 0;
      for (int tt = t; tt < p.Lmax; ++tt) {
        path_out[(size_t)tt * A] = rest;
        if constexpr (LOGP) logp_out[(size_t)(tt - 1) * A] = lp1;
      }
    }
    if (overflow && p.flags && lane == 0) atomicOr(p.flags + b, 2);
  }
  if (dist_b) {
    cost = cost + dpend;
    if constexpr (!CVRP) cost = cost + dist_b[(unsigned)first * (unsigned)n + (unsigned)prev];   // closing edge d[u_0][u_{n-1}] last
    if (lane == 0) p.costs[(size_t)b * A + a] = cost;
  }
  if (nbr_a && lane == 0) {                             // close the cycle: last -> first -> second
    if (n == 2) { nbr_a[(size_t)first * A] = (uint32_t)prev | ((uint32_t)prev << 16); nbr_a[(size_t)prev * A] = (uint32_t)first | ((uint32_t)first << 16); }
    else { nbr_a[(size_t)prev * A] = (uint32_t)pprev | ((uint32_t)first << 16); nbr_a[(size_t)first * A] = (uint32_t)prev | ((uint32_t)second << 16); }
  }
  if (infeasible && p.flags && lane == 0) atomicOr(p.flags + b, 1);
  if constexpr (EPI) {
    // ---- the workgroup's (up to) 4 tours leave LDS together (see tsp_scan32_kernel): paths in 32-byte runs per step
    // row, tour lengths from 64-edge gathers summed in step order by one lane, the neighbour table through an
    // inverse-permutation table in LDS.  Per-step stores / gathers with one active lane cost more than the row stream.
    __syncthreads();
    const int abase = (w - b * bpi) * 4;
    const int nant = A - abase < 4 ? A - abase : 4;
    const int k4 = threadIdx.x & 3;
    {
      int64_t *pb = p.paths + (size_t)b * n * A + abase;
      if (k4 < nant)
        for (int tt = threadIdx.x >> 2; tt < n; tt += 64) pb[(size_t)tt * A + k4] = (int64_t)epi_lds[(size_t)k4 * TL + tt];
    }
    if (p.costs && !dup) {
      const float *dist_e = p.dist + (size_t)b * p.dist_bs;
      float *stage = reinterpret_cast<float *>(epi_lds + (size_t)8 * TL) + wave * 64;
      float cost_e = 0.0f;
      for (int base = 1; base < n; base += 64) {
        const int tt = base + lane;
        stage[lane] = tt < n ? dist_e[(unsigned)tour[tt] * (unsigned)n + (unsigned)tour[tt - 1]] : 0.0f;   // d[u_t][u_{t-1}], tsp/aco.py:127
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
#pragma unroll
          for (int q4 = 0; q4 < 16; ++q4) {
            const float4 v = *reinterpret_cast<const float4 *>(stage + 4 * q4);     // (slots past the tour's end hold +0.0f)
            cost_e = cost_e + v.x; cost_e = cost_e + v.y; cost_e = cost_e + v.z; cost_e = cost_e + v.w;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (lane == 0) {
        cost_e = cost_e + dist_e[(unsigned)tour[0] * (unsigned)n + (unsigned)tour[n - 1]];    // closing edge d[u_0][u_{n-1}] last
        p.costs[(size_t)b * A + a] = cost_e;
      }
    }
    if (p.nbr) {
      uint16_t *inv = epi_lds + (size_t)4 * TL;
      for (int e = threadIdx.x; e < 4 * TL; e += 256) inv[e] = 0;
      __syncthreads();
      if (k4 < nant)
        for (int tt = threadIdx.x >> 2; tt < n; tt += 64) inv[(size_t)k4 * TL + epi_lds[(size_t)k4 * TL + tt]] = (uint16_t)tt;
      __syncthreads();
      uint32_t *nb = p.nbr + (size_t)b * n * A + abase;
      if (k4 < nant) {
        const uint16_t *tk = epi_lds + (size_t)k4 * TL;
        for (int node = threadIdx.x >> 2; node < n; node += 64) {
          const int tt = inv[(size_t)k4 * TL + node];
          const uint32_t pv = tk[tt == 0 ? n - 1 : tt - 1], nx = tk[tt == n - 1 ? 0 : tt + 1];
          nb[(size_t)node * A + k4] = pv | (nx << 16);
        }
      }
    }
  }
}

// ------------------------------------------------------------------ host dispatch
template <int VEC, int CH, int CVRP>
static hipError_t launch_sample(const SampleParams &sp, int mode, bool logp, hipStream_t s) {
  const int bpi = (sp.A + 3) / 4;
  dim3 grid((unsigned)(sp.B * bpi)), block(256);
  // PROB_TSP keeps the workgroup's tours (+ the inverse table and a staging row per wave) in LDS
  const size_t dyn = CVRP == PROB_TSP ? (size_t)8 * ((sp.n + 7) & ~7) * sizeof(uint16_t) + 4 * 64 * sizeof(float) : 0;
#define DACO_LAUNCH(M, L) do { \
    if (dyn > 64 * 1024) (void)hipFuncSetAttribute((const void *)tsp_sample_kernel<VEC, CH, M, L, CVRP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
    hipLaunchKernelGGL((tsp_sample_kernel<VEC, CH, M, L, CVRP>), grid, block, dyn, s, sp); } while (0)
  if (mode == DACO_SCAN) { if (logp) DACO_LAUNCH(DACO_SCAN, true); else DACO_LAUNCH(DACO_SCAN, false); }
  else if (mode == DACO_RACE_PHILOX) { if (logp) DACO_LAUNCH(DACO_RACE_PHILOX, true); else DACO_LAUNCH(DACO_RACE_PHILOX, false); }
  else { if (logp) DACO_LAUNCH(DACO_RACE_NOISE, true); else DACO_LAUNCH(DACO_RACE_NOISE, false); }
#undef DACO_LAUNCH
  return hipGetLastError();
}

template <int CVRP>
static hipError_t dispatch_sample(const SampleParams &sp, int vec, int CH, int mode, bool lp, hipStream_t s) {
  if (vec == 1) return launch_sample<1, 1, CVRP>(sp, mode, lp, s);
  if (vec == 2) return launch_sample<2, 1, CVRP>(sp, mode, lp, s);
  switch (CH) {
    case 1: return launch_sample<4, 1, CVRP>(sp, mode, lp, s);
    case 2: return launch_sample<4, 2, CVRP>(sp, mode, lp, s);
    case 3: return launch_sample<4, 3, CVRP>(sp, mode, lp, s);
    case 4: return launch_sample<4, 4, CVRP>(sp, mode, lp, s);
    case 6: return launch_sample<4, 6, CVRP>(sp, mode, lp, s);
    case 8: return launch_sample<4, 8, CVRP>(sp, mode, lp, s);
    case 12: return launch_sample<4, 12, CVRP>(sp, mode, lp, s);
    default: return launch_sample<4, 16, CVRP>(sp, mode, lp, s);
  }
}

// chunks per lane actually instantiated (compile-time loop bounds): the row is padded with
// zeros up to the next instantiated size; zero padding never changes a sum or a draw.
inline int inst_chunks(int n) {
  const int vec = vec_for_n(n), need = ld_for_n(n) / (64 * vec);
  static const int avail[] = {1, 2, 3, 4, 6, 8, 12, 16};
  for (int c : avail) if (c >= need) return c;
  return -1;
}
// DACO_LD_PAD (floats, multiple of 4; measurement knob): extra zero padding per row, to move the row stride off the
// 2 KB period (tools/l2_bw_shapes.hip measures +7 % L2 row rate at 2304 B) at the price of a larger L2 footprint
inline int ld_alloc(int n) {
  static const int pad = getenv("DACO_LD_PAD") ? atoi(getenv("DACO_LD_PAD")) & ~3 : 0;
  return inst_chunks(n) * 64 * vec_for_n(n) + pad;
}
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- shared by the several-ants-per-wavefront kernels (daco_tsp_scan32.hip, daco_scan16.hip)
constexpr int FCMP_OLT = 4;   // LLVM predicate for __builtin_amdgcn_fcmpf
// x + y + (the lane's bit of carry): one v_addc_co_u32 (the carry-out goes to a scratch SGPR pair).  s_nop 1: gfx950
// wants two wait states between a VALU write of an SGPR (the compare that made `carry`) and a VALU read of it, and the
// hazard recogniser does not look inside the asm.
__device__ inline int add_with_carry(int x, int y, uint64_t carry) {
  int d;
  uint64_t co;
  asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(d), "=&s"(co) : "v"(x), "v"(y), "s"(carry));
  return d;
}
// number of j in 0..NJ-1 with run[j] < t for a nondecreasing run[] (binary search written as selects: 5 compares,
// 11 v_cndmask, no dynamic register index); entries past NJ are +inf and fold away at compile time.  The result is
// put together by the compares' carries (x <- 2x + bit: one v_addc per level) rather than by selects of constants.
template <int NJ>
__device__ inline int count_below(const float (&run)[16], float t) {
  auto R = [&](int j) { return j < NJ ? run[j] : __builtin_inff(); };
  auto pick = [](uint64_t m, float a, float b) { return __builtin_amdgcn_inverse_ballot_w64(m) ? a : b; };
  const uint64_t m3 = __builtin_amdgcn_fcmpf(R(7), t, FCMP_OLT);
  const uint64_t m2 = __builtin_amdgcn_fcmpf(pick(m3, R(11), R(3)), t, FCMP_OLT);
  const float lo = pick(m2, R(5), R(1)), hi = pick(m2, R(13), R(9));
  const uint64_t m1 = __builtin_amdgcn_fcmpf(pick(m3, hi, lo), t, FCMP_OLT);
  const float e0 = pick(m1, R(2), R(0)), e1 = pick(m1, R(6), R(4)), e2 = pick(m1, R(10), R(8)), e3 = pick(m1, R(14), R(12));
  const float f0 = pick(m2, e1, e0), f1 = pick(m2, e3, e2);
  const uint64_t m0 = __builtin_amdgcn_fcmpf(pick(m3, f1, f0), t, FCMP_OLT);
  int x = __builtin_amdgcn_inverse_ballot_w64(m3) ? 1 : 0;
  x = add_with_carry(x, x, m2);
  x = add_with_carry(x, x, m1);
  x = add_with_carry(x, x, m0);
  // (the four probes leave element 15 untested: it is below t only if all sixteen are)
  if constexpr (NJ == 16) x = add_with_carry(x, 0, __builtin_amdgcn_fcmpf(run[15], t, FCMP_OLT));
  return x;
}

// the same for up to 32 running sums: one compare picks the half, 16 selects build it, then the 16-entry search
template <int NJ>
__device__ inline int count_below32(const float (&run)[32], float t) {
  if constexpr (NJ <= 16) {
    float r16[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) r16[j] = run[j];
    return count_below<NJ>(r16, t);
  } else {
    const bool b4 = run[15] < t;
    float sel[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) sel[j] = b4 ? (16 + j < NJ ? run[16 + j] : __builtin_inff()) : run[j];
    return count_below<16>(sel, t) + (b4 ? 16 : 0);
  }
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// layout rule of the scan draw (measured; the oracle restates it): sixteen ants per wavefront (4 lanes per ant) up to
// DACO_SCAN4_MAX_N nodes, eight (8 lanes) up to DACO_SCAN16_MAX_N, two up to DACO_SCAN32_MAX_N (TSP: 1024), one above.
// DACO_SCAN_LAYOUT = 4 | 8 | 16 (measurement knob): that many lanes per ant wherever the kernel supports it
// (4: n <= 128; 8, 16: n <= 256; 16 for TSP: n <= 512), the default rule elsewhere
constexpr int DACO_SCAN4_MAX_N = 128, DACO_SCAN16_MAX_N = 256, DACO_SCAN32_MAX_N = 512;
inline int scan_layout_env() { static const int v = getenv("DACO_SCAN_LAYOUT") ? atoi(getenv("DACO_SCAN_LAYOUT")) : 0; return v; }
inline int scan_small_lanes(int n) {
  const int e = scan_layout_env();
  if (e == 16) return 16;
  if (e == 8 && n <= DACO_SCAN16_MAX_N) return 8;
  if (e == 4 && n <= DACO_SCAN4_MAX_N) return 4;
  return n <= DACO_SCAN4_MAX_N ? 4 : (n <= DACO_SCAN16_MAX_N ? 8 : 16);
}
// TSP: the two-ants-per-wavefront kernel serves n <= 1024 (DACO_SCAN_LAYOUT=64: measurement knob, one ant per wavefront above 512)
inline int tsp_scan32_max_n() { static const int v = (getenv("DACO_SCAN_LAYOUT") && atoi(getenv("DACO_SCAN_LAYOUT")) == 64) ? 512 : 1024; return v; }
// DACO_SCAN_LAYOUT=16 (measurement knob): four ants per wavefront up to n = 512 (TSP)
inline int scan16_max_n() { static const int v = (getenv("DACO_SCAN_LAYOUT") && atoi(getenv("DACO_SCAN_LAYOUT")) == 16) ? 512 : DACO_SCAN16_MAX_N; return v; }
// daco_tsp_scan32.hip: TSP scan draw with two ants per wavefront
hipError_t launch_tsp_scan32(const SampleParams &sp, bool logp, hipStream_t s);
// daco_scan16.hip: sixteen (n <= 128), eight (n <= 256) or four (CVRP: 256 < n <= 512; DACO_SCAN_LAYOUT=16) ants per wavefront
hipError_t launch_tsp_scan16(const SampleParams &sp, bool logp, hipStream_t s);
hipError_t launch_cvrp_scan16(const SampleParams &sp, bool logp, hipStream_t s);

}  // namespace daco