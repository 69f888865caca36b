// daco_tsp_sample.hip -- tour construction (ACO.gen_path / pick_move) for gfx950.
//
// Reference behaviour replaced: tsp/aco.py:134-177, tsp_nls/aco.py:184-220 (torch path) and,
// for the prefix-scan mode, the roulette sampler tsp_nls/aco.py:260-275.
//
// Design: one wavefront per (instance, ant) for the whole tour -- the
// n-1 dependent draws never leave the wave.  Each step streams ONE padded row of the fused
// transition matrix P = tau^alpha * eta^beta (built once per call by prob_matrix_kernel, so
// the per-step traffic is 4*ld bytes instead of the reference's 8n) with 16-byte loads, lane
// l owning candidates (c*64+l)*VEC+v.  The visited set is a per-lane 64-bit register bitset;
// the draw is a DPP reduction / prefix scan; nothing is staged through LDS because no lane
// consumes another lane's candidates.  Workgroups are remapped so each XCD walks whole
// instances (rows stay in its private L2).
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
  uint32_t ant_gid0;
  int64_t *paths;        // [B][n][A]
  float *logp;           // [B][n-1][A] or null
  float *rowsum;         // [B][n-1][A] or null
  int32_t *flags;        // [B] or null
  const float *dist;     // [B][n][n] (fused costs) or null
  long dist_bs;
  float *costs;          // [B][A] or null
  uint32_t *nbr;         // [B][A][n] prev | next << 16 (for the pheromone update) or null
  // CVRP (cvrp/aco.py:138-205): node 0 = depot, variable-length routes
  const float *mask;     // PROB_STEP: [B][A][n] f32, 0 = closed
  int step;              // PROB_STEP: step index (RNG counter word)
  const float *demand;   // [B][n]
  float capacity;
  int Lmax;              // rows of paths (and Lmax-1 rows of logp)
  int noise_steps;       // rows of the noise tensor
  int32_t *lens;         // [B][A] rows used by each ant
};

template <class F, int... I>
__device__ inline void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
// compile-time loop: f receives std::integral_constant<int, j>, j = 0..N-1
template <int N, class F>
__device__ inline void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int VEC> struct Vec;
template <> struct Vec<1> { float v[1]; };
template <> struct Vec<2> { float v[2]; };
template <> struct Vec<4> { float v[4]; };

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

// P = tau^alpha * eta^beta with zero padding; R = 1/P with +inf padding (optional)
__global__ void __launch_bounds__(256)
prob_matrix_kernel(int B, int n, int ld, const float *tau, long tau_bs, const float *eta, long eta_bs,
                   float alpha, float beta, float *P, float *R) {
  const long total = (long)B * n * ld;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % ld);
    const long row = i / ld;            // b*n + r
    const int b = (int)(row / n), r = (int)(row % n);
    float p = 0.0f;
    if (k < n) {
      const float t = tau[b * tau_bs + (long)r * n + k];
      const float e = eta[b * eta_bs + (long)r * n + k];
      p = pw(t, alpha) * pw(e, beta);
    }
    P[i] = p;
    if (R) R[i] = k < n ? 1.0f / p : __builtin_inff();
  }
}

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
  // 0xFFFFFFFF if the candidate is closed, else 0 (v_bfe_i32); `x & ~mask` zeroes closed ones
  template <int BIT> __device__ inline uint32_t closed_mask() const {
    return (uint32_t)__builtin_amdgcn_sbfe((int)(BIT < 32 ? lo : hi), BIT & 31, 1);
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

enum { PROB_TSP = 0, PROB_CVRP = 1, PROB_STEP = 2 };

// PROB_TSP: whole closed tour; PROB_CVRP: whole capacity-constrained route sequence;
// PROB_STEP: ONE draw per ant from an externally maintained mask (ACO.pick_move for the sibling
// problems, whose feasibility logic stays with the caller).
template <int VEC, int CH, int MODE, bool LOGP, int PROB>
__global__ void __launch_bounds__(256)
tsp_sample_kernel(const SampleParams p) {
  constexpr bool CVRP = PROB == PROB_CVRP, STEP = PROB == PROB_STEP;
  constexpr int NJ = CH * VEC;                          // candidates per lane
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int w = xcd_remap(blockIdx.x, gridDim.x);
  const int bpi = (p.A + 3) >> 2;                       // workgroups per instance (4 ants each)
  const int b = w / bpi;
  const int a = (w - b * bpi) * 4 + wave;
  if (a >= p.A) return;                                 // no barriers below: safe
  const int n = p.n, A = p.A, ld = p.ld;
  const uint32_t gid = p.ant_gid0 + (uint32_t)(b * A + a);
  const float *Pb = p.P + (size_t)b * n * ld + lane * VEC;
  const float *Rb = (MODE == DACO_RACE_PHILOX) ? p.R + (size_t)b * n * ld + lane * VEC : nullptr;
  const int rows = CVRP ? p.Lmax : (STEP ? 2 : n);       // rows of paths for one instance (STEP: logp has 1 row)
  int64_t *path_out = p.paths + (STEP ? (size_t)b * A : (size_t)b * rows * A) + a;
  float *logp_out = LOGP ? p.logp + (size_t)b * (rows - 1) * A + a : nullptr;
  float *rs_out = (LOGP && p.rowsum) ? p.rowsum + (size_t)b * (rows - 1) * A + a : nullptr;
  const float *dist_b = (PROB == PROB_TSP && p.costs) ? p.dist + (size_t)b * p.dist_bs : nullptr;
  uint32_t *nbr_a = (PROB == PROB_TSP && p.nbr) ? p.nbr + ((size_t)b * A + a) * n : nullptr;
  int pprev = 0, second = 0;                            // neighbour-table bookkeeping
  const float *demand_b = CVRP ? p.demand + (size_t)b * n : nullptr;

  // ---- start node
  int prev;
  if constexpr (CVRP) prev = 0;
  else if (p.start) prev = (int)p.start[(size_t)b * A + a];
  else if (p.fixed_start >= 0) prev = p.fixed_start;
  else {
    const u32x4 r = rng_block(p.seed, p.iter, STREAM_START, gid, 0);
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
  if constexpr (PROB == PROB_TSP) mark(prev);
  if (lane == 0 && !STEP) path_out[0] = prev;

  // CVRP state: this lane's candidates' demands, customers left, load on the current route
  float dem[CH][VEC];
  int remaining = n - 1;
  float used = 0.0f;
  if constexpr (CVRP) {
    static_for<NJ>([&](auto J) {
      constexpr int j = J, c = j / VEC, v = j % VEC;
      const int k = (c * 64 + lane) * VEC + v;
      dem[c][v] = k < n ? demand_b[k] : __builtin_inff();
    });
    used = used + demand_b[0];
  }

  u32x4 ublk = {0, 0, 0, 0};                            // SCAN: 256 cached uniforms per wave
  uint32_t ucur = 0;
  bool infeasible = false, overflow = false;
  float cost = 0.0f, dpend = 0.0f;                      // fused tour length (edge added one step late)

  const int t0 = STEP ? p.step : 1;                     // STEP: the caller's step index keys the RNG
  int t = t0;
  for (; CVRP ? (t < p.Lmax && !(remaining == 0 && prev == 0)) : (STEP ? t == t0 : t < n); ++t) {
    // ---- candidates closed at this step: visited, plus (CVRP) over capacity / depot rule
    Visited blk = vis;
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
      const float rem = p.capacity - used;
      static_for<NJ>([&](auto J) {
        constexpr int j = J;
        blk.template set_if<j>(dem[j / VEC][j % VEC] > rem);          // strict, cvrp/aco.py:200
      });
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
        if ((t & 255) == 0 || t == t0) ublk = rng_block(p.seed, p.iter, STREAM_SCAN, gid, (uint32_t)(((t >> 8) << 6) + lane));
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
      float r = u01(ux) * S;
      r = r > 0.0f ? r : 1.401298464e-45f;               // keep r > 0 if u*S underflows
      const uint64_t m = __ballot(incl >= r && part > 0.0f);
      if (m == 0) { infeasible = true; choice = 0; }
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
        if (v == 0) r4 = rng_block(p.seed, p.iter, STREAM_RACE, gid, ((uint32_t)t << 12) | (uint32_t)(k >> 2));
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
      const float *q = p.noise + (((size_t)b * (CVRP ? p.noise_steps : (STEP ? 1 : n - 1)) + (t - t0)) * A + a) * n;
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
      else used = 0.0f;
      used = used + demand_b[choice];                  // scalar load
    } else if constexpr (PROB == PROB_TSP) {
      if (MODE == DACO_SCAN && own_lane >= 0) { if (lane == own_lane) vis.set(own_bit); }
      else mark(choice);
    }
    if (lane == 0) path_out[STEP ? 0 : (size_t)t * A] = choice;
    if (dist_b) {                                        // fused gen_path_costs (wave-uniform)
      cost = cost + dpend;
      dpend = dist_b[(unsigned)choice * (unsigned)n + (unsigned)prev];   // d[u_t][u_{t-1}], scalar load
    }
    if (nbr_a) {                                         // node `prev` now knows both neighbours
      if (lane == 0) nbr_a[(unsigned)prev] = (uint32_t)pprev | ((uint32_t)choice << 16);
      if (t == 1) second = choice;
      pprev = prev;
    }
    prev = choice;
  }
  if constexpr (CVRP) {
    // the reference steps every ant until the slowest one is done: a done ant keeps drawing
    // the depot (probability 1), so its column is padded with 0 / log(1-eps)
    if (!(remaining == 0 && prev == 0)) overflow = true;
    if (lane == 0) {
      if (p.lens) p.lens[(size_t)b * A + a] = t;
      const float lp1 = clamp_log(1.0f);
      for (int tt = t; tt < p.Lmax; ++tt) {
        path_out[(size_t)tt * A] = 0;
        if constexpr (LOGP) logp_out[(size_t)(tt - 1) * A] = lp1;
      }
    }
    if (overflow && p.flags && lane == 0) atomicOr(p.flags + b, 2);
  }
  if (dist_b) {
    cost = cost + dpend;
    cost = cost + dist_b[(unsigned)first * (unsigned)n + (unsigned)prev];   // closing edge d[u_0][u_{n-1}] last
    if (lane == 0) p.costs[(size_t)b * A + a] = cost;
  }
  if (nbr_a && lane == 0) {                             // close the cycle: last -> first -> second
    if (n == 2) { nbr_a[first] = (uint32_t)prev | ((uint32_t)prev << 16); nbr_a[prev] = (uint32_t)first | ((uint32_t)first << 16); }
    else { nbr_a[prev] = (uint32_t)pprev | ((uint32_t)first << 16); nbr_a[first] = (uint32_t)prev | ((uint32_t)second << 16); }
  }
  if (infeasible && p.flags && lane == 0) atomicOr(p.flags + b, 1);
}

// ------------------------------------------------------------------ host dispatch
template <int VEC, int CH, int CVRP>
static hipError_t launch_sample(const SampleParams &sp, int mode, bool logp, hipStream_t s) {
  const int bpi = (sp.A + 3) / 4;
  dim3 grid((unsigned)(sp.B * bpi)), block(256);
#define DACO_LAUNCH(M, L) hipLaunchKernelGGL((tsp_sample_kernel<VEC, CH, M, L, CVRP>), grid, block, 0, s, sp)
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

}  // namespace daco

using namespace daco;

extern "C" int daco_vec_for_n(int n) { return vec_for_n(n); }
extern "C" int daco_ld_for_n(int n) { return ld_for_n(n); }

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// chunks per lane actually instantiated (compile-time loop bounds): the row is padded with
// zeros up to the next instantiated size; zero padding never changes a sum or a draw.
static int inst_chunks(int n) {
  const int vec = vec_for_n(n), need = ld_for_n(n) / (64 * vec);
  static const int avail[] = {1, 2, 3, 4, 6, 8, 12, 16};
  for (int c : avail) if (c >= need) return c;
  return -1;
}
static int ld_alloc(int n) { return inst_chunks(n) * 64 * vec_for_n(n); }

extern "C" size_t daco_tsp_sample_workspace_bytes(int B, int n, int mode) {
  if (B <= 0 || n <= 0 || n > DACO_MAX_NODES) return 0;
  const size_t mat = align256((size_t)B * n * ld_alloc(n) * sizeof(float));
  return mode == DACO_RACE_PHILOX ? 2 * mat : mat;
}

extern "C" int daco_tsp_sample(void *stream, int B, int n, int A, const float *tau, long tau_bstride,
                               const float *eta, long eta_bstride, float alpha, float beta, int mode,
                               int norm_passes, const int64_t *start, int fixed_start,
                               const float *noise, uint64_t seed, uint64_t iter, uint32_t ant_gid0,
                               int64_t *paths, float *logp, float *rowsum, int32_t *flags,
                               const float *dist, long dist_bstride, float *costs, uint32_t *nbr,
                               void *workspace, size_t workspace_bytes, void *ev_begin, void *ev_end) {
  if (B <= 0 || n < 2 || A <= 0 || !tau || !eta || !paths || !workspace) {
    set_error("daco_tsp_sample: bad argument (B=%d n=%d A=%d tau=%p eta=%p paths=%p ws=%p)", B, n, A,
              (const void *)tau, (const void *)eta, (void *)paths, workspace);
    return DACO_E_BADARG;
  }
  if (n > DACO_MAX_NODES) { set_error("daco_tsp_sample: n=%d exceeds DACO_MAX_NODES=%d", n, DACO_MAX_NODES); return DACO_E_TOOLARGE; }
  if (mode < 0 || mode > 2 || norm_passes < 0 || norm_passes > 2) { set_error("daco_tsp_sample: bad mode %d / norm_passes %d", mode, norm_passes); return DACO_E_BADARG; }
  if (mode == DACO_RACE_NOISE && !noise) { set_error("daco_tsp_sample: DACO_RACE_NOISE needs a noise tensor"); return DACO_E_BADARG; }
  if (fixed_start >= n) { set_error("daco_tsp_sample: fixed_start %d >= n %d", fixed_start, n); return DACO_E_BADARG; }
  if (costs && !dist) { set_error("daco_tsp_sample: fused costs need the distance matrix"); return DACO_E_BADARG; }
  const size_t need = daco_tsp_sample_workspace_bytes(B, n, mode);
  if (workspace_bytes < need) { set_error("daco_tsp_sample: workspace %zu < %zu bytes", workspace_bytes, need); return DACO_E_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  const int vec = vec_for_n(n), CH = inst_chunks(n), ld = ld_alloc(n);
  float *P = (float *)workspace;
  float *R = mode == DACO_RACE_PHILOX ? (float *)((char *)workspace + need / 2) : nullptr;
  {
    const long total = (long)B * n * ld;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(prob_matrix_kernel, dim3(blocks), dim3(256), 0, s, B, n, ld, tau, tau_bstride, eta,
                       eta_bstride, alpha, beta, P, R);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("prob_matrix_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  }
  SampleParams sp;
  sp.B = B; sp.n = n; sp.A = A; sp.ld = ld; sp.CH = CH;
  sp.P = P; sp.R = R; sp.norm_passes = norm_passes; sp.start = start; sp.fixed_start = fixed_start;
  sp.noise = noise; sp.seed = seed; sp.iter = iter; sp.ant_gid0 = ant_gid0;
  sp.paths = paths; sp.logp = logp; sp.rowsum = rowsum; sp.flags = flags;
  sp.dist = dist; sp.dist_bs = dist_bstride; sp.costs = costs; sp.nbr = nbr;
  sp.demand = nullptr; sp.capacity = 0.0f; sp.Lmax = 0; sp.noise_steps = 0; sp.lens = nullptr;
  sp.mask = nullptr; sp.step = 0;
  const bool lp = logp != nullptr;
  hipError_t e;
  if (ev_begin && hipEventRecord((hipEvent_t)ev_begin, s) != hipSuccess) { set_error("hipEventRecord(ev_begin) failed"); return DACO_E_HIP; }
  e = dispatch_sample<PROB_TSP>(sp, vec, CH, mode, lp, s);
  if (e != hipSuccess) { set_error("tsp_sample_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  if (ev_end && hipEventRecord((hipEvent_t)ev_end, s) != hipSuccess) { set_error("hipEventRecord(ev_end) failed"); return DACO_E_HIP; }
  return DACO_OK;
}

extern "C" int daco_cvrp_sample(void *stream, int B, int n, int A, const float *tau, long tau_bstride,
                                const float *eta, long eta_bstride, float alpha, float beta,
                                const float *demand, float capacity, int mode, const float *noise,
                                int noise_steps, uint64_t seed, uint64_t iter, uint32_t ant_gid0, int Lmax,
                                int64_t *paths, float *logp, float *rowsum, int32_t *lens, int32_t *flags,
                                void *workspace, size_t workspace_bytes) {
  if (B <= 0 || n < 2 || A <= 0 || !tau || !eta || !demand || !paths || !workspace || Lmax < 2) {
    set_error("daco_cvrp_sample: bad argument (B=%d n=%d A=%d Lmax=%d)", B, n, A, Lmax);
    return DACO_E_BADARG;
  }
  if (n > DACO_MAX_NODES) { set_error("daco_cvrp_sample: n=%d exceeds DACO_MAX_NODES=%d", n, DACO_MAX_NODES); return DACO_E_TOOLARGE; }
  if (mode < 0 || mode > 2) { set_error("daco_cvrp_sample: bad mode %d", mode); return DACO_E_BADARG; }
  if (mode == DACO_RACE_NOISE && (!noise || noise_steps <= 0)) { set_error("daco_cvrp_sample: DACO_RACE_NOISE needs a noise tensor"); return DACO_E_BADARG; }
  const size_t need = daco_tsp_sample_workspace_bytes(B, n, mode);
  if (workspace_bytes < need) { set_error("daco_cvrp_sample: workspace %zu < %zu bytes", workspace_bytes, need); return DACO_E_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  const int vec = vec_for_n(n), CH = inst_chunks(n), ld = ld_alloc(n);
  float *P = (float *)workspace;
  float *R = mode == DACO_RACE_PHILOX ? (float *)((char *)workspace + need / 2) : nullptr;
  {
    const long total = (long)B * n * ld;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(prob_matrix_kernel, dim3(blocks), dim3(256), 0, s, B, n, ld, tau, tau_bstride, eta,
                       eta_bstride, alpha, beta, P, R);
  }
  SampleParams sp;
  sp.B = B; sp.n = n; sp.A = A; sp.ld = ld; sp.CH = CH;
  sp.P = P; sp.R = R; sp.norm_passes = 1; sp.start = nullptr; sp.fixed_start = 0;
  sp.noise = noise; sp.seed = seed; sp.iter = iter; sp.ant_gid0 = ant_gid0;
  sp.paths = paths; sp.logp = logp; sp.rowsum = rowsum; sp.flags = flags;
  sp.dist = nullptr; sp.dist_bs = 0; sp.costs = nullptr; sp.nbr = nullptr;
  sp.demand = demand; sp.capacity = capacity; sp.Lmax = Lmax; sp.noise_steps = noise_steps; sp.lens = lens;
  sp.mask = nullptr; sp.step = 0;
  hipError_t e = dispatch_sample<PROB_CVRP>(sp, vec, CH, mode, logp != nullptr, s);
  if (e != hipSuccess) { set_error("cvrp sample kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}

// ------------------------------------------------------------------ step-wise API (sibling problems)
extern "C" int daco_prob_matrix(void *stream, int B, int n, const float *tau, long tau_bstride, const float *eta,
                                long eta_bstride, float alpha, float beta, int mode, void *workspace,
                                size_t workspace_bytes) {
  if (B <= 0 || n < 2 || !tau || !eta || !workspace) { set_error("daco_prob_matrix: bad argument"); return DACO_E_BADARG; }
  if (n > DACO_MAX_NODES) { set_error("daco_prob_matrix: n=%d exceeds DACO_MAX_NODES", n); return DACO_E_TOOLARGE; }
  const size_t need = daco_tsp_sample_workspace_bytes(B, n, mode);
  if (workspace_bytes < need) { set_error("daco_prob_matrix: workspace %zu < %zu bytes", workspace_bytes, need); return DACO_E_WORKSPACE; }
  const int ld = ld_alloc(n);
  float *P = (float *)workspace;
  float *R = mode == DACO_RACE_PHILOX ? (float *)((char *)workspace + need / 2) : nullptr;
  const long total = (long)B * n * ld;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(prob_matrix_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, B, n, ld, tau, tau_bstride, eta,
                     eta_bstride, alpha, beta, P, R);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("prob_matrix_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}

extern "C" int daco_pick_move(void *stream, int B, int n, int A, const void *prob_workspace, size_t workspace_bytes,
                              int mode, const int64_t *prev, const float *mask, const float *noise, uint64_t seed,
                              uint64_t iter, uint32_t ant_gid0, int step, int64_t *actions, float *logp,
                              float *rowsum, int32_t *flags) {
  if (B <= 0 || n < 2 || A <= 0 || !prob_workspace || !prev || !mask || !actions || step < 0) {
    set_error("daco_pick_move: bad argument (B=%d n=%d A=%d step=%d)", B, n, A, step);
    return DACO_E_BADARG;
  }
  if (n > DACO_MAX_NODES) { set_error("daco_pick_move: n=%d exceeds DACO_MAX_NODES", n); return DACO_E_TOOLARGE; }
  if (mode < 0 || mode > 2) { set_error("daco_pick_move: bad mode %d", mode); return DACO_E_BADARG; }
  if (mode == DACO_RACE_NOISE && !noise) { set_error("daco_pick_move: DACO_RACE_NOISE needs a noise tensor"); return DACO_E_BADARG; }
  const size_t need = daco_tsp_sample_workspace_bytes(B, n, mode);
  if (workspace_bytes < need) { set_error("daco_pick_move: workspace %zu < %zu bytes", workspace_bytes, need); return DACO_E_WORKSPACE; }
  const int vec = vec_for_n(n), CH = inst_chunks(n), ld = ld_alloc(n);
  SampleParams sp;
  sp.B = B; sp.n = n; sp.A = A; sp.ld = ld; sp.CH = CH;
  sp.P = (const float *)prob_workspace;
  sp.R = mode == DACO_RACE_PHILOX ? (const float *)((const char *)prob_workspace + need / 2) : nullptr;
  sp.norm_passes = 1; sp.start = prev; sp.fixed_start = -1; sp.noise = noise; sp.seed = seed; sp.iter = iter;
  sp.ant_gid0 = ant_gid0; sp.paths = actions; sp.logp = logp; sp.rowsum = rowsum; sp.flags = flags;
  sp.dist = nullptr; sp.dist_bs = 0; sp.costs = nullptr; sp.nbr = nullptr;
  sp.demand = nullptr; sp.capacity = 0.0f; sp.Lmax = 0; sp.noise_steps = 1; sp.lens = nullptr;
  sp.mask = mask; sp.step = step;
  hipError_t e = dispatch_sample<PROB_STEP>(sp, vec, CH, mode, logp != nullptr, (hipStream_t)stream);
  if (e != hipSuccess) { set_error("pick_move kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}
