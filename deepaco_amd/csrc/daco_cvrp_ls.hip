// daco_cvrp_ls.hip -- local search for CVRP solutions on the device, one workgroup per (instance, ant).
//
// Reference behaviour replaced: cvrp_nls/aco.py:114-126 (multiple_swap_star: one thread-pool task per ant), :443-448
// (neural_swapstar: search on the distances, a short search on the heuristic-derived matrix as perturbation, search
// on the distances again) and the path it takes into the vendored HGS-CVRP C++ through /tmp files
// (cvrp_nls/swapstar.py:240-271, HGS-CVRP-main/Program/C_Interface.cpp:128-172 -> LocalSearch.cpp).  HGS's LocalSearch is
// first improvement over a granular neighbourhood in a shuffled order (std::minstd_rand + std::shuffle), with load
// penalties instead of hard capacity; it is NOT reproduced move for move.  This kernel is a deterministic
// BEST-improvement search over HGS's move families (LocalSearch.cpp move1 .. move9 and swapStar), hard capacity, specified
// below and restated on the CPU in oracle/cvrp_ls.py (the tests hold the kernel bit-exact against that restatement).
// Parity with the reference is pinned on COST: tests/golden/g8_cvrp_ls_*.npz are routes in / routes out of the
// reference's own swapstar() / neural_swapstar() built from its sources; the schedule of cvrp_nls.ACO has to reach their
// mean cost to within the tolerance stated in tests/test_gpu_09_cvrp_ls.py.
//
// Specification.  A solution is the reference's route sequence 0 a b c 0 d e 0 ... 0 (cvrp/aco.py:138-165) without empty
// routes, L entries, s[0] = s[L-1] = 0.  Per move every candidate (kind, i, j) is evaluated in f32 as
//     change = (((a1 + a2) + a3) + a4) - (((r1 + r2) + r3) + r4)  [+ (float)(reversal term, f64)]
// (a* = lengths of the edges the move adds, r* = of those it removes, in the order listed; missing terms are skipped); the
// smallest change wins, ties to the smallest (kind, i, j), and the move is applied if its change is below -eps,
// eps = max(1e-6, M * 2^-17), M = the largest |entry| of the matrix (a change sums up to a dozen f32 terms of size <= M: its
// rounding error stays below 2e-6 * M, so every applied move lowers the true cost and the search cannot cycle; with an
// absolute 1e-6 it did on matrices with entries in the hundreds once SWAP* was in the move set).  Loads are
// f64 sums of the f32 demands along a route; a route is feasible if its load is <= capacity * (1 + 1e-6) (an exactly full
// route of normalised demands must pass: the reference hands HGS capacity 1000.001 for the same reason, swapstar.py:254).
// With u = s[i], a = s[i-1], c = s[i+1], x = s[i+1], c2 = s[i+2], v = s[j], w = s[j+1], e = s[j-1], g = s[j+1], y = s[j+1],
// g2 = s[j+2]:
//   0 REL1   u between v and w (j != i-1, i)                            add (a,c) (v,u) (u,w)       rem (a,u) (u,c) (v,w)
//   1 REL2   the pair u x between v and w (j not in i-1 .. i+1)          add (a,c2) (v,u) (x,w)      rem (a,u) (x,c2) (v,w)
//   2 REL2R  the pair reversed: ... v x u w ...                          add (a,c2) (v,x) (x,u) (u,w)  rem (a,u) (u,x) (x,c2) (v,w)
//   3 SWAP11 u <-> v, i < j; j = i+1:                                    add (a,v) (v,u) (u,g)       rem (a,u) (u,v) (v,g)
//                            else:                                        add (a,v) (v,c) (e,u) (u,g) rem (a,u) (u,c) (e,v) (v,g)
//   4 SWAP21 the pair u x <-> v (j >= i+3 or j <= i-2)                   add (a,v) (v,c2) (e,u) (x,g)  rem (a,u) (x,c2) (e,v) (v,g)
//   5 SWAP22 the pair u x <-> the pair v y (j >= i+3)                    add (a,v) (y,c2) (e,u) (x,g2) rem (a,u) (x,c2) (e,v) (y,g2)
//   6 2OPT   reverse s[i..j] inside one route, i < j                     add (a,v) (u,g)             rem (a,u) (v,g)     + asym[j] - asym[i]
//   7 TAILS  routes r1 < r2 cut after i and after j, tails exchanged     add (u,y) (v,x)             rem (u,x) (v,y)
//   8 CROSS  r1 = head1 + reversed head2, r2 = reversed tail1 + tail2    add (u,v) (x,y)             rem (u,x) (v,y)     + reversal of head2 and tail1
//   9 SWAP*  (LocalSearch.cpp swapStar) customers u in route r1 and v in route r2, r1 < r2, change routes, EACH AT ITS BEST
//            POSITION of the other route:  change = ((remU + remV) + insU) + insV  with
//              remU = D(a,c) - (D(a,u) + D(u,c)),  remV = D(e,g) - (D(e,v) + D(v,g))            (what the removals save)
//              insU = min( in place: (D(e,u) + D(u,g)) - D(e,g),
//                          best of (D(s[p],u) + D(u,s[p+1])) - D(s[p],s[p+1]) over the positions p of r2, opening depot
//                          included, not next to v (p != j, j-1); ties to the smallest p; the place of v wins a tie )
//              insV likewise in r1 without u.
//            Evaluated ONLY when no move of kinds 0-8 improves (HGS tries SWAP* after its classical moves too): the search
//            stops at a solution that no move of the ten families improves.  Every pair of routes is considered (HGS: only
//            pairs whose polar sectors overlap); O(n^2 * route length) direct evaluation, no insertion tables.
// (asym[k]: f64 sum over the route's edges before position k of d[s[t+1]][s[t]] - d[s[t]][s[t+1]] -- what the edges cost
// more walked backwards; exactly 0 for a symmetric matrix, the perturbation matrix 1/(eta/rowmax + 1e-5) is not symmetric.)
// Moves between routes must keep every route within capacity.  The distance matrix is staged in LDS when it fits
// (n <= 160: every gather of the search is then an LDS read).
// Bookkeeping per move is wave-parallel: route ids from ballots over 64 positions at a time, one lane per route for the
// (sequential, f64) loads and reversal sums, the new sequence gathered by all threads through a table of at most six source
// ranges, empty routes squeezed out with ballots.
#include "daco_device.h"
#include "../../include/deepaco_hip.h"

namespace daco {

constexpr int LS_MAXL = 4112;      // longest sequence accepted (2n+1 entries at most: n <= 2000, the largest size of cvrp_nls/utils.py; positions travel in 14-bit fields)
constexpr int LS_STAGE_MAX_N = 160;

// capacities of the per-launch LDS arrays: positions (Lmax + the appended closing depot, rounded up to a multiple of 8) and
// routes (a route takes at least two positions)
__host__ __device__ inline int ls_cap_l(int Lmax) { return (Lmax + 1 + 7) & ~7; }
__host__ __device__ inline int ls_cap_r(int Lmax) { return ((Lmax + 1) / 2 + 2 + 1) & ~1; }
__host__ __device__ inline size_t ls_fixed_bytes(int Lmax, int n) {
  const size_t Lc = ls_cap_l(Lmax), Rc = ls_cap_r(Lmax);
  return 2 * Lc * 8 + 2 * (Rc + 1) * 8 + 2 * (Lc + 4) * 2 + Lc * 2 + (Rc + 2) * 2 + 32 * 4 + 32 * 4 + (size_t)((n + 3) & ~3) * 4 + 16;
}

__device__ inline bool ls_better(float d1, uint32_t c1, float d2, uint32_t c2) { return d1 < d2 || (d1 == d2 && c1 < c2); }

template <bool STAGE, int NT>
__global__ void __launch_bounds__(NT)
cvrp_ls_kernel(int n, int A, int Lmax, const float *dist, long dist_bs, const float *demand, float capacity, int64_t *paths,
               int count, int32_t *lens_out, int32_t *moves_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // (sized by this launch's Lmax, not by the largest sequence the kernel accepts: at CVRP-100 the workgroup needs 45 KB with
  // the staged matrix instead of 71 KB, i.e. three instead of two of them share a CU)
  const int Lc = ls_cap_l(Lmax), Rc = ls_cap_r(Lmax);
  double *pf = reinterpret_cast<double *>(smem);                    // [Lc] load of the route up to and including k
  double *asym = pf + Lc;                                           // [Lc]
  double *rl = asym + Lc;                                           // [Rc + 1] route loads
  double *asymT = rl + Rc + 1;                                      // [Rc + 1] reversal sum of the whole route
  uint16_t *s = reinterpret_cast<uint16_t *>(asymT + Rc + 1);       // [Lc + 4] sequence (4 spare entries of padding)
  uint16_t *s2 = s + Lc + 4;                                        // [Lc + 4] the sequence being built
  uint16_t *rid = s2 + Lc + 4;                                      // [Lc] route of position k
  uint16_t *rstart = rid + Lc;                                      // [Rc + 2] opening depot of route r
  float *redd = reinterpret_cast<float *>(rstart + Rc + 2);         // [16] reduction   (Lc, Rc even: 4-byte aligned)
  uint32_t *redc = reinterpret_cast<uint32_t *>(redd + 16);         // [16]
  int *shared_i = reinterpret_cast<int *>(redc + 16);                // [0] L, [1] R, [2..] piece table (8 x {lo, hi, rev})
  const int n4 = (n + 3) & ~3;
  float *dem = reinterpret_cast<float *>(shared_i + 32);            // [n4] demands
  float *dl = dem + n4;                                             // [n*n] staged distances (STAGE)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / A, a_ = blockIdx.x - b * A;
  const float *dg = dist + (size_t)b * dist_bs;
  int64_t *col = paths + (size_t)b * Lmax * A + a_;
  const double capT = (double)capacity * (1.0 + 1e-6);

  constexpr int NW = NT / 64;
  for (int k = tid; k < n; k += NT) dem[k] = demand[(size_t)b * n + k];
  float mx = 0.0f;                                          // M: largest |entry| (one pass over the matrix per launch)
  for (int k = tid; k < n * n; k += NT) {
    const float x = dg[k];
    if constexpr (STAGE) dl[k] = x;
    mx = fmaxf(mx, fabsf(x));
  }
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (lane == 0) redd[wave] = mx;
  __syncthreads();
  float mall = redd[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) mall = fmaxf(mall, redd[w]);
  const float neg_eps = -fmaxf(1e-6f, mall * 7.62939453125e-6f);   // 2^-17
  __syncthreads();
  // (24-bit multiply: v_mad_u32_u24 runs at full rate, the 32-bit multiply at a quarter -- a candidate is six to eight look-ups)
  auto D = [&](int u, int v) -> float { return STAGE ? dl[__mul24(u, n) + v] : dg[(uint32_t)(__mul24(u, n) + v)]; };
  // squeeze doubled depots out of src[0..len) into dst (wave 0, 64 entries per step); returns the new length (all lanes)
  auto squeeze = [&](const uint16_t *src, int len, uint16_t *dst, bool from_col) -> int {
    int out = 0, last = -1;                                 // last entry written so far (-1: none)
    for (int k0 = 0; k0 < len; k0 += 64) {
      const int k = k0 + lane;
      const int v = k < len ? (from_col ? (int)col[(size_t)k * A] : (int)src[k]) : -1;
      int before = __shfl_up(v, 1, 64);
      if (lane == 0) before = last;
      const bool keep = k < len && !(v == 0 && before == 0);
      const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
      if (keep) dst[out + __builtin_popcountll(m & ((1ull << lane) - 1))] = (uint16_t)v;
      out += __builtin_popcountll(m);
      const int cnt = len - k0 < 64 ? len - k0 : 64;
      last = __builtin_amdgcn_readlane(v, cnt - 1);
    }
    return out;
  };
  if (wave == 0) {
    int L0 = squeeze(nullptr, Lmax, s, true);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      if (L0 == 0 || s[L0 - 1] != 0) s[L0++] = 0;           // the column had no closing depot
      shared_i[0] = L0;
    }
  }
  __syncthreads();
  int L = shared_i[0];
  int moves = 0;
  for (; moves < count; ++moves) {
    // ---- tables.  Wave 0: route of every position and the routes' opening depots (ballots over 64 positions)
    if (wave == 0) {
      int r = -1;
      for (int k0 = 0; k0 < L; k0 += 64) {
        const int k = k0 + lane;
        const bool dep = k < L && s[k] == 0;
        const uint64_t m = __builtin_amdgcn_ballot_w64(dep);
        const int mine = r + __builtin_popcountll(m & ((2ull << lane) - 1));
        if (k < L) rid[k] = (uint16_t)mine;
        if (dep) rstart[mine] = (uint16_t)k;
        r += __builtin_popcountll(m);
      }
      if (lane == 0) shared_i[1] = r;                       // R: routes (the last depot opened the empty sentinel route R)
      if (lane < 4) s[L + lane] = 0;                        // padding read by the pair moves at the sequence's end
    }
    __syncthreads();
    const int R = shared_i[1];
    // one lane per route: loads and reversal sums along it, sequentially in f64
    for (int r = tid; r <= R; r += NT) {
      const int k0 = rstart[r], k1 = r < R ? rstart[r + 1] : k0;
      double load = 0.0, rev = 0.0;
      pf[k0] = 0.0; asym[k0] = 0.0;
      for (int k = k0 + 1; k < k1; ++k) {
        rev = rev + ((double)D(s[k], s[k - 1]) - (double)D(s[k - 1], s[k]));
        load = load + (double)dem[s[k]];
        pf[k] = load; asym[k] = rev;
      }
      rl[r] = load;
      asymT[r] = r < R ? rev + ((double)D(s[k1], s[k1 - 1]) - (double)D(s[k1 - 1], s[k1])) : 0.0;
    }
    __syncthreads();
    float bd = __builtin_inff();
    uint32_t bc = 0xFFFFFFFFu;
    auto offer = [&](float delta, uint32_t kind, int i, int j) {
      const uint32_t code = (kind << 28) | ((uint32_t)i << 14) | (uint32_t)j;
      if (ls_better(delta, code, bd, bc)) { bd = delta; bc = code; }
    };
    // ---- every (i, j), i, j in [0, L-2]
    const int W = L - 1;
    int i = tid / W, j = tid - i * W;
    for (; i < W; ) {
      const int u = s[i], v = s[j];
      const int ri = rid[i], rj = rid[j];
      if (u != 0) {
        const int pa = s[i - 1], pc = s[i + 1], w = s[j + 1];
        const bool other = ri != rj;
        const float du = dem[u];
        if (j != i && j != i - 1 && (!other || rl[rj] + (double)du <= capT)) {
          const float add = (D(pa, pc) + D(v, u)) + D(u, w);
          const float rem = (D(pa, u) + D(u, pc)) + D(v, w);
          offer(add - rem, 0u, i, j);
        }
        const bool pair = pc != 0;                          // x = s[i+1] is a customer
        const int c2 = s[i + 2];
        if (pair && j != i - 1 && j != i && j != i + 1 && (!other || rl[rj] + (double)du + (double)dem[pc] <= capT)) {
          const float add1 = (D(pa, c2) + D(v, u)) + D(pc, w);
          const float rem1 = (D(pa, u) + D(pc, c2)) + D(v, w);
          offer(add1 - rem1, 1u, i, j);
          const float add2 = ((D(pa, c2) + D(v, pc)) + D(pc, u)) + D(u, w);
          const float rem2 = ((D(pa, u) + D(u, pc)) + D(pc, c2)) + D(v, w);
          offer(add2 - rem2, 2u, i, j);
        }
        if (v != 0) {
          const int pe = s[j - 1], pg = w;
          const float dv = dem[v];
          if (j > i) {
            bool ok = true;
            if (other) ok = rl[ri] - (double)du + (double)dv <= capT && rl[rj] - (double)dv + (double)du <= capT;
            if (ok) {
              if (j == i + 1) {
                const float add = (D(pa, v) + D(v, u)) + D(u, pg);
                const float rem = (D(pa, u) + D(u, v)) + D(v, pg);
                offer(add - rem, 3u, i, j);
              } else {
                const float add = ((D(pa, v) + D(v, pc)) + D(pe, u)) + D(u, pg);
                const float rem = ((D(pa, u) + D(u, pc)) + D(pe, v)) + D(v, pg);
                offer(add - rem, 3u, i, j);
              }
            }
            if (!other) {
              const float add = D(pa, v) + D(u, pg);
              const float rem = D(pa, u) + D(v, pg);
              offer((add - rem) + (float)(asym[j] - asym[i]), 6u, i, j);
            }
          }
          if (pair && (j >= i + 3 || j <= i - 2)) {
            const double dp = (double)du + (double)dem[pc];
            bool ok = true;
            if (other) ok = rl[ri] - dp + (double)dv <= capT && rl[rj] - (double)dv + dp <= capT;
            if (ok) {
              const float add = ((D(pa, v) + D(v, c2)) + D(pe, u)) + D(pc, pg);
              const float rem = ((D(pa, u) + D(pc, c2)) + D(pe, v)) + D(v, pg);
              offer(add - rem, 4u, i, j);
            }
            if (j >= i + 3 && pg != 0) {                    // y = s[j+1] is a customer too
              const int g2 = s[j + 2];
              const double dq = (double)dv + (double)dem[pg];
              bool ok2 = true;
              if (other) ok2 = rl[ri] - dp + dq <= capT && rl[rj] - dq + dp <= capT;
              if (ok2) {
                const float add = ((D(pa, v) + D(pg, c2)) + D(pe, u)) + D(pc, g2);
                const float rem = ((D(pa, u) + D(pc, c2)) + D(pe, v)) + D(pg, g2);
                offer(add - rem, 5u, i, j);
              }
            }
          }
        }
      }
      if (ri < rj && rj < R) {
        const int x = s[i + 1], y = s[j + 1];
        const double hi_ = pf[i], hj = pf[j], ti = rl[ri] - hi_, tj = rl[rj] - hj;
        if (hi_ + tj <= capT && hj + ti <= capT) {
          const float add = D(u, y) + D(v, x);
          const float rem = D(u, x) + D(v, y);
          offer(add - rem, 7u, i, j);
        }
        if (hi_ + hj <= capT && ti + tj <= capT) {
          const float add = D(u, v) + D(x, y);
          const float rem = D(u, x) + D(v, y);
          const double rev = asym[j] + (x != 0 ? asymT[ri] - asym[i + 1] : 0.0);
          offer((add - rem) + (float)rev, 8u, i, j);
        }
      }
      j += NT;
      while (j >= W) { j -= W; ++i; }
    }
    // ---- workgroup minimum
    auto block_min = [&]() {
      for (int o = 32; o >= 1; o >>= 1) {
        const float od = __shfl_xor(bd, o);
        const uint32_t oc = (uint32_t)__shfl_xor((int)bc, o);
        if (ls_better(od, oc, bd, bc)) { bd = od; bc = oc; }
      }
      if (lane == 0) { redd[wave] = bd; redc[wave] = bc; }
      __syncthreads();
      bd = redd[0]; bc = redc[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) if (ls_better(redd[w], redc[w], bd, bc)) { bd = redd[w]; bc = redc[w]; }
    };
    block_min();
    // cheapest insertion of `node` into route r without the customer at position skip (kind 9): cost, and the position it
    // goes behind (skip itself: the place of the removed customer)
    auto ins_best = [&](int node, int r, int skip, int &where) -> float {
      const int k0 = rstart[r], k1 = rstart[r + 1];
      float best = __builtin_inff();
      int bp = skip;
      int sp = s[k0];
      for (int p = k0; p < k1; ++p) {
        const int sn = s[p + 1];
        if (p != skip && p != skip - 1) {
          const float cst = (D(sp, node) + D(node, sn)) - D(sp, sn);
          if (cst < best) { best = cst; bp = p; }
        }
        sp = sn;
      }
      const int e = s[skip - 1], g = s[skip + 1];
      const float cin = (D(e, node) + D(node, g)) - D(e, g);
      where = cin <= best ? skip : bp;
      return cin <= best ? cin : best;
    };
    if (!(bd < neg_eps) || bc == 0xFFFFFFFFu) {
      // ---- no classical move improves: SWAP* over every pair of customers in different routes
      __syncthreads();                                      // (everyone has read the reduction slots)
      bd = __builtin_inff(); bc = 0xFFFFFFFFu;
      int i2 = tid / W, j2 = tid - i2 * W;
      for (; i2 < W; ) {
        const int u = s[i2], v = s[j2];
        if (u != 0 && v != 0 && rid[i2] < rid[j2]) {
          const int ri = rid[i2], rj = rid[j2];
          const double du = (double)dem[u], dv = (double)dem[v];
          if (rl[ri] - du + dv <= capT && rl[rj] - dv + du <= capT) {
            const int pa = s[i2 - 1], pc = s[i2 + 1], pe = s[j2 - 1], pg = s[j2 + 1];
            const float remU = D(pa, pc) - (D(pa, u) + D(u, pc));
            const float remV = D(pe, pg) - (D(pe, v) + D(v, pg));
            int w_;
            const float insU = ins_best(u, rj, j2, w_);
            const float insV = ins_best(v, ri, i2, w_);
            offer(((remU + remV) + insU) + insV, 9u, i2, j2);
          }
        }
        j2 += NT;
        while (j2 >= W) { j2 -= W; ++i2; }
      }
      block_min();
      if (!(bd < neg_eps) || bc == 0xFFFFFFFFu) break;
    }
    // ---- apply: the new sequence is at most eight ranges of the old one (some reversed)
    if (tid == 0) {
      const int kind = (int)(bc >> 28), mi = (int)((bc >> 14) & 0x3FFF), mj = (int)(bc & 0x3FFF);
      int *pt = shared_i + 2;
      int np = 0;
      auto piece = [&](int lo, int hi, int rev) { if (lo <= hi) { pt[3 * np] = lo; pt[3 * np + 1] = hi; pt[3 * np + 2] = rev; ++np; } };
      if (kind <= 2) {
        const int len = kind == 0 ? 1 : 2, rv = kind == 2;
        if (mj > mi) { piece(0, mi - 1, 0); piece(mi + len, mj, 0); piece(mi, mi + len - 1, rv); piece(mj + 1, L - 1, 0); }
        else { piece(0, mj, 0); piece(mi, mi + len - 1, rv); piece(mj + 1, mi - 1, 0); piece(mi + len, L - 1, 0); }
      } else if (kind == 3) {
        piece(0, mi - 1, 0); piece(mj, mj, 0); piece(mi + 1, mj - 1, 0); piece(mi, mi, 0); piece(mj + 1, L - 1, 0);
      } else if (kind == 4) {
        if (mj > mi) { piece(0, mi - 1, 0); piece(mj, mj, 0); piece(mi + 2, mj - 1, 0); piece(mi, mi + 1, 0); piece(mj + 1, L - 1, 0); }
        else { piece(0, mj - 1, 0); piece(mi, mi + 1, 0); piece(mj + 1, mi - 1, 0); piece(mj, mj, 0); piece(mi + 2, L - 1, 0); }
      } else if (kind == 5) {
        piece(0, mi - 1, 0); piece(mj, mj + 1, 0); piece(mi + 2, mj - 1, 0); piece(mi, mi + 1, 0); piece(mj + 2, L - 1, 0);
      } else if (kind == 6) {
        piece(0, mi - 1, 0); piece(mi, mj, 1); piece(mj + 1, L - 1, 0);
      } else if (kind == 9) {
        int pu, pv;                                         // u goes behind pu in r2 (pu == mj: where v was), v behind pv in r1
        ins_best(s[mi], rid[mj], mj, pu);
        ins_best(s[mj], rid[mi], mi, pv);
        int next;
        if (pv == mi) { piece(0, mi - 1, 0); piece(mj, mj, 0); next = mi + 1; }
        else if (pv < mi) { piece(0, pv, 0); piece(mj, mj, 0); piece(pv + 1, mi - 1, 0); next = mi + 1; }
        else { piece(0, mi - 1, 0); piece(mi + 1, pv, 0); piece(mj, mj, 0); next = pv + 1; }
        if (pu == mj) { piece(next, mj - 1, 0); piece(mi, mi, 0); piece(mj + 1, L - 1, 0); }
        else if (pu < mj) { piece(next, pu, 0); piece(mi, mi, 0); piece(pu + 1, mj - 1, 0); piece(mj + 1, L - 1, 0); }
        else { piece(next, mj - 1, 0); piece(mj + 1, pu, 0); piece(mi, mi, 0); piece(pu + 1, L - 1, 0); }
      } else {
        const int e1 = rstart[rid[mi] + 1], b2 = rstart[rid[mj]], e2 = rstart[rid[mj] + 1];
        if (kind == 7) { piece(0, mi, 0); piece(mj + 1, e2 - 1, 0); piece(e1, mj, 0); piece(mi + 1, e1 - 1, 0); piece(e2, L - 1, 0); }
        else { piece(0, mi, 0); piece(b2 + 1, mj, 1); piece(e1, b2, 0); piece(mi + 1, e1 - 1, 1); piece(mj + 1, e2 - 1, 0); piece(e2, L - 1, 0); }
      }
      for (int q = np; q < 8; ++q) { pt[3 * q] = 1; pt[3 * q + 1] = 0; pt[3 * q + 2] = 0; }      // empty
    }
    __syncthreads();
    for (int k = tid; k < L; k += NT) {
      const int *pt = shared_i + 2;
      int rest = k, src = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int lo = pt[3 * q], hi = pt[3 * q + 1], len = hi - lo + 1;
        if (len > 0) {
          if (rest >= 0 && rest < len) src = pt[3 * q + 2] ? hi - rest : lo + rest;
          rest -= len;
        }
      }
      s2[k] = s[src];
    }
    __syncthreads();
    if (wave == 0) {
      const int Ln = squeeze(s2, L, s, false);               // a route may have become empty: drop the doubled depot
      if (lane == 0) shared_i[0] = Ln;
    }
    __syncthreads();
    L = shared_i[0];
  }
  __syncthreads();
  for (int t = tid; t < Lmax; t += NT) col[(size_t)t * A] = t < L ? (int64_t)s[t] : 0;
  if (tid == 0) {
    if (lens_out) lens_out[blockIdx.x] = L;
    if (moves_out) moves_out[blockIdx.x] = moves;
  }
}

}  // namespace daco

using namespace daco;

extern "C" int daco_cvrp_local_search(void *stream, int B, int n, int A, int Lmax, const float *dist, long dist_bstride,
                                      const float *demand, float capacity, int64_t *paths, int max_moves, int32_t *lens,
                                      int32_t *moves) {
  if (B <= 0 || n < 2 || A <= 0 || Lmax < 3 || !dist || !demand || !paths || max_moves < 0) {
    set_error("daco_cvrp_local_search: bad argument (B=%d n=%d A=%d Lmax=%d)", B, n, A, Lmax);
    return DACO_E_BADARG;
  }
  // (a column without a closing depot gets one appended: one entry of head room)
  if (Lmax >= LS_MAXL) { set_error("daco_cvrp_local_search: Lmax=%d must stay below %d", Lmax, LS_MAXL); return DACO_E_TOOLARGE; }
  if (n > 16383) { set_error("daco_cvrp_local_search: n=%d is above 16383 (positions travel in 14-bit fields)", n); return DACO_E_TOOLARGE; }
  const bool stage = n <= LS_STAGE_MAX_N;
  const size_t lds = ls_fixed_bytes(Lmax, n) + (stage ? (size_t)n * n * 4 : 0);
  hipStream_t s = (hipStream_t)stream;
  // threads per solution: the pair loop is a chain of LDS look-ups -- with the matrix staged only three workgroups fit a CU, and
  // eight wavefronts each keep it busier than four (DACO_CVRP_LS_THREADS: 256 | 512)
  static const int nt_env = getenv("DACO_CVRP_LS_THREADS") ? atoi(getenv("DACO_CVRP_LS_THREADS")) : 0;
  // (measured at CVRP-100, 64 x 512 solutions: 256 / 512 / 1024 threads -> 67.8 / 93.2 / 57.5 k solutions/s; without the staged
  // matrix a workgroup's LDS is small and eight of 256 threads fill the CU: unchanged)
  const int nt = nt_env == 256 || nt_env == 512 || nt_env == 1024 ? nt_env : (stage ? 512 : 256);
#define DACO_LS_LAUNCH(STAGE_, NT_)                                                                                              \
  do {                                                                                                                             \
    if (lds > 64 * 1024) {                                                                                                         \
      hipError_t e_ = hipFuncSetAttribute((const void *)cvrp_ls_kernel<STAGE_, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e_ != hipSuccess) { set_error("daco_cvrp_local_search: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e_)); return DACO_E_HIP; } \
    }                                                                                                                              \
    hipLaunchKernelGGL((cvrp_ls_kernel<STAGE_, NT_>), dim3(B * A), dim3(NT_), lds, s, n, A, Lmax, dist, dist_bstride, demand, capacity, \
                       paths, max_moves, lens, moves);                                                                             \
  } while (0)
  if (stage) { if (nt == 1024) DACO_LS_LAUNCH(true, 1024); else if (nt == 512) DACO_LS_LAUNCH(true, 512); else DACO_LS_LAUNCH(true, 256); }
  else { if (nt == 1024) DACO_LS_LAUNCH(false, 1024); else if (nt == 512) DACO_LS_LAUNCH(false, 512); else DACO_LS_LAUNCH(false, 256); }
#undef DACO_LS_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("cvrp_ls_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}
