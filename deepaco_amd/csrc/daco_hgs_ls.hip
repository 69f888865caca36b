// daco_hgs_ls.hip -- the reference's CVRP local search, route for route, one wavefront per solution (gfx950).
//
// Replaces, per ant, cvrp_nls/aco.py:114-126 -> swapstar.py:324-346 -> HGS-CVRP-main/Program/C_Interface.cpp:128-172:
// Params (Params.cpp:5-121), LocalSearch::run (LocalSearch.cpp:3-103) with the classical moves 1-9 under the
// 20-nearest granular restriction, first improvement in the order std::shuffle(std::minstd_rand) fixes,
// load penalties 10 * penaltyCapacity, export in route order (LocalSearch.cpp:756-778).  The reference's ctypes structure
// (swapstar.py:62-74, 10 fields against the header's 15) makes the C side read useSwapStar beyond the structure: the
// reference RUNS WITHOUT SWAP* and with all coordinates zero (Params.cpp:40-54), which is the mode built here
// (specification: oracle/hgs_ls.c with use_swap_star = 0, pinned on the reference's own library; tests/golden/g11_*).
//
// Arithmetic: float64 as HGS, the expressions of LocalSearch.cpp:134-484 term for term (-ffp-contract=off); every
// comparison the search branches on is the reference's.  Result: the reference's routes, entry for entry.
//
// Mapping.  A solution is a chain of ~10^2-10^3 DEPENDENT first-improvement steps: the parallelism inside one is the
// granular neighbourhood of the current node U (20-40 candidates V): one lane per V evaluates moves 1-9 (and the
// "insert after the depot" variants) in the reference's order and reports the first that applies; the lowest such lane
// is the move the reference would have taken; it is applied and the lanes above it re-evaluate on the new solution.
// Across solutions: B x A independent wavefronts (10^5 at BASELINE's config 4), fetched from a queue.
// State (linked routes, cumulated loads / reversal distances, route loads and penalties) lives in LDS, ~5 KB per
// solution at n = 100; the float64 matrix (82 KB per instance, shared by the instance's ants) is gathered from L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/deepaco_hip.h"
#include "daco_device.h"

namespace daco {

constexpr double HGS_EPS = 0.00001;                // Params.h:41
constexpr uint32_t MINSTD_M = 2147483647u;         // std::minstd_rand: x <- 48271 x mod (2^31 - 1)
constexpr int HGS_TABLE_HEADER = 64;
constexpr int HGS_MAX_STAGES = 3;

__host__ __device__ inline size_t a16(size_t x) { return (x + 15) & ~(size_t)15; }
// per instance (and matrix): header | orderNodes u16[nc] | len u16[n] | off u32[n] | entries u16[2 * g * nc] | bits u32[n][words]
struct HgsLayout {
  size_t order, len, off, ent, bits, total;
  int words;
  __host__ __device__ HgsLayout(int n, int g) {
    const int nc = n - 1;
    words = (n + 31) / 32;
    order = HGS_TABLE_HEADER;
    len = order + a16(2 * (size_t)nc);
    off = len + a16(2 * (size_t)n);
    ent = off + a16(4 * (size_t)n);
    bits = ent + a16(2 * (size_t)2 * g * nc);
    total = (bits + a16(4 * (size_t)n * words) + 255) & ~(size_t)255;
  }
};
struct HgsHeader { double maxDist; uint32_t rng_state; uint32_t entries; };

// ---------------------------------------------------------------------------------------------- RNG (libstdc++, GCC 11)
__device__ inline uint32_t minstd_next(uint32_t &x) {
  const uint64_t p = (uint64_t)x * 48271u;
  uint32_t r = (uint32_t)(p & MINSTD_M) + (uint32_t)(p >> 31);          // 2^31 == 1 (mod m)
  if (r >= MINSTD_M) r -= MINSTD_M;
  x = r;
  return r;
}
// uniform_int_distribution<unsigned long>{0, hi}: scaling = 2147483645 / (hi + 1), rejection above scaling * (hi + 1)
__device__ inline uint32_t uid(uint32_t &x, uint32_t hi) {
  const uint32_t uerange = hi + 1, scaling = 2147483645u / uerange, past = uerange * scaling;
  uint32_t r;
  do { r = minstd_next(x) - 1; } while (r >= past);
  return r / scaling;
}
// std::shuffle of v[0..n) (two positions per draw; bits/stl_algo.h); SWAP(i, j) exchanges elements i and j
template <typename SwapFn>
__device__ inline void shuffle_minstd(int n, uint32_t &x, SwapFn swap) {
  if (n <= 0) return;
  int i = 1;
  if ((n & 1) == 0) { const int j = (int)uid(x, 1); swap(i, j); ++i; }
  while (i != n) {
    const uint32_t b0 = (uint32_t)i + 1, b1 = b0 + 1;
    const uint32_t r = uid(x, b0 * b1 - 1);
    const int j0 = (int)(r / b1), j1 = (int)(r % b1);
    swap(i, j0); ++i;
    swap(i, j1); ++i;
  }
}

// ---------------------------------------------------------------------------------------------- prepare
// One workgroup per instance: maxDist, correlated vertices (Params.cpp:77-103), the node order of LocalSearch::run and the
// generator state behind it (both depend on the number of clients only).
__global__ __launch_bounds__(256) void hgs_prepare_kernel(int n, int g_req, const double *__restrict__ tc_all, long bstride,
                                                          unsigned char *__restrict__ tables, size_t table_bytes) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nc = n - 1;
  const double *tc = tc_all + (size_t)b * bstride;
  const HgsLayout lay(n, g_req);
  unsigned char *tab = tables + (size_t)b * table_bytes;
  HgsHeader *hdr = reinterpret_cast<HgsHeader *>(tab);
  uint16_t *order = reinterpret_cast<uint16_t *>(tab + lay.order);
  uint16_t *len = reinterpret_cast<uint16_t *>(tab + lay.len);
  uint32_t *off = reinterpret_cast<uint32_t *>(tab + lay.off);
  uint16_t *ent = reinterpret_cast<uint16_t *>(tab + lay.ent);
  uint32_t *bits = reinterpret_cast<uint32_t *>(tab + lay.bits);
  __shared__ double red[4];

  double mx = 0.;
  for (size_t k = tid; k < (size_t)n * n; k += 256) { const double v = tc[k]; if (v > mx) mx = v; }
  for (int o = 32; o; o >>= 1) { const double v = __shfl_xor(mx, o); if (v > mx) mx = v; }
  if (lane == 0) red[wave] = mx;
  for (size_t k = tid; k < (size_t)n * lay.words; k += 256) bits[k] = 0;
  __threadfence();
  __syncthreads();
  if (tid == 0) { double m = red[0]; for (int w = 1; w < 4; ++w) if (red[w] > m) m = red[w]; hdr->maxDist = m; }

  const int g = g_req < nc - 1 ? g_req : nc - 1;
  for (int i = 1 + wave; i <= nc; i += 4) {
    double ld = -1.; int lj = 0;                                         // last pick (d, j): the next is the smallest pair above it
    for (int it = 0; it < g; ++it) {
      double bd = 0.; int bj = 0x7fffffff;
      for (int j = 1 + lane; j <= nc; j += 64) {
        if (j == i) continue;
        const double d = tc[(size_t)i * n + j];
        const bool above = it == 0 || d > ld || (d == ld && j > lj);
        if (above && (bj == 0x7fffffff || d < bd || (d == bd && j < bj))) { bd = d; bj = j; }
      }
      for (int o = 32; o; o >>= 1) {
        const double od = __shfl_xor(bd, o); const int oj = __shfl_xor(bj, o);
        if (oj != 0x7fffffff && (bj == 0x7fffffff || od < bd || (od == bd && oj < bj))) { bd = od; bj = oj; }
      }
      ld = bd; lj = bj;
      if (lane == 0 && bj != 0x7fffffff) {
        atomicOr(&bits[(size_t)i * lay.words + (bj >> 5)], 1u << (bj & 31));
        atomicOr(&bits[(size_t)bj * lay.words + (i >> 5)], 1u << (i & 31));
      }
    }
  }
  __threadfence();
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    int c = 0;
    if (i > 0) for (int w = 0; w < lay.words; ++w) c += __popc(__builtin_nontemporal_load(&bits[(size_t)i * lay.words + w]));
    len[i] = (uint16_t)c;
  }
  __threadfence();
  __syncthreads();
  if (tid == 0) {
    uint32_t o = 0;
    for (int i = 0; i < n; ++i) { off[i] = o; o += __builtin_nontemporal_load(&len[i]); }
    hdr->entries = o;
    // Individual(params) shuffles a client permutation it then drops (Individual.cpp:30-35,38); LocalSearch::run shuffles
    // orderNodes (LocalSearch.cpp:9)
    uint32_t x = 1;                                                      // seed 0 and the seed the reference passes (1) both give state 1
    shuffle_minstd(nc, x, [](int, int) {});
    for (int k = 0; k < nc; ++k) order[k] = (uint16_t)(k + 1);
    shuffle_minstd(nc, x, [&](int p, int q) { const uint16_t t = order[p]; order[p] = order[q]; order[q] = t; });
    hdr->rng_state = x;
  }
  __threadfence();
  __syncthreads();
  for (int i = 1 + tid; i <= nc; i += 256) {
    uint32_t o = __builtin_nontemporal_load(&off[i]);
    for (int w = 0; w < lay.words; ++w) {
      uint32_t m = __builtin_nontemporal_load(&bits[(size_t)i * lay.words + w]);
      while (m) { const int bit = __builtin_ctz(m); m &= m - 1; ent[o++] = (uint16_t)(w * 32 + bit); }
    }
  }
}

// ---------------------------------------------------------------------------------------------- the search
struct HgsStage { const double *tc, *tct; long bstride; const unsigned char *tables; size_t table_bytes; int count; };
struct HgsParams {
  int B, n, A, Lmax, Rmax, g, nstages, budget;
  HgsStage st[HGS_MAX_STAGES];
  const double *demand;              // [B][n] as HGS gets them (swapstar.py:335: demands * 1000)
  double cap;
  int64_t *paths;
  int32_t *status, *stats;           // [B][A] / [B][A][4] or NULL
  uint32_t *queue;                   // work counter (zeroed by the launcher)
  uint16_t *scratch;                 // per resident wavefront: shuffled neighbour lists, 2 * g * nc entries
  size_t scratch_stride;
};

// LDS of one wavefront
struct HgsLds {
  uint16_t *next, *prev, *route, *pos;     // [N]   N = nc + 1 + 2 Rmax; nodes: 1..nc clients, nc+1+r / nc+1+R+r depots of route r
  double *cumLoad, *cumRev;                // [N]
  double *dNext;                           // [N]  timeCost[node][next(node)]: the edges of the routes, refreshed by the route update
  int32_t *whenRI;                         // [n]
  int32_t *rowptr;                         // [n]  >= 0: offset in the table's entries, < 0: -(1 + offset) in the wave's scratch
  double *rLoad, *rPen, *rRev;             // [Rmax]
  uint8_t *e2;                             // [n]  move of a node towards the empty route (valid while no move is applied)
  int32_t *rWhen, *rCnt;                   // [Rmax]
};
__host__ __device__ inline size_t hgs_lds_bytes(int n, int Rmax) {
  const size_t N = (size_t)n + 2 * Rmax;
  return a16(2 * N) * 4 + a16(8 * N) * 3 + a16(4 * (size_t)n) * 2 + a16(8 * (size_t)Rmax) * 3 + a16(4 * (size_t)Rmax) * 2 + a16((size_t)n);
}
__device__ inline HgsLds hgs_carve(unsigned char *p, int n, int Rmax) {
  const size_t N = (size_t)n + 2 * Rmax;
  HgsLds l;
  l.cumLoad = (double *)p; p += a16(8 * N);
  l.cumRev = (double *)p; p += a16(8 * N);
  l.dNext = (double *)p; p += a16(8 * N);
  l.rLoad = (double *)p; p += a16(8 * (size_t)Rmax);
  l.rPen = (double *)p; p += a16(8 * (size_t)Rmax);
  l.rRev = (double *)p; p += a16(8 * (size_t)Rmax);
  l.whenRI = (int32_t *)p; p += a16(4 * (size_t)n);
  l.rowptr = (int32_t *)p; p += a16(4 * (size_t)n);
  l.rWhen = (int32_t *)p; p += a16(4 * (size_t)Rmax);
  l.rCnt = (int32_t *)p; p += a16(4 * (size_t)Rmax);
  l.next = (uint16_t *)p; p += a16(2 * N);
  l.prev = (uint16_t *)p; p += a16(2 * N);
  l.route = (uint16_t *)p; p += a16(2 * N);
  l.pos = (uint16_t *)p; p += a16(2 * N);
  l.e2 = (uint8_t *)p;
  return l;
}

__device__ inline double rl_d(double v, int lane) {
  const uint64_t u = __double_as_longlong(v);
  const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)u, lane), hi = __builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), lane);
  return __longlong_as_double(((uint64_t)hi << 32) | lo);
}

struct HgsCtx {
  HgsLds l;
  const double *tc, *tct, *dem;         // tct: the transposed matrix (tc itself when symmetric)
  int n, nc, R, lane;
  double cap, penCap;
  int nbMoves;
  int budget, fail;                  // watchdog: every chased link / evaluation round costs one; fail = where it ran out
  __device__ inline bool tick(int where) { if (--budget < 0 && !fail) fail = where; return fail != 0; }
  __device__ inline int cour(int v) const { return v <= nc ? v : 0; }
  __device__ inline bool isdep(int v) const { return v > nc; }
  __device__ inline int dep(int r) const { return nc + 1 + r; }
  __device__ inline double TC(int a, int b) const { return tc[(size_t)a * n + b]; }
  // timeCost[b][a] read from row a of the transposed copy: with a wave-uniform `a` the lanes' entries share a few lines
  __device__ inline double TT(int a, int b) const { return tct[(size_t)a * n + b]; }
  __device__ inline double pen(double load) const { const double e = load - cap; return (e > 0. ? e : 0.) * penCap; }   // LocalSearch.h:140
};

// updateRouteData (LocalSearch.cpp:652-707) without the duration, barycentre and sector (unused without SWAP*: every
// coordinate is zero, Params.cpp:49-54).  The chain of nodes is chased once, the matrix entries of its edges are gathered
// by the lanes, the cumulated sums run in route order (one addition after the other, as the reference's loop).
__device__ inline void hgs_update_route(HgsCtx &c, int r) {
  const int lane = c.lane;
  int node = c.dep(r), place = 0;
  double load = 0., rev = 0.;
  if (lane == 0) { c.l.pos[node] = 0; c.l.cumLoad[node] = 0.; c.l.cumRev[node] = 0.; }
  bool done = false;
  while (!done) {
    int mine = 0, mprev = 0, cnt = 0;
    while (cnt < 64 && !done) {
      const int p = node;
      node = c.l.next[node];
      if (lane == cnt) { mine = node; mprev = p; }
      ++cnt;
      done = c.isdep(node) || c.tick(1);
    }
    double dl = 0., dr = 0.;
    if (lane < cnt) {
      const int cc = c.cour(mine), pc = c.cour(mprev);
      dl = c.dem[cc];
      const double fwd = c.TC(pc, cc);
      dr = c.TC(cc, pc) - fwd;
      c.l.dNext[mprev] = fwd;
    }
    double myl = 0., myr = 0.;
    for (int k = 0; k < cnt; ++k) {
      load += rl_d(dl, k);
      rev += rl_d(dr, k);
      if (lane == k) { myl = load; myr = rev; }
    }
    if (lane < cnt) { c.l.pos[mine] = (uint16_t)(place + lane + 1); c.l.cumLoad[mine] = myl; c.l.cumRev[mine] = myr; }
    place += cnt;
  }
  if (lane == 0) {
    c.l.dNext[node] = c.TC(0, 0);                      // the closing depot's own successor is the route's first depot
    c.l.rLoad[r] = load; c.l.rPen[r] = c.pen(load); c.l.rRev[r] = rev;
    c.l.rCnt[r] = place - 1; c.l.rWhen[r] = c.nbMoves;
  }
}

// the pair (U, V) as LocalSearch.cpp:105-132 sets it up; U side wave-uniform, V side per lane
struct USide {
  int U, X, Up, Xn, iU, iX, rU, prevU, nextX, posU;
  bool xDep;
  double loadU, loadX, penU, loadRU, cumLoadU, cumRevX, revDistU;
  double dUpU, dUX, dXXn, dUpX, dUpXn, dXU;
};
__device__ inline double uni_d(double v) {
  const uint64_t w = __double_as_longlong(v);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)w), hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(w >> 32));
  return __longlong_as_double(((uint64_t)hi << 32) | lo);
}
// UNIFORM: U is the same in every lane: the values are moved to scalar registers (40 vector registers less).  Measured at
// CVRP-100 x 512 x 256: not a gain -- the scalar file overflows into v_readlane / v_writelane traffic (three wavefronts per
// SIMD: 309 k solutions/s against 379 k with the values in vector registers; four with 49 spilled registers: 374 k) -- so the
// search loop instantiates <false>
template <bool UNIFORM>
__device__ inline void hgs_set_u(const HgsCtx &c, int U, USide &u) {
  u.U = U; u.X = c.l.next[U]; u.prevU = c.l.prev[U]; u.nextX = c.l.next[u.X];
  u.rU = c.l.route[U]; u.posU = c.l.pos[U];
  if (UNIFORM) {
    u.X = __builtin_amdgcn_readfirstlane(u.X); u.prevU = __builtin_amdgcn_readfirstlane(u.prevU);
    u.nextX = __builtin_amdgcn_readfirstlane(u.nextX); u.rU = __builtin_amdgcn_readfirstlane(u.rU);
    u.posU = __builtin_amdgcn_readfirstlane(u.posU);
  }
  u.iU = U; u.iX = c.cour(u.X); u.Up = c.cour(u.prevU); u.Xn = c.cour(u.nextX);
  u.xDep = c.isdep(u.X);
  u.loadU = c.dem[u.iU]; u.loadX = c.dem[u.iX];
  u.penU = c.l.rPen[u.rU]; u.loadRU = c.l.rLoad[u.rU]; u.revDistU = c.l.rRev[u.rU];
  u.cumLoadU = c.l.cumLoad[U]; u.cumRevX = c.l.cumRev[u.X];
  u.dUpU = c.l.dNext[u.prevU]; u.dUX = c.l.dNext[U]; u.dXXn = c.l.dNext[u.X];       // edges of the route: kept in LDS
  u.dUpX = c.TC(u.Up, u.iX); u.dUpXn = c.TC(u.Up, u.Xn); u.dXU = c.TC(u.iX, u.iU);
  if (UNIFORM) {
    u.loadU = uni_d(u.loadU); u.loadX = uni_d(u.loadX); u.penU = uni_d(u.penU); u.loadRU = uni_d(u.loadRU);
    u.revDistU = uni_d(u.revDistU); u.cumLoadU = uni_d(u.cumLoadU); u.cumRevX = uni_d(u.cumRevX);
    u.dUpU = uni_d(u.dUpU); u.dUX = uni_d(u.dUX); u.dXXn = uni_d(u.dXXn); u.dUpX = uni_d(u.dUpX); u.dUpXn = uni_d(u.dUpXn);
    u.dXU = uni_d(u.dXU);
  }
}

// First move of the reference's sequence that applies to (U, V), 0 if none.  block 0: LocalSearch.cpp:36-44 (moves 1-9),
// block 1: :47-56 (V = the depot in front of the route: 1, 2, 3, 8, 9), block 2: :62-71 (V = the depot of an empty route: 1, 2, 3, 9).
// HOIST (the latency mode): every entry the nine moves can ask for is fetched before the first of them is evaluated.  The moves
// return as they apply, so the compiler keeps each move's loads behind the previous move's branch: nine dependent waits per
// evaluation where one wavefront is alone on its SIMD.  Most (U, V) pairs walk all nine moves and need every entry anyway; with
// the loads up front the wavefront waits for memory once (LDS in this mode).  Same values, same arithmetic, same order of tests.
template <bool HOIST>
__device__ inline int hgs_eval(const HgsCtx &c, const USide &u, int V, int block) {
  const int Y = c.l.next[V], prevV = c.l.prev[V], nextY = c.l.next[Y];
  const int iV = c.cour(V), iY = c.cour(Y), Vp = c.cour(prevV), Yn = c.cour(nextY);
  const int rV = c.l.route[V];
  const bool intra = (u.rU == rV), yDep = c.isdep(Y);
  const double loadV = c.dem[iV], loadY = c.dem[iY];
  const double penV = c.l.rPen[rV], loadRV = c.l.rLoad[rV];
  // matrix entries: the routes' own edges from LDS; everything else has one index on the U side -- read from that node's row
  // of the matrix or of its transpose, so that the 64 lanes' gathers fall on the few lines of (at most six) wave-uniform rows
  const double dVY = c.l.dNext[V], dVU = c.TT(u.iU, iV), dUY = c.TC(u.iU, iY), dXY = c.TC(u.iX, iY);
  double h_dVX = 0., h_dUpV = 0., h_dVpU = 0., h_dVpV = 0., h_tXnV = 0., h_tXnY = 0., h_tXYn = 0., h_dNY = 0., h_tUV = 0., h_cumRevV = 0.,
         h_cumLoadV = 0.;
  int h_posV = 0, h_nextU = 0;
  if constexpr (HOIST) {
    h_dVX = c.TT(u.iX, iV);
    h_tUV = c.TC(u.iU, iV); h_cumRevV = c.l.cumRev[V]; h_cumLoadV = c.l.cumLoad[V];
    if (block == 0) {
      h_dUpV = c.TC(u.Up, iV); h_dVpU = c.TT(u.iU, Vp); h_dVpV = c.l.dNext[prevV];
      h_tXnV = c.TT(u.Xn, iV); h_tXnY = c.TT(u.Xn, iY); h_tXYn = c.TC(u.iX, Yn); h_dNY = c.l.dNext[Y];
      h_posV = (int)c.l.pos[V]; h_nextU = (int)c.l.next[u.U];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const double sumPen = u.penU + penV;
  {   // move1 (LocalSearch.cpp:134-162)
    double cU = u.dUpX - u.dUpU - u.dUX;
    double cV = dVU + dUY - dVY;
    bool ok = true;
    if (!intra) {
      if (cU + cV >= sumPen) ok = false;
      cU += c.pen(u.loadRU - u.loadU) - u.penU;
      cV += c.pen(loadRV + u.loadU) - penV;
    }
    if (ok && !(cU + cV > -HGS_EPS) && u.iU != iY) return 1;
  }
  {   // move2 (:164-193)
    double cU = u.dUpXn - u.dUpU - u.dXXn;
    double cV = dVU + dXY - dVY;
    bool ok = true;
    if (!intra) {
      if (cU + cV >= sumPen) ok = false;
      cU += c.pen(u.loadRU - u.loadU - u.loadX) - u.penU;
      cV += c.pen(loadRV + u.loadU + u.loadX) - penV;
    }
    if (ok && !(cU + cV > -HGS_EPS) && !(u.U == Y || V == u.X || u.xDep)) return 2;
  }
  const double dVX = HOIST ? h_dVX : c.TT(u.iX, iV);
  {   // move3 (:195-224)
    double cU = u.dUpXn - u.dUpU - u.dUX - u.dXXn;
    double cV = dVX + u.dXU + dUY - dVY;
    bool ok = true;
    if (!intra) {
      if (cU + cV >= sumPen) ok = false;
      cU += c.pen(u.loadRU - u.loadU - u.loadX) - u.penU;
      cV += c.pen(loadRV + u.loadU + u.loadX) - penV;
    }
    if (ok && !(cU + cV > -HGS_EPS) && !(u.U == Y || u.X == V || u.xDep)) return 3;
  }
  if (block == 0) {
    const double dUpV = HOIST ? h_dUpV : c.TC(u.Up, iV), dVpU = HOIST ? h_dVpU : c.TT(u.iU, Vp), dVpV = HOIST ? h_dVpV : c.l.dNext[prevV];
    if (u.iU <= iV) {   // move4 (:226-254)
      double cU = dUpV + dVX - u.dUpU - u.dUX;
      double cV = dVpU + dUY - dVpV - dVY;
      bool ok = true;
      if (!intra) {
        if (cU + cV >= sumPen) ok = false;
        cU += c.pen(u.loadRU + loadV - u.loadU) - u.penU;
        cV += c.pen(loadRV + u.loadU - loadV) - penV;
      }
      if (ok && !(cU + cV > -HGS_EPS) && !(u.iU == Vp || u.iU == iY)) return 4;
    }
    {   // move5 (:256-285)
      double cU = dUpV + (HOIST ? h_tXnV : c.TT(u.Xn, iV)) - u.dUpU - u.dXXn;
      double cV = dVpU + dXY - dVpV - dVY;
      bool ok = true;
      if (!intra) {
        if (cU + cV >= sumPen) ok = false;
        cU += c.pen(u.loadRU + loadV - u.loadU - u.loadX) - u.penU;
        cV += c.pen(loadRV + u.loadU + u.loadX - loadV) - penV;
      }
      if (ok && !(cU + cV > -HGS_EPS) && !(u.U == prevV || u.X == prevV || u.U == Y || u.xDep)) return 5;
    }
    if (u.iU <= iV) {   // move6 (:287-316)
      double cU = dUpV + (HOIST ? h_tXnY : c.TT(u.Xn, iY)) - u.dUpU - u.dXXn;
      double cV = dVpU + (HOIST ? h_tXYn : c.TC(u.iX, Yn)) - dVpV - (HOIST ? h_dNY : c.l.dNext[Y]);
      bool ok = true;
      if (!intra) {
        if (cU + cV >= sumPen) ok = false;
        cU += c.pen(u.loadRU + loadV + loadY - u.loadU - u.loadX) - u.penU;
        cV += c.pen(loadRV + u.loadU + u.loadX - loadV - loadY) - penV;
      }
      if (ok && !(cU + cV > -HGS_EPS) &&
          !(u.xDep || yDep || Y == u.prevU || u.U == Y || u.X == V || V == u.nextX)) return 6;
    }
    if (intra) {        // move7 (:318-352)
      if (!(u.posU > (HOIST ? h_posV : (int)c.l.pos[V]))) {
        const double cost = (HOIST ? h_tUV : c.TC(u.iU, iV)) + dXY - u.dUX - dVY + (HOIST ? h_cumRevV : c.l.cumRev[V]) - u.cumRevX;
        if (!(cost > -HGS_EPS) && (HOIST ? h_nextU : (int)c.l.next[u.U]) != V) return 7;
      }
    }
  }
  if ((block == 0 && !intra) || block == 1) {
    if (!intra) {       // move8 (:354-425)
      const double cumLoadV = HOIST ? h_cumLoadV : c.l.cumLoad[V];
      double cost = (HOIST ? h_tUV : c.TC(u.iU, iV)) + dXY - u.dUX - dVY + (HOIST ? h_cumRevV : c.l.cumRev[V]) + u.revDistU - u.cumRevX - u.penU - penV;
      if (!(cost >= 0)) {
        cost += c.pen(u.cumLoadU + cumLoadV) + c.pen(u.loadRU + loadRV - u.cumLoadU - cumLoadV);
        if (!(cost > -HGS_EPS)) return 8;
      }
    }
  }
  if (!intra || block == 2) {   // move9 (:427-484)
    const double cumLoadV = HOIST ? h_cumLoadV : c.l.cumLoad[V];
    double cost = dUY + dVX - u.dUX - dVY - u.penU - penV;
    if (!(cost >= 0)) {
      cost += c.pen(u.cumLoadU + loadRV - cumLoadV) + c.pen(cumLoadV + u.loadRU - u.cumLoadU);
      if (!(cost > -HGS_EPS)) return 9;
    }
  }
  return 0;
}

__device__ inline void hgs_insert_node(const HgsCtx &c, int U, int V) {           // LocalSearch.cpp:617-626
  if (c.lane == 0) {
    uint16_t *nx = c.l.next, *pv = c.l.prev;
    nx[pv[U]] = nx[U];
    pv[nx[U]] = pv[U];
    pv[nx[V]] = (uint16_t)U;
    pv[U] = (uint16_t)V;
    nx[U] = nx[V];
    nx[V] = (uint16_t)U;
    c.l.route[U] = c.l.route[V];
  }
}
__device__ inline void hgs_swap_node(const HgsCtx &c, int U, int V) {             // LocalSearch.cpp:628-650
  if (c.lane == 0) {
    uint16_t *nx = c.l.next, *pv = c.l.prev;
    const uint16_t vp = pv[V], vn = nx[V], up = pv[U], un = nx[U], ru = c.l.route[U], rv = c.l.route[V];
    nx[up] = (uint16_t)V; pv[un] = (uint16_t)V; nx[vp] = (uint16_t)U; pv[vn] = (uint16_t)U;
    pv[U] = vp; nx[U] = vn; pv[V] = up; nx[V] = un;
    c.l.route[U] = rv; c.l.route[V] = ru;
  }
}

// apply move `mv` to (U, V) (wave-uniform): the pointer surgery of LocalSearch.cpp:134-484, then the route data
__device__ inline void hgs_apply(HgsCtx &c, int mv, int U, int V) {
  uint16_t *nx = c.l.next, *pv = c.l.prev, *rt = c.l.route;
  const int X = nx[U], Y = nx[V], rU = rt[U], rV = rt[V];
  const bool intra = rU == rV;
  switch (mv) {
    case 1: hgs_insert_node(c, U, V); break;
    case 2: hgs_insert_node(c, U, V); hgs_insert_node(c, X, U); break;
    case 3: hgs_insert_node(c, X, V); hgs_insert_node(c, U, X); break;
    case 4: hgs_swap_node(c, U, V); break;
    case 5: hgs_swap_node(c, U, V); hgs_insert_node(c, X, U); break;
    case 6: hgs_swap_node(c, U, V); hgs_swap_node(c, X, Y); break;
    case 7:
      if (c.lane == 0) {
        int node = nx[X];
        pv[X] = (uint16_t)node; nx[X] = (uint16_t)Y;
        int guard = 0;
        while (node != V && ++guard < 70000) { const int t = nx[node]; nx[node] = pv[node]; pv[node] = (uint16_t)t; node = t; }
        nx[V] = pv[V]; pv[V] = (uint16_t)U; nx[U] = (uint16_t)V; pv[Y] = (uint16_t)X;
      }
      break;
    case 8:
      if (c.lane == 0) {
        const int depU = c.dep(rU), depV = c.dep(rV), depUFin = pv[depU], depVFin = pv[depV], depVNext = nx[depV];
        int xx = X, vv = V, t;
        int guard = 0;
        while (!c.isdep(xx) && ++guard < 70000) { t = nx[xx]; nx[xx] = pv[xx]; pv[xx] = (uint16_t)t; rt[xx] = (uint16_t)rV; xx = t; }
        guard = 0;
        while (!c.isdep(vv) && ++guard < 70000) { t = pv[vv]; pv[vv] = nx[vv]; nx[vv] = (uint16_t)t; rt[vv] = (uint16_t)rU; vv = t; }
        nx[U] = (uint16_t)V; pv[V] = (uint16_t)U; nx[X] = (uint16_t)Y; pv[Y] = (uint16_t)X;
        if (c.isdep(X)) {
          nx[depUFin] = (uint16_t)depU; pv[depUFin] = (uint16_t)depVNext; nx[pv[depUFin]] = (uint16_t)depUFin;
          nx[depV] = (uint16_t)Y; pv[Y] = (uint16_t)depV;
        } else if (c.isdep(V)) {
          nx[depV] = pv[depUFin]; pv[nx[depV]] = (uint16_t)depV; pv[depV] = (uint16_t)depVFin;
          pv[depUFin] = (uint16_t)U; nx[U] = (uint16_t)depUFin;
        } else {
          nx[depV] = pv[depUFin]; pv[nx[depV]] = (uint16_t)depV;
          pv[depUFin] = (uint16_t)depVNext; nx[pv[depUFin]] = (uint16_t)depUFin;
        }
      }
      break;
    default:  // 9
      if (c.lane == 0) {
        const int depU = c.dep(rU), depV = c.dep(rV), depUFin = pv[depU], depVFin = pv[depV], depUpred = pv[depUFin];
        int k = Y;
        int guard = 0;
        while (!c.isdep(k) && ++guard < 70000) { rt[k] = (uint16_t)rU; k = nx[k]; }
        k = X;
        guard = 0;
        while (!c.isdep(k) && ++guard < 70000) { rt[k] = (uint16_t)rV; k = nx[k]; }
        nx[U] = (uint16_t)Y; pv[Y] = (uint16_t)U; nx[V] = (uint16_t)X; pv[X] = (uint16_t)V;
        if (c.isdep(X)) {
          pv[depUFin] = pv[depVFin]; nx[pv[depUFin]] = (uint16_t)depUFin;
          nx[V] = (uint16_t)depVFin; pv[depVFin] = (uint16_t)V;
        } else {
          pv[depUFin] = pv[depVFin]; nx[pv[depUFin]] = (uint16_t)depUFin;
          pv[depVFin] = (uint16_t)depUpred; nx[pv[depVFin]] = (uint16_t)depVFin;
        }
      }
      break;
  }
  c.nbMoves++;
  hgs_update_route(c, rU);
  if (!intra) hgs_update_route(c, rV);
}

// LM (round 6, "latency mode"): a launch of FEW solutions -- the reference's own calling pattern hands the local search the 8
// best ants of ONE instance per iteration (cvrp_nls/aco.py:143-146) -- has one wavefront alone on its SIMD and nothing to overlap:
// a round is a chain of instruction latency and four or five dependent trips to the L2 for matrix entries (3.7 us per round, 7.2 ms
// for eight solutions of CVRP-100 against ~3 ms for the reference's eight thread-pool tasks).  Here a workgroup is one
// wavefront that keeps the stage's float64 matrix (n^2 x 8 bytes: 82 KB at n = 101) and the demands in LDS next to its state, so
// those trips are LDS gathers; an asymmetric matrix keeps its transpose in global memory (the two do not fit together).  The
// arithmetic, the order of evaluation and every decision are the kernel's above: the routes are the same.
template <int WAVES, int WPS, bool LM = false>
__global__ __launch_bounds__(WAVES * 64, WPS) void hgs_ls_kernel(const HgsParams p) {
  static_assert(!LM || WAVES == 1, "latency mode: one wavefront per workgroup");
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = p.n, nc = n - 1;
  const size_t per_wave = hgs_lds_bytes(n, p.Rmax);
  double *mat_lds = reinterpret_cast<double *>(lds_raw + per_wave);          // LM: [n][n], then the demands [n]
  double *dem_lds = mat_lds + (size_t)n * n;
  HgsCtx c;
  c.l = hgs_carve(lds_raw + (size_t)wave * per_wave, n, p.Rmax);
  c.n = n; c.nc = nc; c.lane = lane; c.cap = p.cap;
  uint16_t *scratch = p.scratch + (size_t)(blockIdx.x * WAVES + wave) * p.scratch_stride;
  const int nitems = p.B * p.A;
  const HgsLayout lay(n, p.g);

  for (;;) {
    // one lane takes the next item and the wave reads it back: a single returning atomic, written out as such (the
    // form "every lane adds one, item = old / 64" was right only while the compiler folded the 64 adds into one).  The lane is
    // selected INSIDE the asm statement (exec = 1 around the instruction): an `if (lane == 0)` around the atomic -- as an
    // intrinsic or as inline asm, both were tried -- makes the compiler treat everything derived from `item` as divergent and
    // wrap the item's processing in a loop over exec subsets in which the other lanes' zero re-runs item 0 for ever.
    uint32_t fetched;
    {
      unsigned long long exec_save;
      asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, off sc0\n\ts_waitcnt vmcnt(0)\n\ts_mov_b64 exec, %1"
                   : "=&v"(fetched), "=&s"(exec_save) : "v"(p.queue), "v"(1u) : "memory");
    }
    const int item = (int)__builtin_amdgcn_readfirstlane(fetched);
    if (item >= nitems) break;
    const int b = item / p.A, a = item - b * p.A;
    int64_t *col = p.paths + (size_t)b * p.Lmax * p.A + a;
    c.dem = p.demand + (size_t)b * n;
    if constexpr (LM) {
      for (int i = lane; i < n; i += 64) dem_lds[i] = c.dem[i];
      c.dem = dem_lds;
    }
    c.fail = 0;

    // ---- the routes of the input (cvrp_nls/aco.py:12-20 get_subroutes: the non-empty pieces between zeros)
    int R = 0, status = 0, totalMoves = 0, totalLoops = 0, nRounds = 0;
#ifdef DACO_HGS_PROFILE   // (measurement build: shader-clock cycles per phase instead of loops / rounds / watchdog in the stats)
    unsigned long long pf_eval = 0, pf_apply = 0, pf_other = 0, pf_t0 = 0;
    const unsigned long long pf_begin = __builtin_readcyclecounter();
#define PF_START() (pf_t0 = __builtin_readcyclecounter())
#define PF_ADD(acc) ((acc) += __builtin_readcyclecounter() - pf_t0)
#else
#define PF_START() ((void)0)
#define PF_ADD(acc) ((void)0)
#endif
    {   // pass A: entries in range, number of routes, every client exactly once (Individual.cpp:69)
      int last = 0, seen = 0;
      bool bad = false;
      for (int t0 = 0; t0 < p.Lmax; t0 += 64) {
        const int t = t0 + lane;
        const int v = t < p.Lmax ? (int)col[(size_t)t * p.A] : 0;
        const int pvv = __builtin_amdgcn_ds_bpermute(((lane + 63) & 63) << 2, v);
        const int before = lane == 0 ? last : pvv;
        bad = bad || v < 0 || v > nc;
        R += __popcll(__ballot(v != 0 && before == 0));
        seen += __popcll(__ballot(v != 0));
        if (v > 0 && v <= nc) c.l.whenRI[v] = t;                       // (scratch use: the position that lists the client)
        last = __builtin_amdgcn_readlane(v, 63);
      }
      if (__any(bad) || R > p.Rmax) status = 2;                         // not a route sequence of this instance: left untouched
      else {
        bool dup = false;
        for (int t0 = 0; t0 < p.Lmax; t0 += 64) {
          const int t = t0 + lane;
          const int v = t < p.Lmax ? (int)col[(size_t)t * p.A] : 0;
          if (v != 0 && c.l.whenRI[v] != t) dup = true;
        }
        if (__any(dup) || seen != nc) status = 2;                       // HGS throws (or walks a broken list): left untouched
      }
    }
    c.R = R;
    if (!status) {  // pass B: the links.  Route r: start depot nc+1+r, end depot nc+1+R+r
      int last = 0, rbase = 0;
      for (int t0 = 0; t0 < p.Lmax; t0 += 64) {
        const int t = t0 + lane;
        const int v = t < p.Lmax ? (int)col[(size_t)t * p.A] : 0;
        const int nv = t + 1 < p.Lmax ? (int)col[(size_t)(t + 1) * p.A] : 0;
        const int pvv = __builtin_amdgcn_ds_bpermute(((lane + 63) & 63) << 2, v);
        const int before = lane == 0 ? last : pvv;
        const uint64_t sm = __ballot(v != 0 && before == 0);
        const int rid = rbase + __popcll(sm & ((2ull << lane) - 1ull)) - 1;
        if (v != 0) {
          c.l.route[v] = (uint16_t)rid;
          if (before != 0) c.l.prev[v] = (uint16_t)before;
          else { c.l.prev[v] = (uint16_t)(nc + 1 + rid); c.l.next[nc + 1 + rid] = (uint16_t)v; }
          if (nv != 0) c.l.next[v] = (uint16_t)nv;
          else { c.l.next[v] = (uint16_t)(nc + 1 + R + rid); c.l.prev[nc + 1 + R + rid] = (uint16_t)v; }
        }
        rbase += __popcll(sm);
        last = __builtin_amdgcn_readlane(v, 63);
      }
      for (int r = lane; r < R; r += 64) {
        const int d0 = nc + 1 + r, d1 = nc + 1 + R + r;
        c.l.route[d0] = (uint16_t)r; c.l.route[d1] = (uint16_t)r;
        c.l.prev[d0] = (uint16_t)d1; c.l.next[d1] = (uint16_t)d0;
      }
    }
    // Params.cpp:29-37: largest and total demand (the total in index order, as the reference adds it)
    double maxDem = 0., totDem = 0.;
    for (int i0 = 0; i0 <= nc; i0 += 64) {
      const double d = i0 + lane <= nc ? c.dem[i0 + lane] : 0.;
      const int cnt = nc + 1 - i0 < 64 ? nc + 1 - i0 : 64;
      for (int k = 0; k < cnt; ++k) { const double dk = rl_d(d, k); totDem += dk; if (dk > maxDem) maxDem = dk; }
    }

    // ---- the stages (cvrp_nls/aco.py:443-448: one local_search call each)
    for (int sgi = 0; sgi < p.nstages && status != 2; ++sgi) {
      const HgsStage &sg = p.st[sgi];
      const unsigned char *tab = sg.tables + (size_t)b * sg.table_bytes;
      const HgsHeader *hdr = reinterpret_cast<const HgsHeader *>(tab);
      const uint16_t *order = reinterpret_cast<const uint16_t *>(tab + lay.order);
      const uint16_t *tlen = reinterpret_cast<const uint16_t *>(tab + lay.len);
      const uint32_t *toff = reinterpret_cast<const uint32_t *>(tab + lay.off);
      const uint16_t *tent = reinterpret_cast<const uint16_t *>(tab + lay.ent);
      c.tc = sg.tc + (size_t)b * sg.bstride;
      c.tct = sg.tct + (size_t)b * sg.bstride;
      if constexpr (LM) {
        const double *src = c.tc;
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < n * n; i += 64) mat_lds[i] = src[i];
        __threadfence_block();
        if (sg.tct == sg.tc) c.tct = mat_lds;
        c.tc = mat_lds;
      }
      R = c.R;
      // Params: scale checks, penalty (Params.cpp:106-118)
      const double maxDist = hdr->maxDist;
      bool thrown = maxDist < 0.1 || maxDist > 100000 || maxDem < 0.1 || maxDem > 100000 ||
                    (double)R < ceil(totDem / c.cap);
      double pc = maxDist / maxDem; if (pc > 1000.) pc = 1000.; if (pc < 0.1) pc = 0.1;
      c.penCap = pc * 10.;
      c.nbMoves = 0;
      c.budget = p.budget;
      if (!thrown) {
        for (int r = 0; r < R; ++r) hgs_update_route(c, r);                    // loadIndividual (LocalSearch.cpp:709-754)
        // Individual.cpp:26-35,70: feasible within MY_EPSILON, or it throws
        double excess = 0.;
        for (int r = 0; r < R; ++r) { const double ld = c.l.rLoad[r]; if (ld > c.cap) excess += ld - c.cap; }
        if (!(excess < HGS_EPS)) thrown = true;
      }
      if (thrown) { status = 1; continue; }                                    // swapstar.py:341-345: the input routes stay
      for (int i = lane; i <= nc; i += 64) { c.l.whenRI[i] = -1; c.l.rowptr[i] = (int32_t)toff[i]; }

      // LocalSearch.cpp:9-14: orderNodes is in the table; orderRoutes is shuffled (its draws count, its order is only used by
      // SWAP*); every client's neighbour list is shuffled with probability 1 / nbGranular
      uint32_t x = hdr->rng_state;
      shuffle_minstd(R, x, [](int, int) {});
      {
        uint32_t so = 0;
        for (int i = 1; i <= nc; ++i) {
          if (minstd_next(x) % (uint32_t)p.g == 0) {
            const int ln = tlen[i];
            const uint16_t *src = tent + toff[i];
            uint16_t *dst = scratch + so;
            for (int k = lane; k < ln; k += 64) dst[k] = src[k];
            if (lane == 0) {
              c.l.rowptr[i] = -(int32_t)(1 + so);
              shuffle_minstd(ln, x, [&](int pa, int pb) { const uint16_t t = dst[pa]; dst[pa] = dst[pb]; dst[pb] = t; });
            } else {
              shuffle_minstd(ln, x, [](int, int) {});
            }
            so += (uint32_t)ln;
          }
        }
      }
      __threadfence_block();

      bool searchCompleted = false;
      int loops = 0, e2stamp = -1, e2er = -1;
      for (int loopID = 0; !searchCompleted && loopID <= sg.count && !c.fail; ++loopID) {
        ++loops;
        if (loopID > 1) searchCompleted = true;
        // the neighbour list of the next node is fetched while the current one is evaluated (one L2 round trip less per node)
        int nU = order[0];
        int nLn = tlen[nU], nRp = c.l.rowptr[nU];
        int nV = lane < nLn ? (int)(nRp >= 0 ? tent + nRp : scratch + (-(nRp + 1)))[lane] : 0;
        for (int posU = 0; posU < nc && !c.fail; ++posU) {
          const int U = nU, ln = nLn, rp = nRp;
          const int firstV = nV;
          if (posU + 1 < nc) {
            nU = order[posU + 1];
            nLn = tlen[nU]; nRp = c.l.rowptr[nU];
            nV = lane < nLn ? (int)(nRp >= 0 ? tent + nRp : scratch + (-(nRp + 1)))[lane] : 0;
          }
          const int lastTest = c.l.whenRI[U];
          if (lane == 0) c.l.whenRI[U] = c.nbMoves;
          const uint16_t *list = rp >= 0 ? tent + rp : scratch + (-(rp + 1));
          USide u;
          for (int ch = 0; ch < ln; ch += 64) {
            const int myV = ch == 0 ? firstV : (ch + lane < ln ? (int)list[ch + lane] : 0);
            int start = 0;
            for (;;) {
              if (c.tick(2)) break;
              // who is to be evaluated (LocalSearch.cpp:32): LDS only -- after the first loop most nodes have nobody
              const int rUn = c.l.route[U];
              const int wu = c.l.rWhen[rUn];
              bool act = lane >= start && myV != 0;
              const int Vs = myV != 0 ? myV : U;                              // lanes without a candidate look at U itself (masked)
              if (loopID != 0) { const int wv = c.l.rWhen[c.l.route[Vs]]; act = act && (wu > wv ? wu : wv) > lastTest; }
              if (!__ballot(act)) break;
              ++nRounds;
              PF_START();
              hgs_set_u<false>(c, U, u);
              // the lanes that are not evaluated look at U itself: their matrix gathers fall on the lines the U side reads
              // anyway (a gated-off lane that gathered ITS V's entries cost the data-return unit as much as a live one: the
              // kernel sits on that unit, TD busy 0.92 in profiles/r05_pmc_hgs_ls.txt)
              const int Ve = act ? Vs : U;
              const int pvn = c.l.prev[Ve];
              const bool depPrev = c.isdep(pvn);
              const int c0 = hgs_eval<LM>(c, u, Ve, 0);
              int code = act ? c0 : 0;
              // "insert after the depot" (LocalSearch.cpp:47-56) for the lanes whose V opens its route and found nothing: only
              // when such a lane exists (in the late loops few lanes are evaluated at all)
              const bool need1 = act && c0 == 0 && depPrev;
              if (__ballot(need1)) {
                const int c1 = hgs_eval<LM>(c, u, need1 ? pvn : U, 1);
                if (need1 && c1) code = c1 + 16;
              }
              const uint64_t m = __ballot(code != 0);
              PF_ADD(pf_eval);
              if (!m) break;
              const int pl = __builtin_ctzll(m);
              PF_START();
              const int mv = __builtin_amdgcn_readlane(code, pl);
              int V = __builtin_amdgcn_readlane(myV, pl);
              if (mv & 16) V = c.l.prev[V];
              hgs_apply(c, mv & 15, U, V);
              PF_ADD(pf_apply);
              searchCompleted = false;
              start = pl + 1;
              if (start >= 64) break;
            }
          }
          if (loopID > 0) {                                                   // LocalSearch.cpp:60-71: an empty route
            PF_START();
            int er = -1;
            for (int r0 = 0; r0 < R && er < 0; r0 += 64) {
              const uint64_t m = __ballot(r0 + lane < R && c.l.rCnt[r0 + lane] == 0);
              if (m) er = r0 + __builtin_ctzll(m);
            }
            if (er >= 0) {
              // moves 1, 2, 3, 9 of U towards the empty route depend on the solution only: evaluated for 64 nodes at a time
              // (one lane per node) and kept until the next move is applied -- in the late loops that is one evaluation
              // per 64 nodes instead of one per node
              if (e2stamp != c.nbMoves || e2er != er) {
                for (int u0 = 1; u0 <= nc; u0 += 64) {
                  const int Ul = u0 + lane <= nc ? u0 + lane : 1;
                  USide ul;
                  hgs_set_u<false>(c, Ul, ul);
                  const int mvl = hgs_eval<LM>(c, ul, c.dep(er), 2);
                  if (u0 + lane <= nc) c.l.e2[Ul] = (uint8_t)mvl;
                }
                e2stamp = c.nbMoves; e2er = er;
              }
              const int mv = c.l.e2[U];
              if (mv) { hgs_apply(c, mv, U, c.dep(er)); searchCompleted = false; }
            }
            PF_ADD(pf_other);
          }
        }
      }
      totalMoves += c.nbMoves; totalLoops += loops;
      if (c.fail) { status = 100 + c.fail; break; }

      // exportIndividual: every barycentre angle is atan2(0, 0) = 0 (1e30 for an empty route): routes in index order, the
      // empty ones dropped (Individual.cpp:85-102).  Between stages the next Params sees the non-empty routes, renumbered:
      // the links are rebuilt through the sequence in the wave's scratch-free way: renumber in place.
      if (sgi + 1 < p.nstages) {
        // new route index = rank among the non-empty ones; the depots move to their new ids (nc + 1 + r', nc + 1 + R' + r')
        int Rn = 0;
        for (int r = 0; r < R; ++r) {
          const int first = c.l.next[nc + 1 + r];
          if (c.isdep(first)) continue;
          const int lastn = c.l.prev[nc + 1 + R + r];
          // nodes of the route take the new route id as they are walked by update_route of the next stage: set here
          if (lane == 0) {
            int node = first;
            while (!c.isdep(node)) { c.l.route[node] = (uint16_t)Rn; node = c.l.next[node]; }
          }
          // (start depots only move down: nc + 1 + Rn <= nc + 1 + r; end depots are re-linked after the loop, R' is not known yet)
          if (lane == 0) {
            c.l.next[nc + 1 + Rn] = (uint16_t)first; c.l.prev[first] = (uint16_t)(nc + 1 + Rn);
            c.l.cumLoad[nc + 1 + Rn] = 0.;        // scratch: remember the last node of new route Rn in pos[] of its start depot
            c.l.pos[nc + 1 + Rn] = (uint16_t)lastn;
          }
          ++Rn;
        }
        if (lane == 0) {
          for (int r = 0; r < Rn; ++r) {
            const int d0 = nc + 1 + r, d1 = nc + 1 + Rn + r, lastn = c.l.pos[d0];
            c.l.next[lastn] = (uint16_t)d1; c.l.prev[d1] = (uint16_t)lastn;
            c.l.prev[d0] = (uint16_t)d1; c.l.next[d1] = (uint16_t)d0;
            c.l.route[d0] = c.l.route[d1] = (uint16_t)r;
          }
        }
        c.R = Rn;
      }
    }

    // ---- write the column back (cvrp_nls/aco.py:22-33 merge_subroutes: "0 c1 .. ck" per non-empty route, zero padded)
    if (status != 2 && status < 100) {
      int base = 0;
      for (int r = 0; r < c.R; ++r) {
        int node = c.l.next[nc + 1 + r];
        if (c.isdep(node)) continue;
        if (lane == 0) col[(size_t)base * p.A] = 0;
        ++base;
        bool done = false;
        int first = node;
        while (!done) {                                   // the chain is chased once per 64 nodes, the lanes write their entry
          int mine = 0, cnt = 0;
          node = first;
          while (cnt < 64 && !done) {
            if (lane == cnt) mine = node;
            ++cnt;
            node = c.l.next[node];
            done = c.isdep(node) || base + cnt >= p.Lmax;
          }
          if (lane < cnt) col[(size_t)(base + lane) * p.A] = mine;
          base += cnt;
          first = node;
        }
      }
      for (int t = base + lane; t < p.Lmax; t += 64) col[(size_t)t * p.A] = 0;
    }
    if (lane == 0) {
      if (p.status) p.status[item] = status;
      if (p.stats) { p.stats[(size_t)item * 4] = totalMoves; p.stats[(size_t)item * 4 + 1] = totalLoops; p.stats[(size_t)item * 4 + 2] = nRounds; p.stats[(size_t)item * 4 + 3] = c.fail; }
#ifdef DACO_HGS_PROFILE
      if (p.stats) {
        p.stats[(size_t)item * 4 + 0] = (int)((__builtin_readcyclecounter() - pf_begin) >> 10);
        p.stats[(size_t)item * 4 + 1] = (int)(pf_eval >> 10); p.stats[(size_t)item * 4 + 2] = (int)(pf_apply >> 10); p.stats[(size_t)item * 4 + 3] = (int)(pf_other >> 10);
      }
#endif
    }
  }
}

}  // namespace daco

using namespace daco;

extern "C" size_t daco_hgs_table_bytes(int n, int nb_granular) {
  if (n < 2 || nb_granular < 1) return 0;
  return HgsLayout(n, nb_granular).total;
}

extern "C" int daco_hgs_prepare(void *stream, int B, int n, const double *matrix, long bstride, int nb_granular, void *tables) {
  if (B <= 0 || n < 2 || n > 16000 || !matrix || !tables || nb_granular < 1 || nb_granular > 64) {
    set_error("daco_hgs_prepare: bad argument (B=%d n=%d nb_granular=%d)", B, n, nb_granular);
    return DACO_E_BADARG;
  }
  hipLaunchKernelGGL(hgs_prepare_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, n, nb_granular, matrix, bstride,
                     (unsigned char *)tables, HgsLayout(n, nb_granular).total);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("hgs_prepare_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}

// wavefronts per SIMD the kernel is compiled for: 168 registers.  Measured at CVRP-100 x 512 ants: two (198 registers) 232 k, three
// 309 k, four (128 registers, 49 of them spilled) 283 k solutions/s at 16 instances, 374 k against 384 k at 256.
constexpr int HGS_WPS = 3;

static int hgs_grid(int n, int Rmax, int *waves_out, size_t *lds_out) {
  const size_t per_wave = hgs_lds_bytes(n, Rmax);
  int waves = 4;
  while (waves > 1 && per_wave * waves > 64 * 1024) waves >>= 1;
  *waves_out = waves;
  *lds_out = per_wave * waves;
  static int cus = 0;                                  // (all devices of a node are the same part)
  if (!cus) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    else cus = 256;
  }
  const size_t lds_cu = 160 * 1024;
  int wg_per_cu = (int)(lds_cu / (*lds_out ? *lds_out : 1));
  const int cap = 4 * HGS_WPS / waves > 0 ? 4 * HGS_WPS / waves : 1;      // wavefronts per CU the register budget of the kernel allows
  if (wg_per_cu > cap) wg_per_cu = cap;
  if (wg_per_cu < 1) wg_per_cu = 1;
  return cus * wg_per_cu;
}

extern "C" size_t daco_hgs_workspace_bytes(int B, int n, int A, int Lmax, int nb_granular) {
  if (B <= 0 || n < 2 || A <= 0 || Lmax < 3 || nb_granular < 1) return 0;
  int Rmax = Lmax - (n - 1); if (Rmax > n - 1) Rmax = n - 1; if (Rmax < 1) Rmax = 1;
  int waves; size_t lds;
  const int grid = hgs_grid(n, Rmax, &waves, &lds);
  const size_t stride = a16(2 * (size_t)2 * nb_granular * (n - 1)) / 2;
  return 256 + (size_t)grid * waves * stride * 2;
}

extern "C" int daco_hgs_local_search(void *stream, int B, int n, int A, int Lmax, int nstages, const double *const *matrices,
                                     const double *const *matrices_t, const long *bstrides, const void *const *tables, const int *counts, const double *demand,
                                     double capacity, int nb_granular, int64_t *paths, int32_t *status, int32_t *stats,
                                     void *workspace, size_t workspace_bytes) {
  if (B <= 0 || n < 2 || A <= 0 || Lmax < 3 || nstages < 1 || nstages > HGS_MAX_STAGES || !matrices || !bstrides || !tables ||
      !counts || !demand || !paths || !workspace || nb_granular < 1 || nb_granular > 64) {
    set_error("daco_hgs_local_search: bad argument (B=%d n=%d A=%d Lmax=%d stages=%d)", B, n, A, Lmax, nstages);
    return DACO_E_BADARG;
  }
  if (n > 16000) { set_error("daco_hgs_local_search: n=%d is above 16000 (node ids are 16-bit with two depots per route)", n); return DACO_E_TOOLARGE; }
  if (workspace_bytes < daco_hgs_workspace_bytes(B, n, A, Lmax, nb_granular)) {
    set_error("daco_hgs_local_search: workspace of %zu bytes, %zu needed", workspace_bytes, daco_hgs_workspace_bytes(B, n, A, Lmax, nb_granular));
    return DACO_E_BADARG;
  }
  HgsParams p;
  p.B = B; p.n = n; p.A = A; p.Lmax = Lmax; p.g = nb_granular; p.nstages = nstages;
  int Rmax = Lmax - (n - 1); if (Rmax > n - 1) Rmax = n - 1; if (Rmax < 1) Rmax = 1;
  p.Rmax = Rmax;
  // watchdog of one stage of one solution (links chased + evaluation rounds): far above any search, finite on a broken table
  static const int budget_env = getenv("DACO_HGS_BUDGET") ? atoi(getenv("DACO_HGS_BUDGET")) : 0;
  p.budget = budget_env > 0 ? budget_env : 0x7fffffff;
  if (n + 2 * Rmax > 65535) { set_error("daco_hgs_local_search: n + 2 routes = %d does not fit 16-bit node ids", n + 2 * Rmax); return DACO_E_TOOLARGE; }
  for (int s = 0; s < nstages; ++s) {
    if (!matrices[s] || !tables[s] || counts[s] < 0) { set_error("daco_hgs_local_search: stage %d: null matrix / table or negative count", s); return DACO_E_BADARG; }
    p.st[s].tc = matrices[s]; p.st[s].tct = (matrices_t && matrices_t[s]) ? matrices_t[s] : matrices[s]; p.st[s].bstride = bstrides[s]; p.st[s].tables = (const unsigned char *)tables[s];
    p.st[s].table_bytes = HgsLayout(n, nb_granular).total; p.st[s].count = counts[s];
  }
  p.demand = demand; p.cap = capacity; p.paths = paths; p.status = status; p.stats = stats;
  p.queue = (uint32_t *)workspace;
  p.scratch = (uint16_t *)((unsigned char *)workspace + 256);
  p.scratch_stride = a16(2 * (size_t)2 * nb_granular * (n - 1)) / 2;
  int waves; size_t lds;
  int grid = hgs_grid(n, Rmax, &waves, &lds);
  if (lds > 160 * 1024) { set_error("daco_hgs_local_search: %zu bytes of LDS per solution (n=%d, up to %d routes)", lds, n, Rmax); return DACO_E_TOOLARGE; }
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = zero_async(workspace, 256, st);
  if (e != hipSuccess) { set_error("daco_hgs_local_search: clearing the queue: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  // latency mode (hgs_ls_kernel<1, .., true>): no more solutions than a quarter of the CUs (each then has a CU's LDS to itself and
  // nothing would have overlapped anyway) and the matrix fits.  DACO_HGS_LATENCY=0 / 1 forces (1: whenever it fits).
  {
    const size_t lm_lds = hgs_lds_bytes(n, Rmax) + ((size_t)n * n + n) * sizeof(double);
    const char *ev = getenv("DACO_HGS_LATENCY");
    const int force = ev ? atoi(ev) : -1;
    const long items = (long)B * A;
    if (lm_lds <= 160 * 1024 - 256 && force != 0 && (force == 1 || items <= 64)) {
      e = hipFuncSetAttribute((const void *)hgs_ls_kernel<1, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm_lds);
      if (e != hipSuccess) { set_error("daco_hgs_local_search: cannot reserve %zu bytes of LDS: %s", lm_lds, hipGetErrorString(e)); return DACO_E_HIP; }
      // (the scratch stride was sized for hgs_grid's wavefronts: one wavefront per workgroup here, never more workgroups than that)
      const long lm_grid = items < (long)grid * waves ? items : (long)grid * waves;
      hipLaunchKernelGGL((hgs_ls_kernel<1, 1, true>), dim3((unsigned)lm_grid), dim3(64), lm_lds, st, p);
      e = hipGetLastError();
      if (e != hipSuccess) { set_error("hgs_ls_kernel (latency mode) launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
      return DACO_OK;
    }
  }
#define DACO_HGS_LAUNCH(W_, S_)                                                                                                   \
  do {                                                                                                                              \
    if (lds > 64 * 1024) {                                                                                                          \
      e = hipFuncSetAttribute((const void *)hgs_ls_kernel<W_, S_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
      if (e != hipSuccess) { set_error("daco_hgs_local_search: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return DACO_E_HIP; } \
    }                                                                                                                               \
    hipLaunchKernelGGL((hgs_ls_kernel<W_, S_>), dim3(grid), dim3(W_ * 64), lds, st, p);                                             \
  } while (0)
  if (waves == 4) DACO_HGS_LAUNCH(4, HGS_WPS);
  else if (waves == 2) DACO_HGS_LAUNCH(2, HGS_WPS);
  else DACO_HGS_LAUNCH(1, HGS_WPS);
#undef DACO_HGS_LAUNCH
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("hgs_ls_kernel launch: %s", hipGetErrorString(e)); return DACO_E_HIP; }
  return DACO_OK;
}
