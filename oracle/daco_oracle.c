/*
 * daco_oracle.c -- CPU restatement (plain C, scalar, single thread) of DeepACO's ant-rollout
 * hot path.  TEST INFRASTRUCTURE ONLY: it is the checker for the HIP kernels.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product package
 * (deepaco_amd/) never does.
 *
 * Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py
 * against vectors captured by importing the reference in the build container
 * (tests/golden/gen_golden.py; fixtures: the .npz files under tests/golden).
 *
 * Each function cites the reference lines (henry-yeh/DeepACO @ /root/reference) it restates.
 * Arithmetic that must agree bit-for-bit with the HIP kernels (summation trees, Philox
 * counters, the -log2 polynomial) is restated here independently from the written
 * specification in DESIGN.md section 4, not shared as code with deepaco_amd/csrc.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORC_OK 0
#define ORC_INFEASIBLE 1 /* a row had no feasible candidate (reference: Categorical raises) */

/* ------------------------------------------------------------------ lane layout */
/* A row of n candidates is dealt to 64 lanes in vectors of VEC: candidate k of lane l is
 * k = (c*64 + l)*VEC + v, c = chunk, v = 0..VEC-1.  Row sums follow this layout. */
int orc_vec_for_n(int n) { return n > 128 ? 4 : (n > 64 ? 2 : 1); }
int orc_ld_for_n(int n) { int w = 64 * orc_vec_for_n(n); return (n + w - 1) / w * w; }

/* Inclusive scan over the 64 lanes in the order the hardware's DPP network offers
 * (DESIGN 4.2): Kogge-Stone inside each row of 16 lanes (d = 1,2,4,8), then rows 1 and 3
 * add the total of the row before them (lane 15 / 47), then lanes 32..63 add lane 31. */
static void lane_scan(float x[64]) {
  float t[64];
  for (int d = 1; d < 16; d <<= 1) {
    for (int l = 0; l < 64; ++l) t[l] = x[l] + ((l & 15) >= d ? x[l - d] : 0.0f);
    memcpy(x, t, sizeof t);
  }
  for (int l = 0; l < 64; ++l) t[l] = x[l] + (((l >> 4) & 1) ? x[(l & ~15) - 1] : 0.0f);
  memcpy(x, t, sizeof t);
  for (int l = 0; l < 64; ++l) t[l] = x[l] + (l >= 32 ? x[31] : 0.0f);
  memcpy(x, t, sizeof t);
}

/* row sum = lane-partial sums (c ascending, v ascending, from +0.0f), then lane 63 of the
 * inclusive lane scan. */
static float row_sum_tree(const float *p, int n) {
  int vec = orc_vec_for_n(n), ld = orc_ld_for_n(n), ch = ld / (64 * vec);
  float part[64];
  for (int l = 0; l < 64; ++l) {
    float s = 0.0f;
    for (int c = 0; c < ch; ++c)
      for (int v = 0; v < vec; ++v) {
        int k = (c * 64 + l) * vec + v;
        s = s + (k < n ? p[k] : 0.0f);
      }
    part[l] = s;
  }
  lane_scan(part);
  return part[63];
}

/* ------------------------------------------------------------------ Philox4x32-10 (Salmon et al. 2011) */
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  memcpy(out, ctr, 16);
  philox4x32_10(out, key[0], key[1]);
}

/* stream ids (counter word 3, top byte) */
enum { STREAM_START = 1, STREAM_RACE = 2, STREAM_SCAN = 3 };

static inline void rng_block(uint64_t seed, uint64_t iter, uint32_t stream, uint32_t ant_gid,
                             uint32_t idx, uint32_t out[4]) {
  out[0] = idx; out[1] = ant_gid; out[2] = (uint32_t)iter;
  out[3] = (stream << 24) | (uint32_t)((iter >> 32) & 0xFFFFFFu);
  philox4x32_10(out, (uint32_t)seed, (uint32_t)(seed >> 32));
}

/* uniform in (0,1), exactly representable: (2m+1) * 2^-24 with m the top 23 bits */
static inline float u01(uint32_t x) { return (float)(2u * (x >> 9) + 1u) * 0x1p-24f; }

/* -log2(1 - w), w = u01(x): exponential variate scaled by 1/ln2.  Polynomial with explicit
 * fmaf so CPU and GPU agree bit for bit.  Relative error < 1e-7. */
static inline float neg_log2_1m(float w) {
  float y = 1.0f - w;                         /* exact */
  uint32_t b; memcpy(&b, &y, 4);
  int e = (int)(b >> 23) - 127;
  b = (b & 0x007FFFFFu) | 0x3F800000u;
  float t; memcpy(&t, &b, 4);
  if (t > 1.41421354f) { t = t * 0.5f; e += 1; }
  float s = t - 1.0f;
  float q = 0x1.025a2p-3f;
  q = fmaf(q, s, -0x1.a8cc5cp-3f);
  q = fmaf(q, s, 0x1.b9b11ep-3f);
  q = fmaf(q, s, -0x1.e94f12p-3f);
  q = fmaf(q, s, 0x1.26d41p-2f);
  q = fmaf(q, s, -0x1.715c9cp-2f);
  q = fmaf(q, s, 0x1.ec73d4p-2f);
  q = fmaf(q, s, -0x1.71547p-1f);
  q = fmaf(q, s, 0x1.715476p+0f);
  return -fmaf(s, q, (float)e);
}
float orc_u01(uint32_t x) { return u01(x); }
float orc_neg_log2_1m(float w) { return neg_log2_1m(w); }

/* ------------------------------------------------------------------ T2: transition weights */
/* P[i][k] = tau[i][k]^alpha * eta[i][k]^beta   (tsp/aco.py:173, tsp_nls/aco.py:195).
 * x**1 is x, x**2 is x*x, x**0 is 1 (all exact, as in torch); other exponents use powf and
 * are outside the bit-exact claim. */
static inline float pw(float x, float a) {
  if (a == 1.0f) return x;
  if (a == 2.0f) return x * x;
  if (a == 0.0f) return 1.0f;
  return powf(x, a);
}
void orc_prob_matrix(int n, const float *tau, const float *eta, float alpha, float beta, float *P) {
  for (long i = 0; i < (long)n * n; ++i) P[i] = pw(tau[i], alpha) * pw(eta[i], beta);
}

#define EPS_F32 1.1920928955078125e-07f /* torch.finfo(float32).eps, clamp in probs_to_logits */

static inline float clamp_log(float pr) {
  if (pr < EPS_F32) pr = EPS_F32;
  if (pr > 1.0f - EPS_F32) pr = 1.0f - EPS_F32;
  return logf(pr);
}

/* ------------------------------------------------------------------ the three draws
 * Each takes the row of `prev` and a per-candidate `blocked` flag (visited / infeasible) and
 * returns the chosen candidate (or -1: no feasible candidate).  *pr receives the probability
 * of the choice as the reference's Categorical would report it (p_k / S, or the twice
 * normalised value for the tsp_nls variant). */

/* recorded noise: argmax_k ((p_k / S) [/ S']) / q_k, first maximum wins (torch.argmax);
 * tsp/aco.py:171-177 (norm_passes = 1), tsp_nls/aco.py:205-211 (norm_passes = 2) */
static int draw_noise(int n, const float *row, const unsigned char *blocked, const float *q,
                      int norm_passes, float *p, float *pr) {
  for (int k = 0; k < n; ++k) p[k] = blocked[k] ? 0.0f : row[k];   /* row * mask, mask in {0,1} */
  for (int pass = 0; pass < norm_passes; ++pass) {
    float S = row_sum_tree(p, n);
    for (int k = 0; k < n; ++k) p[k] = p[k] / S;
  }
  int best = -1; float bk = -INFINITY;
  for (int k = 0; k < n; ++k) {
    float key = p[k] / q[k];
    if (key > bk) { bk = key; best = k; }
  }
  if (best < 0 || !(bk > 0.0f)) return -1;
  if (pr) *pr = norm_passes ? p[best] : p[best] / row_sum_tree(p, n);
  return best;
}

/* in-kernel Philox race, division-free: argmin_k L_k * (1/row[k]), L_k = -log2(1-u_k),
 * u_k = component (k&3) of Philox(ctr = ((t<<12)|(k>>2), ant_gid, iter, STREAM_RACE)); ties ->
 * smallest k. */
static int draw_race(int n, const float *row, const unsigned char *blocked, uint64_t seed,
                     uint64_t iter, uint32_t gid, int t, float *p, float *pr) {
  uint32_t r4[4];
  int best = -1; float bk = INFINITY;
  for (int g = 0; g * 4 < n; ++g) {
    rng_block(seed, iter, STREAM_RACE, gid, ((uint32_t)t << 12) | (uint32_t)g, r4);
    for (int v = 0; v < 4; ++v) {
      int k = g * 4 + v;
      if (k >= n || blocked[k]) continue;
      float key = neg_log2_1m(u01(r4[v])) * (1.0f / row[k]);
      if (key < bk) { bk = key; best = k; }
    }
  }
  if (best < 0) return -1;
  if (pr) {
    for (int k = 0; k < n; ++k) p[k] = blocked[k] ? 0.0f : row[k];
    *pr = row[best] / row_sum_tree(p, n);
  }
  return best;
}

/* prefix-scan (roulette) draw -- the wave-shaped analogue of tsp_nls/aco.py:266-274.
 * `lanes` = 64 (one ant per wavefront), 32 (two ants per wavefront, 256 < n <= 512), 8 (eight ants per
 * wavefront, 128 < n <= 256) or 4 (sixteen, n <= 128); 16 (four, the layout before round 3) by knob; vec = 4 below 64 lanes:
 *   candidate k sits in lane (k/vec) % lanes, chunk k / (lanes*vec)
 *   part[l]  = lane-partial sum (c asc, v asc) of the unblocked p;  incl = lane scan of part
 *              (Kogge-Stone in rows of 16 -- lanes = 8: steps 1, 2, 4 only, lanes = 4: steps 1, 2; lanes = 32: lanes 16..31 then add lane 15;
 *               lanes = 64: rows 1, 3 add the row before, then lanes 32..63 add lane 31)
 *   S = incl[lanes-1];  r = max(u * S, denorm_min)
 *     u = component ((t>>lg)&3) of Philox(ctr=(((t>>(lg+2))<<lg) + (t&(lanes-1)), gid, iter, STREAM_SCAN)),
 *     lg = log2(lanes)
 *   L = first lane with incl[L] >= r and part[L] > 0
 *   inside lane L: thr = r - incl[L-1] (incl[-1] = 0);
 *     walk its candidates in (c,v) order with the lane's own running sum (from +0.0f, closed
 *       candidates add +0.0f); pick the first whose running sum >= thr, else the last open
 *       candidate with p > 0 of the lane (the several-ants-per-wave kernels find it by binary
 *       search over the running sums they keep in registers).  One rule for all five layouts. */
/* DACO_SCAN_LAYOUT (the library's measurement knob, daco_sample_kernel.h scan_small_lanes) is honoured here too, so that
 * the layouts it selects can be held against this restatement */
static int small_lanes(int n) {
  const char *e = getenv("DACO_SCAN_LAYOUT");
  int v = e ? atoi(e) : 0;
  if (v == 16) return 16;
  if (v == 8 && n <= 256) return 8;
  if (v == 4 && n <= 128) return 4;
  return n <= 128 ? 4 : 8;
}
int orc_scan_lanes(int n, int mode) { return mode != 2 ? 64 : (n <= 256 ? small_lanes(n) : (n <= 512 ? 16 : 64)); }
/* TSP: the two-ants-per-wavefront kernel serves n <= 1024 (its LDS tour / flag buffers hold 1024 entries) */
int orc_scan_lanes_tsp(int n, int mode) { return mode != 2 ? 64 : (n <= 256 ? small_lanes(n) : (n <= 1024 ? 32 : 64)); }

static int draw_scan(int n, const float *row, const unsigned char *blocked, uint64_t seed,
                     uint64_t iter, uint32_t gid, int t, float *pr, int lanes, const float *u_inj) {
  int vec = lanes < 64 ? 4 : orc_vec_for_n(n), w = lanes * vec, ch = (n + w - 1) / w;
  int lg = lanes == 64 ? 6 : (lanes == 32 ? 5 : (lanes == 16 ? 4 : (lanes == 8 ? 3 : 2)));
  float part[64], incl[64];
  uint32_t r4[4];
  for (int l = 0; l < 64; ++l) part[l] = incl[l] = 0.0f;
  for (int l = 0; l < lanes; ++l) {
    float s = 0.0f;
    for (int c = 0; c < ch; ++c)
      for (int v = 0; v < vec; ++v) {
        int k = (c * lanes + l) * vec + v;
        s = s + ((k < n && !blocked[k]) ? row[k] : 0.0f);
      }
    part[l] = s; incl[l] = s;
  }
  lane_scan(incl);                /* lanes 0..31 (0..15, 0..7, 0..3) of the 64-lane scan are exactly the half-wave (row, half-row, quad) scan */
  float S = incl[lanes - 1];
  uint32_t ut = (uint32_t)t;
  rng_block(seed, iter, STREAM_SCAN, gid, ((ut >> (lg + 2)) << lg) + (ut & (uint32_t)(lanes - 1)), r4);
  /* u_inj: the uniform comes from the caller instead of Philox (the reference's roulette with an injected
   * stream, tsp_nls/aco.py:266-274; how the GPU scan draw is tied to it, tests/test_gpu_00_tsp.py) */
  float r = (u_inj ? *u_inj : u01(r4[(ut >> lg) & 3u])) * S;
  if (!(r > 0.0f)) r = 1.401298464e-45f;               /* keep r > 0 if u*S underflows */
  int L = -1;
  for (int l = 0; l < lanes; ++l) if (incl[l] >= r && part[l] > 0.0f) { L = l; break; }
  if (L < 0) return -1;
  float thr = r - (L ? incl[L - 1] : 0.0f);
  float run = 0.0f;
  int best = -1, last = -1;
  for (int c = 0; c < ch && best < 0; ++c)
    for (int v = 0; v < vec; ++v) {
      int k = (c * lanes + L) * vec + v;
      if (k >= n || blocked[k] || !(row[k] > 0.0f)) continue;
      run = run + row[k];
      last = k;
      if (run >= thr) { best = k; break; }
    }
  if (best < 0) best = last;
  if (best >= 0 && pr) *pr = row[best] / S;
  return best;
}

enum { MODE_NOISE = 0, MODE_RACE = 1, MODE_SCAN = 2, MODE_SCAN_WAVE = 3 };

/* ------------------------------------------------------------------ T1/T2/T3: TSP tour construction
 * tsp/aco.py:134-177 and tsp_nls/aco.py:184-220.  start: given array (noise mode), else
 * fixed_start >= 0, else floor(n * u32 / 2^32) from STREAM_START.
 * noise: [n-1][A][n]; paths: [n][A] int64; logp: [n-1][A] or NULL. */
static int tsp_sample(int mode, int n, int A, const float *P, const int64_t *start,
                      const float *noise, int norm_passes, uint64_t seed, uint64_t iter,
                      uint32_t ant_gid0, int fixed_start, int64_t *paths, float *logp) {
  int rc = ORC_OK;
  float *p = (float *)malloc(sizeof(float) * n);
  unsigned char *vis = (unsigned char *)malloc(n);
  for (int a = 0; a < A; ++a) {
    uint32_t gid = ant_gid0 + (uint32_t)a, r4[4];
    int prev;
    if (start) prev = (int)start[a];
    else if (fixed_start >= 0) prev = fixed_start;
    else {
      rng_block(seed, iter, STREAM_START, gid, 0, r4);
      prev = (int)(((uint64_t)r4[0] * (uint64_t)n) >> 32);
    }
    memset(vis, 0, n);
    vis[prev] = 1;
    paths[a] = prev;
    for (int t = 1; t < n; ++t) {
      const float *row = P + (long)prev * n;
      float pr = 0.0f;
      int best;
      if (mode == MODE_NOISE) best = draw_noise(n, row, vis, noise + ((long)(t - 1) * A + a) * n, norm_passes, p, &pr);
      else if (mode == MODE_RACE) best = draw_race(n, row, vis, seed, iter, gid, t, p, logp ? &pr : NULL);
      else best = draw_scan(n, row, vis, seed, iter, gid, t, &pr, orc_scan_lanes_tsp(n, mode),
                            noise ? noise + ((long)(t - 1) * A + a) : NULL);     /* scan modes: noise = uniforms [n-1][A] */
      if (best < 0) { rc = ORC_INFEASIBLE; best = 0; pr = 0.0f; }
      if (logp) logp[(long)(t - 1) * A + a] = clamp_log(pr);
      vis[best] = 1;
      paths[(long)t * A + a] = best;
      prev = best;
    }
  }
  free(p); free(vis);
  return rc;
}
int orc_tsp_sample_noise(int n, int A, const float *P, const int64_t *start, const float *noise,
                         int norm_passes, int64_t *paths, float *logp) {
  return tsp_sample(MODE_NOISE, n, A, P, start, noise, norm_passes, 0, 0, 0, -1, paths, logp);
}
int orc_tsp_sample_race(int n, int A, const float *P, uint64_t seed, uint64_t iter,
                        uint32_t ant_gid0, int fixed_start, int64_t *paths, float *logp) {
  return tsp_sample(MODE_RACE, n, A, P, NULL, NULL, 0, seed, iter, ant_gid0, fixed_start, paths, logp);
}
/* scan draw with the uniforms supplied by the caller: u [n-1][A] f32 in (0,1) */
int orc_tsp_sample_scan_injected(int n, int A, const float *P, const float *u, int wave, int fixed_start,
                                 const int64_t *start, int64_t *paths, float *logp) {
  return tsp_sample(wave ? MODE_SCAN_WAVE : MODE_SCAN, n, A, P, start, u, 0, 0, 0, 0, fixed_start, paths, logp);
}
int orc_tsp_sample_scan(int n, int A, const float *P, uint64_t seed, uint64_t iter,
                        uint32_t ant_gid0, int fixed_start, int64_t *paths, float *logp) {
  return tsp_sample(MODE_SCAN, n, A, P, NULL, NULL, 0, seed, iter, ant_gid0, fixed_start, paths, logp);
}
/* DACO_SCAN_WAVE: the one-ant-per-wavefront layout for every n */
int orc_tsp_sample_scan_wave(int n, int A, const float *P, uint64_t seed, uint64_t iter,
                             uint32_t ant_gid0, int fixed_start, int64_t *paths, float *logp) {
  return tsp_sample(MODE_SCAN_WAVE, n, A, P, NULL, NULL, 0, seed, iter, ant_gid0, fixed_start, paths, logp);
}

/* ------------------------------------------------------------------ scan_sparse: the roulette draw on head / tail rows
 * (tsp/aco.py:52-67 sparsify + tsp_nls/aco.py:260-275 roulette).
 * The reference's inference heuristic is k-sparse: k live entries per row, 1e-10 elsewhere.  A row is split into a HEAD --
 * up to 63 candidates given by the caller (ids, any subset; the colony takes the k largest heuristic entries) -- and the
 * TAIL, everything else.  A step draws r = u * (H + T): H = the head's LIVE mass (its open candidates), T = the tail's
 * STATIC mass (all tail entries, visited or not).  r inside the head -> inverse CDF over the open head candidates.  r past
 * the head -> inverse CDF over ALL tail entries, and if the one it lands on is visited the step draws ONCE more: the dense
 * masked draw over all open candidates with a second uniform.  Rejection over a superset with an exact fallback: the outcome
 * j has probability P_ij / sum(open P) exactly as in the reference's categorical (up to float rounding, like every draw
 * here) and a step never reads the row more than twice.  No live head candidate (H = 0) -> the dense
 * masked draw of the 64-lane scan specification with the same uniform.  So most steps read 384 bytes instead of a row.
 *   head slot m = 4 * lane + v, 16 lanes; w_m = val_m if m < cnt and its node is open, else +0; lane partial = its four
 *   slots in order from +0.0f; incl = Kogge-Stone over the 16 lanes; H = incl[15]; T = head_val[63] (so cnt <= 63), made
 *   by orc_sparse_head_values as the 64-lane (vec 4) scan total of the row's non-head entries.
 *   u(t, attempt 0) = component (t>>4)&3 of Philox(ctr = ((t>>6)<<4) + (t&15), gid, iter, STREAM_SPARSE)
 *   u(t, second draw) = component 0 of Philox(ctr = t<<8, gid, iter, STREAM_SPARSE_RETRY)
 *   in-lane picks as in draw_scan (first running sum >= thr over the positive terms, else the last positive one).
 * stats: [0] dense masked draws (H = 0, or after a rejection), [1] tail walks, [2] rejections. */
enum { STREAM_SPARSE = 4, STREAM_SPARSE_RETRY = 5 };
/* kh = head slots per row, 64 or 128 (4 or 8 per lane, 16 lanes): slot m = (kh / 16) * lane + v, cnt <= kh - 1 live slots,
 * slot kh - 1 = the tail total.  Everything above holds with "four" read as kh / 16 and "63" as kh - 1. */
#define SPARSE_KH_MAX 128

static void sparse_tail_scan(int n, const float *row, const unsigned char *is_head, float part[64], float incl[64]) {
  int ch = (n + 255) / 256;
  for (int l = 0; l < 64; ++l) {
    float s = 0.0f;
    for (int c = 0; c < ch; ++c)
      for (int v = 0; v < 4; ++v) {
        int k = (c * 64 + l) * 4 + v;
        s = s + ((k < n && !is_head[k]) ? row[k] : 0.0f);
      }
    part[l] = s; incl[l] = s;
  }
  lane_scan(incl);
}

/* head_val [n][kh] from the dense matrix P [n][n], the head ids [n][kh] (uint16) and counts [n] (<= kh - 1): slot m < cnt holds
 * P[i][id_m], the other slots +0, slot kh - 1 the tail total T_i */
void orc_sparse_head_values(int n, const float *P, const uint16_t *head_id, const uint8_t *head_cnt, float *head_val, int kh) {
  unsigned char *is_head = (unsigned char *)malloc(n);
  float part[64], incl[64];
  for (int i = 0; i < n; ++i) {
    memset(is_head, 0, n);
    for (int m = 0; m < kh; ++m) {
      int live = m < head_cnt[i];
      head_val[(long)i * kh + m] = live ? P[(long)i * n + head_id[(long)i * kh + m]] : 0.0f;
      if (live) is_head[head_id[(long)i * kh + m]] = 1;
    }
    sparse_tail_scan(n, P + (long)i * n, is_head, part, incl);
    head_val[(long)i * kh + kh - 1] = incl[63];
  }
  free(is_head);
}

/* how often a head draw took the rounding branch (no running sum reached the remainder): a soak reads it to know that its
 * inputs exercise that branch (tools/soak_scan_sparse.py) */
static long sparse_rounding_picks = 0;
long orc_sparse_rounding_picks(void) { return sparse_rounding_picks; }

static int draw_scan_sparse(int n, const float *row, const float *hval, const uint16_t *hid, int cnt, int kh,
                            const unsigned char *blocked, unsigned char *is_head, uint64_t seed, uint64_t iter,
                            uint32_t gid, int t, long stats[3]) {
  float part[64], incl[64], w[SPARSE_KH_MAX];
  uint32_t r4[4];
  const int spl = kh / 16;
  for (int l = 0; l < 64; ++l) part[l] = incl[l] = 0.0f;
  for (int l = 0; l < 16; ++l) {
    float s = 0.0f;
    for (int v = 0; v < spl; ++v) {
      int m = spl * l + v;
      w[m] = (m < cnt && !blocked[hid[m]]) ? hval[m] : 0.0f;
      s = s + w[m];
    }
    part[l] = s; incl[l] = s;
  }
  lane_scan(incl);                                       /* lanes 0..15 of the 64-lane scan = the 16-lane row scan */
  const float H = incl[15], T = hval[kh - 1];
  const uint32_t ut = (uint32_t)t;
  float u;
  rng_block(seed, iter, STREAM_SPARSE, gid, ((ut >> 6) << 4) + (ut & 15u), r4);
  u = u01(r4[(ut >> 4) & 3u]);
  if (!(H > 0.0f)) {                                     /* no live head candidate: the dense masked draw */
    if (stats) stats[0]++;
    return draw_scan(n, row, blocked, 0, 0, 0, t, NULL, 64, &u);
  }
  float r = u * (H + T);
  if (!(r > 0.0f)) r = 1.401298464e-45f;
  int L = -1;
  for (int l = 0; l < 16; ++l) if (incl[l] >= r && part[l] > 0.0f) { L = l; break; }
  if (L >= 0) {
    float thr = r - (L ? incl[L - 1] : 0.0f), run = 0.0f;
    int best = -1, last = -1;
    for (int v = 0; v < spl; ++v) {
      int m = spl * L + v;
      if (!(w[m] > 0.0f)) continue;
      run = run + w[m];
      last = m;
      if (run >= thr) { best = m; break; }
    }
    if (best < 0) { best = last; sparse_rounding_picks++; }
    return hid[best];
  }
  /* past the head: walk the whole tail, visited entries included */
  if (stats) stats[1]++;
  float rp = r - H;
  if (!(rp > 0.0f)) rp = 1.401298464e-45f;
  memset(is_head, 0, n);
  for (int m = 0; m < cnt; ++m) is_head[hid[m]] = 1;
  float tpart[64], tincl[64];
  sparse_tail_scan(n, row, is_head, tpart, tincl);
  int Lt = -1;
  for (int l = 0; l < 64; ++l) if (tincl[l] >= rp && tpart[l] > 0.0f) { Lt = l; break; }
  if (Lt < 0) for (int l = 63; l >= 0; --l) if (tpart[l] > 0.0f) { Lt = l; break; }      /* rounding: the last lane with mass */
  int j = -1;
  if (Lt >= 0) {
    float thr = rp - (Lt ? tincl[Lt - 1] : 0.0f), run = 0.0f;
    int last = -1, ch = (n + 255) / 256;
    for (int c = 0; c < ch && j < 0; ++c)
      for (int v = 0; v < 4; ++v) {
        int k = (c * 64 + Lt) * 4 + v;
        if (k >= n || is_head[k] || !(row[k] > 0.0f)) continue;
        run = run + row[k];
        last = k;
        if (run >= thr) { j = k; break; }
      }
    if (j < 0) j = last;
  }
  if (j < 0) {                                           /* a tail without mass: the head's last live candidate */
    for (int m = cnt - 1; m >= 0; --m) if (w[m] > 0.0f) return hid[m];
    return -1;
  }
  if (!blocked[j]) return j;
  /* the tail entry is visited: the dense masked draw over ALL open candidates with the step's second uniform.  The outcome x
   * then has probability p_x / (H + T) + (T_visited / (H + T)) * p_x / (H + T_open) = p_x / (H + T_open): the categorical. */
  if (stats) { stats[2]++; stats[0]++; }
  rng_block(seed, iter, STREAM_SPARSE_RETRY, gid, ut << 8, r4);
  u = u01(r4[0]);
  return draw_scan(n, row, blocked, 0, 0, 0, t, NULL, 64, &u);
}

/* P [n][n] dense, head_id [n][kh] uint16, head_cnt [n] uint8 (<= kh - 1), head_val [n][kh] from orc_sparse_head_values */
int orc_tsp_sample_scan_sparse(int n, int A, const float *P, const uint16_t *head_id, const uint8_t *head_cnt,
                               const float *head_val, int kh, uint64_t seed, uint64_t iter, uint32_t ant_gid0, int fixed_start,
                               int64_t *paths, long *stats) {
  int rc = ORC_OK;
  if (kh != 64 && kh != 128) return ORC_INFEASIBLE;
  unsigned char *vis = (unsigned char *)malloc(n), *is_head = (unsigned char *)malloc(n);
  for (int a = 0; a < A; ++a) {
    uint32_t gid = ant_gid0 + (uint32_t)a, r4[4];
    int prev;
    if (fixed_start >= 0) prev = fixed_start;
    else { rng_block(seed, iter, STREAM_START, gid, 0, r4); prev = (int)(((uint64_t)r4[0] * (uint64_t)n) >> 32); }
    memset(vis, 0, n);
    vis[prev] = 1;
    paths[a] = prev;
    for (int t = 1; t < n; ++t) {
      int best = draw_scan_sparse(n, P + (long)prev * n, head_val + (long)prev * kh, head_id + (long)prev * kh,
                                  head_cnt[prev], kh, vis, is_head, seed, iter, gid, t, stats);
      if (best < 0) { rc = ORC_INFEASIBLE; best = 0; }
      vis[best] = 1;
      paths[(long)t * A + a] = best;
      prev = best;
    }
  }
  free(vis); free(is_head);
  return rc;
}

/* ------------------------------------------------------------------ T4 / C5: tour costs
 * closed: sum_k dist[u_k][u_{k-1 mod n}]  (tsp/aco.py:121-132)
 * open:   sum_{k<len-1} dist[u_k][u_{k+1}] (cvrp/aco.py:133-136)
 * Order: sequential from +0.0f, closed tours add the closing edge last (the reference's
 * torch.sum order is unspecified; agreement
 * with it is to 1e-5 relative, agreement with the HIP kernel is bitwise). */
void orc_tour_costs(int n, int len, int A, const float *dist, const int64_t *paths, int closed,
                    float *costs) {
  for (int a = 0; a < A; ++a) {
    float s = 0.0f;
    if (closed) {      /* edges k = 1..len-1 in order, then the closing edge d[u_0][u_{len-1}] */
      for (int k = 1; k < len; ++k) {
        long u = paths[(long)k * A + a], v = paths[(long)(k - 1) * A + a];
        s = s + dist[u * n + v];
      }
      s = s + dist[paths[a] * n + paths[(long)(len - 1) * A + a]];
    } else {
      for (int k = 0; k + 1 < len; ++k) {
        long u = paths[(long)k * A + a], v = paths[(long)(k + 1) * A + a];
        s = s + dist[u * n + v];
      }
    }
    costs[a] = s;
  }
}

/* ------------------------------------------------------------------ U1: pheromone update, TSP
 * tsp/aco.py:95-118.  tau <- tau*decay; for each ant in index order (or only the best ant if
 * elitist; torch.min returns the first minimum): w = 1/cost;
 * tau[path, roll(path,1)] += w, then tau[roll(path,1), path] += w (non-accumulating
 * index_put: gather old values, add, scatter).  MMAS clamp if clamp_max > 0. */
static void deposit_tsp(int n, int A, float *tau, const int64_t *paths, int a, float w, float *buf) {
  for (int pass = 0; pass < 2; ++pass) {
    for (int k = 0; k < n; ++k) {
      long u = paths[(long)k * A + a], v = paths[(long)((k + n - 1) % n) * A + a];
      buf[k] = (pass == 0 ? tau[u * n + v] : tau[v * n + u]) + w;
    }
    for (int k = 0; k < n; ++k) {
      long u = paths[(long)k * A + a], v = paths[(long)((k + n - 1) % n) * A + a];
      if (pass == 0) tau[u * n + v] = buf[k]; else tau[v * n + u] = buf[k];
    }
  }
}
void orc_pheromone_update_tsp(int n, int A, float *tau, const int64_t *paths, const float *costs,
                              float decay, int elitist, float clamp_min, float clamp_max) {
  float *buf = (float *)malloc(sizeof(float) * n);
  for (long i = 0; i < (long)n * n; ++i) tau[i] = tau[i] * decay;
  if (elitist) {
    int b = 0;
    for (int a = 1; a < A; ++a) if (costs[a] < costs[b]) b = a;
    deposit_tsp(n, A, tau, paths, b, 1.0f / costs[b], buf);
  } else {
    for (int a = 0; a < A; ++a) deposit_tsp(n, A, tau, paths, a, 1.0f / costs[a], buf);
  }
  if (clamp_max > 0.0f) {
    for (long i = 0; i < (long)n * n; ++i) {
      if (tau[i] < clamp_min) tau[i] = clamp_min;   /* ((tau>1e-9)*tau < min) == (tau < min) */
      if (tau[i] > clamp_max) tau[i] = clamp_max;
    }
  }
  free(buf);
}

/* ------------------------------------------------------------------ C5: pheromone update, CVRP
 * cvrp/aco.py:107-130.  Directed: tau[path[:-1], path[1:]] += w, duplicates of an index pair
 * (the padding edge (0,0)) collapse to one add; floor tau < 1e-10 -> 1e-10. */
void orc_pheromone_update_directed(int n, int len, int A, float *tau, const int64_t *paths,
                                   const float *costs, const float *weights, float decay, int elitist,
                                   float clamp_min, float clamp_max, float floor_val);
void orc_pheromone_update_cvrp(int n, int len, int A, float *tau, const int64_t *paths,
                               const float *costs, float decay, int elitist, float clamp_min,
                               float clamp_max) {
  orc_pheromone_update_directed(n, len, A, tau, paths, costs, NULL, decay, elitist, clamp_min, clamp_max, 1e-10f);
}
/* The same directed deposit with an explicit amount per ant (weights != NULL) for the siblings
 * whose rule is not 1/cost: op/aco.py:128-143, pctsp/aco.py:87-102, sop/aco.py:77-98,
 * smtwtp/aco.py:77-97, bpp/aco.py:100-118, mkp/aco.py:88-103.  `costs` picks the elitist ant
 * (first minimum); floor_val > 0 applies tau < floor -> floor at the end. */
void orc_pheromone_update_directed(int n, int len, int A, float *tau, const int64_t *paths,
                                   const float *costs, const float *weights, float decay, int elitist,
                                   float clamp_min, float clamp_max, float floor_val) {
  float *buf = (float *)malloc(sizeof(float) * (len > 0 ? len : 1));
  for (long i = 0; i < (long)n * n; ++i) tau[i] = tau[i] * decay;
  int lo = 0, hi = A;
  if (elitist) {
    int b = 0;
    for (int a = 1; a < A; ++a) if (costs[a] < costs[b]) b = a;
    lo = b; hi = b + 1;
  }
  for (int a = lo; a < hi; ++a) {
    float w = weights ? weights[a] : 1.0f / costs[a];
    for (int k = 0; k + 1 < len; ++k)
      buf[k] = tau[paths[(long)k * A + a] * n + paths[(long)(k + 1) * A + a]] + w;
    for (int k = 0; k + 1 < len; ++k)
      tau[paths[(long)k * A + a] * n + paths[(long)(k + 1) * A + a]] = buf[k];
  }
  if (clamp_max > 0.0f)
    for (long i = 0; i < (long)n * n; ++i) {
      if (tau[i] < clamp_min) tau[i] = clamp_min;
      if (tau[i] > clamp_max) tau[i] = clamp_max;
    }
  if (floor_val > 0.0f)
    for (long i = 0; i < (long)n * n; ++i) if (tau[i] < floor_val) tau[i] = floor_val;
  free(buf);
}

/* ------------------------------------------------------------------ T2 as a service: one draw per ant
 * ACO.pick_move (tsp/aco.py:165-177 and its copies in the sibling problems) on a caller-supplied
 * mask [A][n] (0 = closed).  prev [A]; noise [A][n] for the recorded-noise mode; `step` keys the
 * Philox counters exactly as the step index t does in the fused samplers. */
int orc_pick_move(int mode, int n, int A, const float *P, const int64_t *prev, const float *mask,
                  const float *noise, uint64_t seed, uint64_t iter, uint32_t ant_gid0, int step,
                  int64_t *actions, float *logp) {
  int rc = ORC_OK;
  float *p = (float *)malloc(sizeof(float) * n);
  unsigned char *blocked = (unsigned char *)malloc(n);
  for (int a = 0; a < A; ++a) {
    const float *row = P + (long)prev[a] * n;
    for (int k = 0; k < n; ++k) blocked[k] = mask[(long)a * n + k] == 0.0f;
    float pr = 0.0f;
    int best;
    if (mode == MODE_NOISE) best = draw_noise(n, row, blocked, noise + (long)a * n, 1, p, &pr);
    else if (mode == MODE_RACE) best = draw_race(n, row, blocked, seed, iter, ant_gid0 + (uint32_t)a, step, p, logp ? &pr : NULL);
    else best = draw_scan(n, row, blocked, seed, iter, ant_gid0 + (uint32_t)a, step, &pr, 64, NULL);
    if (best < 0) { rc = ORC_INFEASIBLE; best = 0; pr = 0.0f; }
    actions[a] = best;
    if (logp) logp[a] = clamp_log(pr);
  }
  free(p); free(blocked);
  return rc;
}

/* ------------------------------------------------------------------ O1/O2: 2-opt
 * tsp_nls/two_opt.py:6-39.  Best-improvement sweep over 1 <= i < j <= n-1 with
 * change = d[t[i-1]][t[j]] + d[t[i]][t[(j+1)%n]] - d[t[i-1]][t[i]] - d[t[j]][t[(j+1)%n]]
 * evaluated left to right in f32; strict '<' keeps the first minimum in row-major (i,j)
 * order; reverse t[i..j] if delta < -1e-6.  Returns delta (0 if no move). */
float orc_two_opt_once(int n, const float *d, uint16_t *t) {
  int p = 0, q = 0;
  float delta = 0.0f;
  for (int i = 1; i < n - 1; ++i)
    for (int j = i + 1; j < n; ++j) {
      int ni = t[i], nj = t[j], np_ = t[i - 1], nn = t[(j + 1) % n];
      if (np_ == nj || nn == ni) continue;
      float change = d[(long)np_ * n + nj] + d[(long)ni * n + nn];
      change = change - d[(long)np_ * n + ni];
      change = change - d[(long)nj * n + nn];
      if (change < delta) { p = i; q = j; delta = change; }
    }
  if ((double)delta < -1e-6) {           /* numba compares the f32 delta with the f64 literal */
    for (int i = p, j = q; i < j; ++i, --j) { uint16_t x = t[i]; t[i] = t[j]; t[j] = x; }
    return delta;
  }
  return 0.0f;
}
/* returns the number of sweeps performed */
int orc_two_opt(int n, const float *d, uint16_t *t, long max_iterations) {
  long it = 0;
  float mc = -1.0f;
  while ((double)mc < -1e-6 && it < max_iterations) {
    mc = orc_two_opt_once(n, d, t);
    ++it;
  }
  return (int)it;
}
void orc_two_opt_batch(int n, int T, const float *d, uint16_t *tours, long max_iterations,
                       int32_t *sweeps) {
  for (int r = 0; r < T; ++r) {
    int s = orc_two_opt(n, d, tours + (long)r * n, max_iterations);
    if (sweeps) sweeps[r] = s;
  }
}

/* ------------------------------------------------------------------ I1: roulette sampler, literal restatement
 * tsp_nls/aco.py:260-275 with an injected uniform stream (the reference's RNG is numba's
 * private generator and cannot be seeded): rand = U * sum_f32(prob*mask) (numpy's pairwise
 * float32 sum), then rand -= prob[k] in float64 (numba types U as float64) until rand <= 0. */
static float np_pairwise_sum_f32(const float *a, int n) {
  /* numpy's FLOAT_pairwise_sum (contiguous float32 .sum()) */
  if (n < 8) { float s = 0.0f; for (int i = 0; i < n; ++i) s += a[i]; return s; }
  if (n <= 128) {
    float r[8], sum; int i;
    for (i = 0; i < 8; ++i) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
      for (int u = 0; u < 8; ++u) r[u] += a[i + u];
    sum = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) sum += a[i];
    return sum;
  }
  int n2 = n / 2; n2 -= n2 % 8;
  return np_pairwise_sum_f32(a, n2) + np_pairwise_sum_f32(a + n2, n - n2);
}
void orc_roulette_route(int n, const float *probmat, const double *uniforms, int start,
                        uint16_t *route) {
  unsigned char *mask = (unsigned char *)malloc(n);
  float *prob = (float *)malloc(sizeof(float) * n);
  memset(mask, 1, n);
  int last = start;
  route[0] = (uint16_t)start;
  for (int j = 1; j < n; ++j) {
    mask[last] = 0;
    for (int k = 0; k < n; ++k) prob[k] = probmat[(long)last * n + k] * (float)mask[k];
    float sum = np_pairwise_sum_f32(prob, n);
    double rnd = uniforms[j - 1] * (double)sum;
    int k;
    for (k = 0; k < n; ++k) { rnd -= (double)prob[k]; if (rnd <= 0) break; }
    if (k == n) k = n - 1;
    last = k;
    route[j] = (uint16_t)k;
  }
  free(mask); free(prob);
}

/* ------------------------------------------------------------------ C1-C4: CVRP tour construction
 * cvrp/aco.py:138-205.  Node 0 is the depot; n1 = customers + 1.  Every ant starts at the
 * depot; each step: p = P[prev] * visit_mask * capacity_mask, Categorical -> one of the draws
 * above (noise mode: norm_passes = 1).  visit mask: visited customers 0; depot 1 unless the
 * ant is at the depot and customers remain (C2).  capacity: used = 0 at depot;
 * used += demand[cur]; candidates with demand > capacity - used masked (strict, C3).  An ant
 * is done when it is at the depot with every customer visited (C4); the reference keeps
 * stepping until all ants are done, so a done ant keeps choosing the depot (its only
 * candidate, probability 1) and its column is padded with 0.
 * noise: [noise_steps][A][n1]; paths: [Lmax][A] zero-initialised by the caller.
 * Returns L (number of rows used, max over ants) or -1 on infeasible/overflow. */
static int cvrp_sample_impl(int mode, int n1, int A, const float *P, const float *demand, float capacity,
                            const double *demand64, double capacity64,
                            const float *noise, int noise_steps, uint64_t seed, uint64_t iter,
                            uint32_t ant_gid0, int Lmax, int64_t *paths, float *logp) {
  float *p = (float *)malloc(sizeof(float) * n1);
  unsigned char *vis = (unsigned char *)malloc(n1);
  unsigned char *blocked = (unsigned char *)malloc(n1);
  int *lens = (int *)malloc(sizeof(int) * A);
  int L = 1, fail = 0;
  for (int a = 0; a < A && !fail; ++a) {
    uint32_t gid = ant_gid0 + (uint32_t)a;
    memset(vis, 0, n1);
    int prev = 0, remaining = n1 - 1, len = 1;
    float used = 0.0f;
    used = used + demand[0];
    double used64 = 0.0;                 /* cvrp_nls keeps its data in float64: the load bookkeeping then runs in double */
    if (demand64) used64 = used64 + demand64[0];
    paths[a] = 0;
    while (!(remaining == 0 && prev == 0)) {
      if (len >= Lmax || (mode == MODE_NOISE && len - 1 >= noise_steps)) { fail = 1; break; }
      const float *row = P + (long)prev * n1;
      float rem = capacity - used;
      double rem64 = capacity64 - used64;
      for (int k = 0; k < n1; ++k) {
        int visit_ok = (k == 0) ? !(prev == 0 && remaining > 0) : !vis[k];
        int cap_ok = demand64 ? !(demand64[k] > rem64) : !(demand[k] > rem);      /* cvrp_nls/aco.py:267-270 in double */
        blocked[k] = !(visit_ok && cap_ok);
      }
      float pr = 0.0f;
      int best;
      if (mode == MODE_NOISE) best = draw_noise(n1, row, blocked, noise + ((long)(len - 1) * A + a) * n1, 1, p, &pr);
      else if (mode == MODE_RACE) best = draw_race(n1, row, blocked, seed, iter, gid, len, p, logp ? &pr : NULL);
      else best = draw_scan(n1, row, blocked, seed, iter, gid, len, &pr, orc_scan_lanes(n1, mode), NULL);
      if (best < 0) { fail = 1; break; }
      if (logp) logp[(long)(len - 1) * A + a] = clamp_log(pr);
      if (best != 0) { vis[best] = 1; --remaining; }
      if (best == 0) { used = 0.0f; used64 = 0.0; }
      used = used + demand[best];
      if (demand64) used64 = used64 + demand64[best];
      paths[(long)len * A + a] = best;
      prev = best;
      ++len;
    }
    lens[a] = len;
    if (len > L) L = len;
  }
  /* a done ant keeps drawing the depot with probability 1 -> log(clamp(1)) = log(1-eps) */
  if (logp && !fail)
    for (int a = 0; a < A; ++a)
      for (int k = lens[a]; k < L; ++k) logp[(long)(k - 1) * A + a] = clamp_log(1.0f);
  free(p); free(vis); free(blocked); free(lens);
  return fail ? -1 : L;
}
int orc_cvrp_sample(int mode, int n1, int A, const float *P, const float *demand, float capacity,
                    const float *noise, int noise_steps, uint64_t seed, uint64_t iter,
                    uint32_t ant_gid0, int Lmax, int64_t *paths, float *logp) {
  return cvrp_sample_impl(mode, n1, A, P, demand, capacity, NULL, 0.0, noise, noise_steps, seed, iter, ant_gid0, Lmax, paths, logp);
}
/* float64 demands and capacity (cvrp_nls/): the capacity mask is decided in double; demand = their float32 image (unused by the mask) */
int orc_cvrp_sample64(int mode, int n1, int A, const float *P, const float *demand, const double *demand64, double capacity64,
                      const float *noise, int noise_steps, uint64_t seed, uint64_t iter,
                      uint32_t ant_gid0, int Lmax, int64_t *paths, float *logp) {
  return cvrp_sample_impl(mode, n1, A, P, demand, (float)capacity64, demand64, capacity64, noise, noise_steps, seed, iter, ant_gid0,
                          Lmax, paths, logp);
}
int orc_cvrp_sample_noise(int n1, int A, const float *P, const float *demand, float capacity,
                          const float *noise, int noise_steps, int Lmax, int64_t *paths,
                          float *logp) {
  return orc_cvrp_sample(MODE_NOISE, n1, A, P, demand, capacity, noise, noise_steps, 0, 0, 0, Lmax, paths, logp);
}
