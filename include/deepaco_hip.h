/*
 * deepaco_hip.h -- C ABI of libdeepaco_hip.so: DeepACO's ant-rollout hot path on MI355X (gfx950).
 *
 * This is the drop-in boundary.  The reference (henry-yeh/DeepACO) has no native boundary for
 * this path: its ACO classes issue stock PyTorch ops.  Each entry point below replaces the
 * body of one reference method and is bound from Python with ctypes and tensor.data_ptr(),
 * the same mechanism the reference already uses for its one native dependency
 * (cvrp_nls/swapstar.py:134-185 loads libhgscvrp.so with ctypes).  INTEGRATION.md shows the
 * binding a maintainer would add on the reference side.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless stated; arrays are dense, row-major;
 *  - B = instances in the batch (the reference always runs B = 1), n = nodes, A = ants;
 *  - `stream` is a hipStream_t (void* here so the header needs no HIP include); all work is
 *    enqueued on it and no entry point synchronises with the host;
 *  - return value: 0 = ok, < 0 = error (see DACO_E_*); the message is in daco_last_error()
 *    (thread-local).  No entry point ever falls back to a CPU path.
 *  - the library keeps no global mutable state: entry points are re-entrant across streams
 *    and devices; scratch memory is the caller's (size from the *_workspace_bytes queries).
 */
#ifndef DEEPACO_HIP_H
#define DEEPACO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DACO_VERSION 126 /* 0.1.21: bumped whenever an entry point's signature or the draw stream of a mode changes (120: daco_tsp_sample_sparse; 121: head_slots; 122: scan_sparse draws once after a rejection; 123: its workspace takes the ant count; 124: daco_hgs_*, daco_cvrp_sample takes ant_gid_bstride; 125: daco_tsp_sample_heads / daco_pheromone_update_heads, the sparse workspace no longer holds dense rows; 126: daco_tsp_sparse_tours_offset / daco_track_best_tours16, the sparse workspace holds the u16 tours at every n) */

/* error codes */
#define DACO_OK 0
#define DACO_E_BADARG (-1)   /* null pointer, size out of range */
#define DACO_E_TOOLARGE (-2) /* n exceeds what the kernel's register/LDS plan supports */
#define DACO_E_HIP (-3)      /* a HIP runtime call failed */
#define DACO_E_WORKSPACE (-4)/* workspace too small */

/* sampler modes */
#define DACO_RACE_NOISE 0  /* exponential race, noise read from memory (bit-exact parity mode) */
#define DACO_RACE_PHILOX 1 /* exponential race, Philox4x32-10 noise generated in-kernel */
#define DACO_SCAN 2        /* roulette / inverse-CDF by wavefront prefix scan, one uniform per step.
                            * daco_tsp_sample and daco_cvrp_sample pack four ants per wavefront for
                            * n <= 256 and two for 256 < n <= 512 (the 16- and 32-lane variants of the
                            * scan); everything else uses one ant per wavefront. */
#define DACO_SCAN_WAVE 3   /* DACO_SCAN with the one-ant-per-wavefront layout for every n: the draw
                            * daco_pick_move / daco_sibling_sample make (there it is a synonym of
                            * DACO_SCAN) */

/* limits */
#define DACO_MAX_NODES 4096

int daco_version(void);
const char *daco_last_error(void);

/* Layout helpers (the padded leading dimension the sampler uses internally; exported so the
 * oracle-side tests can state the summation order). */
int daco_vec_for_n(int n);
int daco_ld_for_n(int n);

/* ---------------------------------------------------------------------------------------------
 * daco_tsp_sample -- replaces ACO.gen_path + ACO.pick_move
 *   tsp/aco.py:134-177 (random start, Categorical normalises once -> norm_passes = 1)
 *   tsp_nls/aco.py:184-220 (start 0, explicit renormalisation  -> norm_passes = 2)
 *
 * For every instance b and ant a builds a tour of n nodes: start node, then n-1 draws from
 * p_k = tau[prev][k]^alpha * eta[prev][k]^beta * [k unvisited].
 *
 *   tau, eta   [B][n][n] f32.  tau_bstride / eta_bstride: element stride between instances
 *              (n*n for dense batches, 0 to share one matrix across the batch).
 *   mode       DACO_RACE_NOISE: action = argmax_k ((p_k/S)/q_k) with q = noise, exactly the
 *              arithmetic of torch.multinomial's one-sample path (first maximum wins);
 *              norm_passes in {0,1,2} = how many times p is divided by its row sum first.
 *              DACO_RACE_PHILOX: the same race with q_k = -log2(1-u_k), u from Philox.
 *              DACO_SCAN: r = u*S, first candidate (lane-major order) whose running sum >= r.
 *   start      [B][A] int64 or NULL.  If NULL: fixed_start >= 0 -> every ant starts there,
 *              fixed_start < 0 -> start = floor(n * u32 / 2^32) from the Philox stream.
 *   noise      DACO_RACE_NOISE: [B][n-1][A][n] f32 (the reference's q tensors, step-major), required.
 *              DACO_SCAN / DACO_SCAN_WAVE: NULL, or [B][n-1][A] f32 uniforms in (0,1) that replace the Philox
 *              stream (the reference's roulette with an injected uniform stream, tsp_nls/aco.py:266-274).
 *   seed, iter, ant_gid0   Philox key and counter words: ant (b,a) uses global ant id
 *              ant_gid0 + b*A + a; `iter` must differ between calls that should be independent.
 *   ant_gid_bstride  0, or the ant-id stride between instances when this call builds only a slice of a
 *              colony's ants: ant (b,a) then uses id ant_gid0 + b*ant_gid_bstride + a.  A rank that owns ants
 *              [lo, hi) of colonies with A_total ants passes ant_gid0 = lo, ant_gid_bstride = A_total and draws
 *              exactly the tours a single call with all A_total ants would draw for those ants.
 *   iter_offset  optional device pointer to a uint64 that the kernel adds to `iter` when it starts:
 *              a caller that captures its iteration into a HIP graph keeps the counter in device
 *              memory and bumps it inside the graph (arguments are frozen at capture).  NULL = 0.
 *   paths      out [B][n][A] int64  (reference layout: paths[:, i] is ant i's tour)
 *   logp       out [B][n-1][A] f32 or NULL: log(clamp(p_chosen/S, eps, 1-eps)), eps = 2^-23
 *   rowsum     out [B][n-1][A] f32 or NULL: S at each step (saved for daco_tsp_sample_backward)
 *   flags      out [B] int32 or NULL: set to 1 if some draw had no feasible candidate
 *              (the reference's Categorical raises ValueError there); caller zeroes it.
 *   dist, costs   optional fusion of ACO.gen_path_costs: if costs != NULL, dist [B][n][n]
 *              (dist_bstride as for tau) is read once per step and costs [B][A] receives the
 *              closed-tour length in daco_tour_costs' summation order.
 *   nbr        optional out [B][n][A] uint32: prev(node) | next(node) << 16 per (node, ant), the
 *              form daco_pheromone_update consumes (saves its own pass over `paths`).
 *   workspace  daco_tsp_sample_workspace_bytes(B, n, mode) bytes of device scratch.
 *   ev_begin, ev_end   optional hipEvent_t pair (NULL to skip) recorded on `stream` immediately
 *              before and after the tour-construction kernel itself (not the P = tau^a*eta^b
 *              pre-pass), so a caller can time the dominant kernel without a profiler.
 */
size_t daco_tsp_sample_workspace_bytes(int B, int n, int mode);
int daco_tsp_sample(void *stream, int B, int n, int A,
                    const float *tau, long tau_bstride, const float *eta, long eta_bstride,
                    float alpha, float beta, int mode, int norm_passes,
                    const int64_t *start, int fixed_start, const float *noise,
                    uint64_t seed, uint64_t iter, const uint64_t *iter_offset, uint32_t ant_gid0,
                    int ant_gid_bstride,
                    int64_t *paths, float *logp, float *rowsum, int32_t *flags,
                    const float *dist, long dist_bstride, float *costs, uint32_t *nbr,
                    void *workspace, size_t workspace_bytes, void *ev_begin, void *ev_end);

/* ---------------------------------------------------------------------------------------------
 * daco_tsp_sample_sparse -- ACO.gen_path on HEAD / TAIL rows (sampler "scan_sparse"); replaces
 *   tsp/aco.py:134-177 with the roulette of tsp_nls/aco.py:260-275, for heuristics made k-sparse by
 *   tsp/aco.py:52-67 (sparsify: k live entries per row, 1e-10 elsewhere) -- the reference's inference setting.
 *
 * Same distribution as daco_tsp_sample(DACO_SCAN): p_k = tau^alpha * eta^beta * [k unvisited].  A row is split into a
 * head (up to 63 or 127 candidates, head_id) and the tail (the rest).  A step draws r = u * (H + T) with H the head's
 * live mass and T the tail's static mass: inside the head -> inverse CDF over 64 / 128 slots (384 / 768 bytes instead of a
 * row of 4n); past the head -> inverse CDF over all tail entries, and if that lands on a visited one the step draws once
 * more, the dense masked draw over all open candidates with a second uniform (rejection over a superset with an exact
 * fallback: the outcome is the categorical above, a step reads the row at most twice); no live head candidate -> the
 * dense masked draw of the 64-lane scan specification.  Its own uniform stream: tours differ from DACO_SCAN's under the
 * same seed, the distribution does not.  Specification: oracle/daco_oracle.c draw_scan_sparse.
 *   129 <= n <= 1024.
 *   head_slots  64 or 128: slots per row of head_id (four / eight per lane of the 16-lane row)
 *   head_id  [B][n][head_slots] uint16: slots 0..cnt-1 the head's node ids (any subset of the row; the colony passes the
 *            k largest heuristic entries, ids ascending), the other slots 0, the last slot = cnt (<= head_slots - 1)
 *   paths, flags, dist / costs, nbr, start / fixed_start, seed / iter / iter_offset / ant_gid0 / ant_gid_bstride,
 *   ev_begin / ev_end: as daco_tsp_sample (log-probabilities are not produced: an inference sampler)
 *   stats    optional out [3] uint64 (caller zeroes): dense masked draws (no live head candidate, or after a rejection), tail walks, rejections
 *   workspace  daco_tsp_sparse_workspace_bytes(B, n, A) bytes of device scratch: the head rows of this iteration (16 lanes x
 *              {4 or 8 f32 values, as many u16 ids} per row of tau, at offset 0) and, for n > 512, the tours as they are built.
 *              The steps that leave the head walk tau^alpha * eta^beta formed from the rows of `tau` and `eta` themselves
 *              (until version 124 a dense copy of that matrix lived here: 65 MB written per call at TSP-500 x 64).
 *              alpha, beta other than 1: the kernels themselves take unit exponents; tau^alpha and eta^beta are formed first (one
 *              elementwise launch; x^2 = x x and x^0 = 1 exactly) into a workspace of daco_tsp_sparse_workspace_bytes_general(B, n, A)
 *              bytes (= the above + two [B][n][n] f32 matrices), which such a call needs (DACO_E_WORKSPACE otherwise).
 */
size_t daco_tsp_sparse_workspace_bytes(int B, int n, int A);
size_t daco_tsp_sparse_workspace_bytes_general(int B, int n, int A);
/* daco_tsp_sparse_tours_offset -- where the workspace keeps the tours in their compact form: uint16 [B][A][ld], ld = 512 (n <= 512) or
 * 1024, entry t of ant a = the node visited at step t (entries >= n: undefined).  Valid after a call of daco_tsp_sample_sparse /
 * _race_head / _heads with n > 512 (the tours are built there), and for n <= 512 after a call with paths = NULL (nbr given): the
 * colony loop of ACO.run (tsp/aco.py:75-92) keeps `paths` to itself -- costs, best tour and deposit are all it takes from them -- so
 * its iteration asks for no int64 [B][n][A] tensor (131 MB per iteration at TSP-500 x 512 x 64) and hands these rows to
 * daco_track_best_tours16. */
size_t daco_tsp_sparse_tours_offset(int B, int n, int A);
int daco_tsp_sample_sparse(void *stream, int B, int n, int A,
                           const float *tau, long tau_bstride, const float *eta, long eta_bstride,
                           float alpha, float beta, const uint16_t *head_id, int head_slots,
                           const int64_t *start, int fixed_start,
                           uint64_t seed, uint64_t iter, const uint64_t *iter_offset, uint32_t ant_gid0,
                           int ant_gid_bstride,
                           int64_t *paths, int32_t *flags,
                           const float *dist, long dist_bstride, float *costs, uint32_t *nbr,
                           unsigned long long *stats,
                           void *workspace, size_t workspace_bytes, void *ev_begin, void *ev_end);

/* daco_tsp_sample_race_head -- daco_tsp_sample(DACO_RACE_PHILOX) on the same head rows, with the SAME tours (same seed,
 * same noise indexing by node id): the exponential race of torch.multinomial's one-sample path (tsp/aco.py:174-175) is won
 * by a head candidate whenever its key L/p is below what any tail candidate could draw, L_min / max_tail(p); that is checked
 * every step and the dense race runs for the ant otherwise.  head_slots variates per step instead of n.  Arguments as
 * daco_tsp_sample_sparse (stats[0] = steps that took the dense race). */
int daco_tsp_sample_race_head(void *stream, int B, int n, int A,
                              const float *tau, long tau_bstride, const float *eta, long eta_bstride,
                              float alpha, float beta, const uint16_t *head_id, int head_slots,
                              const int64_t *start, int fixed_start,
                              uint64_t seed, uint64_t iter, const uint64_t *iter_offset, uint32_t ant_gid0,
                              int ant_gid_bstride,
                              int64_t *paths, int32_t *flags,
                              const float *dist, long dist_bstride, float *costs, uint32_t *nbr,
                              unsigned long long *stats,
                              void *workspace, size_t workspace_bytes, void *ev_begin, void *ev_end);

/* daco_tsp_sample_heads -- the two calls above behind switches, for a colony loop that keeps its workspace (tsp/aco.py:75-92
 * ACO.run: sample -> costs -> best -> update, again and again):
 *   race         0: daco_tsp_sample_sparse, 1: daco_tsp_sample_race_head
 *   heads_ready  0: the head rows are formed first (one pass over tau and eta, as the two calls above do);
 *                1: `workspace` already holds this iteration's head rows -- daco_pheromone_update_heads wrote them when it
 *                   finished the rows of tau, for the same tau / eta / alpha / beta / head_id / head_slots / race -- and no
 *                   pass over tau runs.  The caller vouches that tau has not changed since.  Same tours either way.
 *   head_live_max   0, or an upper bound of the live counts in head_id (the k of the colony's head table, <= 62): a launch of
 *                FEW ants (B * ceil(A / 4) <= 256 -- one instance with a few hundred ants, the reference's own call pattern,
 *                tsp/test.ipynb:66-68) then keeps the instance's head rows in LDS when they fit (n <= 512, 64-slot heads,
 *                n * ceil((k + 1) / 4) * 24 bytes <= ~150 KB: TSP-500 with k = 50 does), one wavefront of four ants per
 *                workgroup: the same tours, the per-step L2 round trip gone.  A row with more live slots than the bound sets
 *                bit 2 of flags[b].
 *   nbr_grouped  0: nbr as [B][n][A] (what daco_pheromone_update takes); 1: nbr as [B][ceil(A/8)][n][8] -- the eight ants' entries
 *                of a node together, so that the sampler writes the table in 1 KB pieces instead of 32-byte runs and the update
 *                reads it in runs of R x 32 bytes (daco_pheromone_update_heads(nbr_grouped = 1) takes this form). */
int daco_tsp_sample_heads(void *stream, int race, int heads_ready, int head_live_max, int nbr_grouped, int B, int n, int A,
                          const float *tau, long tau_bstride, const float *eta, long eta_bstride,
                          float alpha, float beta, const uint16_t *head_id, int head_slots,
                          const int64_t *start, int fixed_start,
                          uint64_t seed, uint64_t iter, const uint64_t *iter_offset, uint32_t ant_gid0,
                          int ant_gid_bstride,
                          int64_t *paths, int32_t *flags,
                          const float *dist, long dist_bstride, float *costs, uint32_t *nbr,
                          unsigned long long *stats,
                          void *workspace, size_t workspace_bytes, void *ev_begin, void *ev_end);

/* ---------------------------------------------------------------------------------------------
 * daco_cvrp_sample -- replaces the CVRP ACO.gen_path / pick_move / update_visit_mask /
 * update_capacity_mask / check_done  (cvrp/aco.py:138-205 = cvrp_nls/aco.py:205-272)
 *
 * Node 0 is the depot, n counts the depot.  Every ant starts at the depot and draws from
 * p_k = tau[prev][k]^alpha * eta[prev][k]^beta * visit_mask_k * capacity_mask_k until it is back
 * at the depot with every customer served.  Masks as in the reference: visited customers
 * closed; the depot closed only while the ant stands on it with customers left; candidates with
 * demand_k > capacity - used closed (strict); used resets to 0 at the depot.
 *   demand   [B][n] f32 (demand[0] = 0), capacity scalar
 *   mode     as daco_tsp_sample (DACO_RACE_NOISE: noise [B][noise_steps][A][n], Categorical's
 *            single normalisation)
 *   paths    out [B][Lmax][A] int64; rows past an ant's route are 0 (the reference pads with the
 *            depot until the slowest ant is done); Lmax <= 2n+1 always suffices
 *   logp     out [B][Lmax-1][A] f32 or NULL (padding rows = log(1-eps), as in the reference)
 *   rowsum   out [B][Lmax-1][A] f32 or NULL (with logp): S per draw, for daco_sample_backward
 *   lens     out [B][A] int32 or NULL: rows used by each ant; the reference's L = max(lens)
 *   flags    out [B] int32 or NULL: bit 0 = a draw had no feasible candidate, bit 1 = Lmax or
 *            the noise tensor was too short; caller zeroes it.
 *   dist, costs   optional fusion of gen_path_costs (cvrp/aco.py:133-136): if costs != NULL,
 *            costs [B][A] receives sum_k dist[u_k][u_k+1] over the ant's own route (k ascending).
 *            The reference also adds the (0,0) padding edges of the shorter routes, dist[0][0] =
 *            1e-10 each (cvrp/utils.py): no-ops in f32 for any route longer than 2e-3.
 *   next_table  optional out, daco_directed_table_bytes(B, n, A) bytes: per (node, ant) the node that
 *            follows it plus, per ant, the set of nodes that follow the depot -- the form
 *            daco_pheromone_update(symmetric = 0, nbr = next_table, hub = 0) consumes.
 *   workspace daco_tsp_sample_workspace_bytes(B, n, mode)
 *   demand64, capacity64   NULL / 0, or [B][n] float64 demands and the capacity as a double: the load bookkeeping of
 *            cvrp_nls/aco.py:254-272 (used = used + demand[cur]; demand > capacity - used) then runs in double, as it
 *            does in the reference for that directory's float64 instance data (demands k / capacity: a customer that
 *            fits exactly is common and the last bit decides); `demand` / `capacity` must still be given (their float32
 *            images).  Every layout of the scan draw has the variant (packed kernels for n <= 512, one ant per wavefront above and in the
 *            race modes).
 *   ant_gid_bstride    as in daco_tsp_sample: 0, or the colony's ant count when this call draws a slice [ant_gid0, ant_gid0 + A)
 *            of every colony's ants (ant-sharded colonies keep the single-GPU ant ids)
 *   ev_begin, ev_end   optional hipEvent_t recorded on `stream` right before / after the construction kernel (as in
 *            daco_tsp_sample: the kernel alone, without the weight-matrix kernel and the table memsets before it).
 */
size_t daco_directed_table_bytes(int B, int n, int A);
int daco_cvrp_sample(void *stream, int B, int n, int A,
                     const float *tau, long tau_bstride, const float *eta, long eta_bstride,
                     float alpha, float beta, const float *demand, float capacity, int mode,
                     const float *noise, int noise_steps, uint64_t seed, uint64_t iter,
                     const uint64_t *iter_offset, uint32_t ant_gid0, int ant_gid_bstride, int Lmax, int64_t *paths, float *logp,
                     float *rowsum,
                     int32_t *lens, int32_t *flags,
                     const float *dist, long dist_bstride, float *costs, void *next_table,
                     void *workspace, size_t workspace_bytes, const double *demand64, double capacity64,
                     void *ev_begin, void *ev_end);

/* ---------------------------------------------------------------------------------------------
 * daco_track_best -- replaces the best-so-far bookkeeping inside ACO.run
 *   tsp/aco.py:78-88, cvrp/aco.py:78-100: best_cost, best_idx = costs.min(dim=0);
 *   if best_cost < self.lowest_cost: shortest_path = paths[:, best_idx]; lowest_cost = best_cost
 *   (MMAS: max = n / lowest_cost).  Done on the device, so the loop needs no host branch.
 *   costs [B][A]; paths [B][len][A] int64; lowest [B] f32 in/out (start at +inf); shortest [B][len]
 *   int64 in/out or NULL; best_idx [B] int32 out or NULL (first minimum, torch.min semantics);
 *   mmas_max [B] f32 out or NULL = (1 / lowest) * mmas_scale, the two roundings of the reference's
 *   `problem_size / lowest_cost` on a tensor (reciprocal, then multiply).
 */
int daco_track_best(void *stream, int B, int len, int A, const float *costs, const int64_t *paths,
                    float *lowest, int64_t *shortest, int32_t *best_idx, float *mmas_max, float mmas_scale);
/* the same bookkeeping on the compact tours of daco_tsp_sparse_tours_offset: tours16 [B][A][ld] uint16; shortest stays int64 [B][len] */
int daco_track_best_tours16(void *stream, int B, int len, int A, int ld, const float *costs, const uint16_t *tours16,
                            float *lowest, int64_t *shortest, int32_t *best_idx, float *mmas_max, float mmas_scale);

/* ---------------------------------------------------------------------------------------------
 * daco_prob_matrix + daco_pick_move -- ACO.pick_move as a step-wise service
 *   tsp/aco.py:165-177 and its copies in op/aco.py:186-193, pctsp/aco.py:157-164,
 *   sop/aco.py:156-169, smtwtp/aco.py:139-151, bpp/aco.py:157-164, mkp/aco.py:147-154
 * The sibling problems keep their feasibility rules on the caller's side and only need the
 * draw: dist = tau[prev]^alpha * eta[prev]^beta * mask -> Categorical(dist).sample()/log_prob.
 * daco_prob_matrix builds the fused transition matrix once per solution construction into
 * `workspace` (daco_tsp_sample_workspace_bytes(B, n, mode) bytes); daco_pick_move then performs
 * ONE draw per ant: prev [B][A] int64, mask [B][A][n] f32 (0 = closed), actions out [B][A] int64,
 * logp / rowsum out [B][A] or NULL, noise [B][A][n] for DACO_RACE_NOISE.  `step` keys the Philox
 * counter (use the step index of the construction loop).
 */
int daco_prob_matrix(void *stream, int B, int n, const float *tau, long tau_bstride,
                     const float *eta, long eta_bstride, float alpha, float beta, int mode,
                     void *workspace, size_t workspace_bytes);
int daco_pick_move(void *stream, int B, int n, int A, const void *prob_workspace,
                   size_t workspace_bytes, int mode, const int64_t *prev, const float *mask,
                   const float *noise, uint64_t seed, uint64_t iter, uint32_t ant_gid0, int step,
                   int64_t *actions, float *logp, float *rowsum, int32_t *flags);

/* ---------------------------------------------------------------------------------------------
 * daco_sibling_sample -- fused solution construction for sop / pctsp / op / mkp (one launch)
 *   DACO_SIB_SOP   sop/aco.py:114-180    aux_vec[k] = number of predecessors of k (row sums of
 *                  prec_cons), aux_mat[k][j] = prec_cons[j][k] ("who waits for k"); n-1 draws
 *                  from node 0; paths [B][n][A]; Lmax/lens unused
 *   DACO_SIB_PCTSP pctsp/aco.py:131-188  aux_vec = prizes, scalar0 = min prize; the depot opens
 *                  when the collected prize exceeds scalar0 or nothing is left; ends at the depot
 *   DACO_SIB_OP    op/aco.py:156-224     n counts the dummy end node n-1; aux_mat = distances
 *                  (with the dummy row/column), aux_vec[k] = distances[k][0], scalar0 = max_len;
 *                  a candidate closes for good once route + leg + way home exceeds scalar0
 *   DACO_SIB_MKP   mkp/aco.py:113-183    n counts the dummy item n-1; item_weights [B][n][m],
 *                  scalar0 = capacity (n_items // 2); start [B][A] or NULL (Philox)
 * Variable-length kinds write paths [B][Lmax][A] padded with the resting node (depot 0 / dummy
 * n-1) and lens [B][A]; logp/rowsum [B][rows-1][A] optional; flags as daco_cvrp_sample.  aux_mat
 * is dense [B][n][n] (aux_mat_bstride between instances, 0 = shared).  Draws, modes, noise layout
 * ([B][noise_steps][A][n]) and Philox counters are those of daco_tsp_sample / daco_pick_move.
 */
#define DACO_SIB_SOP 3
#define DACO_SIB_PCTSP 4
#define DACO_SIB_OP 5
#define DACO_SIB_MKP 6
size_t daco_sibling_workspace_bytes(int B, int n, int mode);
int daco_sibling_sample(void *stream, int kind, int B, int n, int A,
                        const float *tau, long tau_bstride, const float *eta, long eta_bstride,
                        float alpha, float beta, const float *aux_vec, const float *aux_mat,
                        long aux_mat_bstride, float scalar0, const float *item_weights, int m,
                        int mode, const int64_t *start, const float *noise, int noise_steps,
                        uint64_t seed, uint64_t iter, uint32_t ant_gid0, int Lmax,
                        int64_t *paths, float *logp, float *rowsum, int32_t *lens, int32_t *flags,
                        void *workspace, size_t workspace_bytes);

/* ---------------------------------------------------------------------------------------------
 * daco_sibling_backward -- daco_sample_backward for the fused sibling constructions: the gradient of
 * sum(grad_logp * log_probs) w.r.t. eta for solutions drawn by daco_sibling_sample (same kind, aux_vec,
 * aux_mat, scalar0, item_weights; paths / rowsum / lens as that call returned them).  aux_mat is the
 * caller's dense [B][n][n] matrix.  n <= 1024.  grad_eta [B][n][n] is accumulated into (caller zeroes).
 */
int daco_sibling_backward(void *stream, int kind, int B, int n, int A, int rows,
                          const float *tau, long tau_bstride, const float *eta, long eta_bstride,
                          float alpha, float beta, const float *aux_vec, const float *aux_mat,
                          long aux_mat_bstride, float scalar0, const float *item_weights, int m,
                          const int64_t *paths, const float *rowsum, const float *grad_logp,
                          const int32_t *lens, float *grad_eta);

/* ---------------------------------------------------------------------------------------------
 * daco_sample_backward -- replaces autograd through ACO.gen_path(require_prob=True)
 *   (tsp/aco.py:154-176, cvrp/aco.py:153-173; consumed by the REINFORCE losses in
 *    tsp/train.ipynb:45-49, tsp_nls/train.py:31-44, cvrp/train.ipynb:45-51)
 * grad_eta[b][i][k] += sum over draws (t,a) made from node i of
 *     grad_logp[t][a] * beta * ( [k = action] / eta_ik - p_k / (eta_ik * S_ta) ),  0 where the
 * probability was clamped.  rows = n (TSP) or Lmax (CVRP: pass demand, capacity and lens; NULL
 * demand selects TSP).  rowsum is what the sampler wrote.  grad_eta [B][n][n] must be zeroed (or
 * hold a gradient to accumulate into).  f32 hardware atomics: reproducible to rounding.
 * demand64 / capacity64: NULL / 0, or the float64 demands and capacity daco_cvrp_sample drew the routes with (the
 * capacity rule is then replayed in double as well; demand must still be given).
 */
int daco_sample_backward(void *stream, int B, int n, int A, int rows,
                         const float *tau, long tau_bstride, const float *eta, long eta_bstride,
                         float alpha, float beta, const int64_t *paths, const float *rowsum,
                         const float *grad_logp, const int32_t *lens, const float *demand,
                         float capacity, float *grad_eta, const double *demand64, double capacity64);

/* ---------------------------------------------------------------------------------------------
 * daco_tour_costs -- replaces ACO.gen_path_costs
 *   closed = 1: sum_k dist[u_k][u_{k-1}] over the closed tour     (tsp/aco.py:121-132),
 *               added in the order k = 1..len-1 and the closing edge dist[u_0][u_{len-1}] last
 *   closed = 0: sum_{k < len-1} dist[u_k][u_{k+1}]                  (cvrp/aco.py:133-136)
 * paths [B][len][A] int64, dist [B][n][n] (dist_bstride as above), costs out [B][A] f32.
 * Summation is sequential from +0.0f (documented order; the reference's torch.sum order is
 * unspecified and agreement with it is 1e-5 relative).
 */
int daco_tour_costs(void *stream, int B, int n, int len, int A, const float *dist,
                    long dist_bstride, const int64_t *paths, int closed, float *costs);

/* ---------------------------------------------------------------------------------------------
 * daco_pheromone_update -- replaces ACO.update_pheronome
 *   tsp/aco.py:95-118   (symmetric = 1: both directions of every tour edge, closed tour)
 *   cvrp/aco.py:107-130 (symmetric = 0: directed path[k] -> path[k+1], k < len-1; duplicate
 *                        index pairs of one ant collapse to a single add; floor applied)
 * tau <- tau*decay, then for each ant in index order (elitist = 1: only the first-minimum-cost
 * ant) w = 1/cost and every edge of its tour receives += w; then optional MMAS clamp
 * (clamp_max > 0: tau < clamp_min -> clamp_min, tau > clamp_max -> clamp_max) and optional
 * floor (floor_val > 0: tau < floor_val -> floor_val).  Bit-identical to the reference's
 * sequential loop: each row of tau is owned by one lane-pair and receives its adds in ant
 * order (no atomics).
 *   tau   in/out [B][n][n] f32 (dense, stride n*n)
 *   paths [B][len][A] int64, costs [B][A] f32
 *   clamp_min/clamp_max: [B] f32 device arrays or NULL (per-instance MMAS bounds)
 *   nbr   optional: symmetric: [B][n][A] uint32 as written by daco_tsp_sample; directed: the
 *         next_table written by daco_cvrp_sample (hub must be 0); if NULL it is rebuilt from
 *         `paths` in the workspace
 *   weights optional [B][A] f32: the amount each ant deposits, for the siblings whose rule is
 *         not 1/cost (op/aco.py:134-139 Q*obj, bpp/aco.py:113-118 fit/n_ants, smtwtp/aco.py:90-95
 *         1/(cost+1)); NULL = 1/cost.  `costs` still selects the elitist ant (first minimum).
 *   hub   directed only: the one node a solution may leave several times (depot 0 in cvrp/
 *         pctsp/bpp, the dummy end node in op/mkp); every other node is left at most once;
 *         -1 if there is none (sop, smtwtp).  Edges (hub,hub) of one ant collapse to one add.
 *   workspace: daco_pheromone_update_workspace_bytes(B, n, len, A)
 */
size_t daco_pheromone_update_workspace_bytes(int B, int n, int len, int A);
int daco_pheromone_update(void *stream, int B, int n, int len, int A, float *tau,
                          const int64_t *paths, const float *costs, float decay, int elitist,
                          int symmetric, const float *clamp_min, const float *clamp_max,
                          float floor_val, const uint32_t *nbr, const float *weights, int hub,
                          void *workspace, size_t workspace_bytes);

/* daco_pheromone_update_heads -- daco_pheromone_update(symmetric = 1) for a colony whose next construction runs on head rows
 * (tsp/aco.py:95-118 followed by the next iteration's tsp/aco.py:165-172): the workgroup that has just finished rows of tau
 * (evaporation, deposits in ant order, clamp, floor -- bit for bit the update above) also forms their head rows for
 * daco_tsp_sample_heads(heads_ready = 1) from the copy it still holds on chip, so tau is read once per iteration instead of twice.
 *   eta, eta_bstride, alpha, beta, head_id, head_slots, race   as the sampler will be called; alpha = beta = 1 only (other
 *                exponents take the sampler's own pass: DACO_E_BADARG here)
 *   nbr_grouped        the layout of `nbr` (see daco_tsp_sample_heads); a NULL nbr is rebuilt from `paths` either way (paths may be
 *                      NULL when nbr is given: the table is all the deposit reads)
 *   sparse_workspace   the sampler's workspace (daco_tsp_sparse_workspace_bytes(B, n, A)); its head rows are (re)written
 *   129 <= n <= 1024; the other arguments as daco_pheromone_update (len = n, hub unused) */
int daco_pheromone_update_heads(void *stream, int B, int n, int A, float *tau,
                                const int64_t *paths, const float *costs, float decay, int elitist,
                                const float *clamp_min, const float *clamp_max, float floor_val,
                                const uint32_t *nbr, const float *weights, void *workspace, size_t workspace_bytes,
                                const float *eta, long eta_bstride, float alpha, float beta,
                                const uint16_t *head_id, int head_slots, int race, int nbr_grouped,
                                void *sparse_workspace, size_t sparse_workspace_bytes);

/* daco_allreduce_delta_tau -- the ant-sharded colony's one data-path collective (every rank builds the deposits
 * of ITS ants, delta-tau [B][n][n] f32 = daco_pheromone_update(decay = 1) onto zeros; the ranks sum them; every rank applies
 * tau <- decay * tau + delta) for hosts without torch.distributed: ncclAllReduce(sum, f32, in place) on `stream` through the
 * RCCL the process holds (resolved at run time; no link-time dependency).  comm: the caller's ncclComm_t (one per rank, created
 * with that same RCCL); count: elements of delta.  The Python host of this package issues the same collective through
 * torch.distributed (deepaco_amd/parallel.py), whose communicators are torch's. */
int daco_allreduce_delta_tau(void *comm, void *stream, float *delta, size_t count);

/* ---------------------------------------------------------------------------------------------
 * daco_two_opt -- replaces batched_two_opt_python / _two_opt_python / two_opt_once
 *   tsp_nls/two_opt.py:6-49
 * Best-improvement 2-opt on every tour until no move improves by more than 1e-6 or
 * max_iterations sweeps were done (the reference's loop, including the final non-improving
 * sweep in the count).  Bit-identical to the reference: same f32 expression order, strict
 * minimum with ties to the first (i,j) in row-major order.
 *   dist   [B][n][n] f32 (dist_bstride as above; need not be symmetric -- the NLS driver also
 *          runs it on the perturbed matrix, tsp_nls/aco.py:230-232,248)
 *   dist_T NULL, or the transposed matrices [B][n][n] (dist itself when it is symmetric): lets the incremental
 *          kernel read d[t[i-1]][t[j]] as a row access of the changed segment's nodes (same values, ~18x less L2
 *          traffic); results do not depend on whether it is given
 *   tours  in/out [B][T][n] uint16, one row per tour (the reference's numpy layout, note: the
 *          transpose of `paths`)
 *   sweeps out [B][T] int32 or NULL: sweeps performed per tour
 */
int daco_two_opt(void *stream, int B, int T, int n, const float *dist, const float *dist_T, long dist_bstride,
                 uint16_t *tours, long max_iterations, int32_t *sweeps);

/* ---------------------------------------------------------------------------------------------
 * daco_two_opt_prepare / daco_two_opt_nbr -- the same search as daco_two_opt (tsp_nls/two_opt.py:6-49: same moves,
 * same tours, same sweep counts, bit for bit), evaluating per sweep only the pairs that can be the reference's strict
 * minimum: for the tour edge (x, y) the nodes v with d[x][v] < d[x][y] + tol and the nodes u with d[u][y] < d[x][y] + tol
 * (tol = 4 ulp(2 max|d|) covers the three f32 roundings of the reference's expression; csrc/daco_two_opt_nbr.hip has
 * the argument).  A few thousand evaluations per sweep instead of n^2/2 once tours are near a local optimum (the
 * perturbation / repair passes of the NLS); more than n^2/2 for tours with many long edges (daco_two_opt_auto chooses).
 *   daco_two_opt_prepare: builds, per instance, the sorted neighbour lists and tolerance ranks of `dist` [B][n][n]
 *          (n <= 1024) into `tables` (daco_two_opt_tables_bytes(B, n) bytes).  Once per matrix.
 *   daco_two_opt_nbr: tables = prepare(dist); tables_T = prepare(transposed dist), or the same pointer when dist is
 *          symmetric.  tours / max_iterations / sweeps as daco_two_opt.
 */
size_t daco_two_opt_tables_bytes(int B, int n);
int daco_two_opt_prepare(void *stream, int B, int n, const float *dist, long dist_bstride, void *tables, size_t tables_bytes);
int daco_two_opt_nbr(void *stream, int B, int T, int n, const float *dist, long dist_bstride, const void *tables,
                     const void *tables_T, uint16_t *tours, long max_iterations, int32_t *sweeps);
/*   daco_two_opt_auto: both kernels on one call, chosen per tour and per phase of its search without a host round
 *          trip (three launches: candidates, dense, candidates): the candidate-list kernel while a tour's lists hold
 *          fewer than ~n^2/6 entries (n^2/10 for a non-symmetric matrix), the dense incremental kernel (daco_two_opt's;
 *          dist_T as there, may be NULL) above, until its running rank sum has fallen under half of that -- tours fresh
 *          from a dense heuristic start dense and finish on the candidate lists; with at most 2048 tours the candidate
 *          kernel keeps them all (and runs 1024 threads per tour below 512 tours).  Same result as either kernel alone.
 *          sweeps [B][T] int32 is REQUIRED here (it carries the per-tour state between the launches).
 */
int daco_two_opt_auto(void *stream, int B, int T, int n, const float *dist, const float *dist_T, long dist_bstride,
                      const void *tables, const void *tables_T, uint16_t *tours, long max_iterations, int32_t *sweeps);

/* ---------------------------------------------------------------------------------------------
 * daco_tsp_nls -- replaces ACO.nls (and, with T_nls = 0, ACO.two_opt) in ONE launch
 *   tsp_nls/aco.py:234-258 over tsp_nls/two_opt.py:6-39
 * Per tour: 2-opt on `dist` (at most max_iterations sweeps); then T_nls rounds of { T_p sweeps of 2-opt on `hdist` (the
 * perturbation matrix 1/(eta/rowmax + 1e-5), tsp_nls/aco.py:230-232), 2-opt on `dist` again, keep the tour if its f32
 * length (daco_tour_costs' summation order) is strictly below the best so far }.  The moves of every pass are the
 * reference's (same candidate rule and pair arithmetic as daco_two_opt_nbr); a sweep re-walks only the candidate lists
 * that the previous move can have changed and reuses the cached minimum of every other list (csrc/daco_nls.hip).
 *   tables / tables_T, htables / htables_T: daco_two_opt_prepare of dist / hdist and of their transposes (the same
 *          pointer when the matrix is symmetric); hdist and its tables may be NULL when T_nls = 0
 *   tours  in/out [B][T][n] uint16: the best tour found
 *   sweeps out [B][T] int32 or NULL: sweeps over all passes;  costs out [B][T] f32 or NULL: length of the returned tour
 *   counters NULL or two uint64 on the device, incremented by: [0] sweeps, [1] list entries walked (bench bookkeeping)
 */
int daco_tsp_nls(void *stream, int B, int T, int n, const float *dist, long dist_bstride, const void *tables,
                 const void *tables_T, const float *hdist, long hdist_bstride, const void *htables, const void *htables_T,
                 uint16_t *tours, long max_iterations, int T_nls, long T_p, int32_t *sweeps, float *costs,
                 unsigned long long *counters);

/* ---------------------------------------------------------------------------------------------
 * daco_gnn_forward -- replaces Net.forward in eval mode
 *   EmbNet.forward tsp/net.py:27-45, MLP/ParNet.forward :59-66,74-75, Net.forward :84-88
 * One graph: n nodes with `feats` input features, E directed edges with one attribute each.
 *   x [n][feats], edge_attr [E], src/dst [E] int32 (edge_index rows), rowptr [n+1] int32 = CSR
 *   offsets of the edges grouped by src, perm [E] int32 = edge ids in that grouped order or NULL
 *   if the edge list is already sorted by src (the reference's kNN graphs are).
 *   params: daco_gnn_param_floats(feats) floats, layout documented in csrc/daco_gnn.hip
 *           (BatchNorm folded to scale/shift from the running statistics -> eval mode only).
 *   heu out [E] f32 in (0,1);  emb out [E][32] or NULL (the edge embedding before the head).
 */
size_t daco_gnn_param_floats(int feats);
size_t daco_gnn_workspace_bytes(int n, int E);
int daco_gnn_forward(void *stream, int n, int E, int feats, const float *x, const int32_t *src,
                     const int32_t *dst, const int32_t *rowptr, const int32_t *perm,
                     const float *edge_attr, const float *params, float *heu, float *emb,
                     void *workspace, size_t workspace_bytes);

/* ---------------------------------------------------------------------------------------------
 * daco_gnn_train_forward / daco_gnn_train_backward -- replace Net.forward in TRAINING mode and the backward
 * autograd derives from it
 *   EmbNet.forward tsp/net.py:27-45 with BatchNorm on the statistics of the one graph (:21,24,43-44),
 *   MLP/ParNet.forward :59-75; loss.backward() of tsp_nls/train.py:15-44 as far as the network goes.
 * G equal-sized graphs side by side (n = G * n_g nodes, E = G * E_g edges, graph g owns nodes [g*n_g, (g+1)*n_g) and
 * edges [g*E_g, (g+1)*E_g), node ids already offset): every graph is normalised with its own statistics.
 *   x, src, dst, rowptr, perm, edge_attr   as daco_gnn_forward
 *   params     daco_gnn_param_floats(feats) floats in the layout of csrc/daco_gnn.hip with the BatchNorm slots
 *              holding weight (gamma) and bias (beta) instead of the folded scale / shift
 *   heu        out [E]
 *   stats_out  out [12][2][G][32][2] f32 or NULL: (mean, biased variance) of every BatchNorm (index 0 = edge BN,
 *              1 = node BN of the layer), for the caller's running-statistics update
 *   fixed_stats (forward) NULL, or [12][2][32][2] f32 (mean, variance) per layer and BatchNorm (0 = edge, 1 = node): BatchNorm in
 *              EVALUATION mode -- these running statistics normalise every graph instead of its own batch statistics (a
 *              module in eval() whose output still needs a gradient); (backward) the same fact as a flag: the statistics
 *              were constants, g_z = gamma * rstd * g_y
 *   workspace  daco_gnn_train_workspace_bytes(n, E, G) bytes; the forward leaves the activations the backward needs
 *              there: pass the SAME, untouched block to daco_gnn_train_backward
 *   grad_heu   [E] d loss / d heu;   grad_params out: d loss / d params, same layout as params
 *   rowptr_dst, perm_dst (backward)  CSR of the edges grouped by DESTINATION (offsets [n+1], edge ids [E]) or NULL/NULL
 *              (the backward then builds it in the workspace: count, scan, fill, ids ascending per row).  The node
 *              gradients are CSR row sums of per-edge contributions: fixed order, no float atomics.
 *              perm (backward): as in the forward.
 * No library GEMM is called: the 32x32 linears and their weight gradients run on v_mfma_f32_32x32x2_f32.
 */
size_t daco_gnn_train_workspace_bytes(int n, int E, int G);
int daco_gnn_train_forward(void *stream, int n, int E, int feats, int G, const float *x, const int32_t *src,
                           const int32_t *dst, const int32_t *rowptr, const int32_t *perm, const float *edge_attr,
                           const float *params, float *heu, float *stats_out, const float *fixed_stats, void *workspace,
                           size_t workspace_bytes);
int daco_gnn_train_backward(void *stream, int n, int E, int feats, int G, const float *x, const int32_t *src,
                            const int32_t *dst, const int32_t *rowptr, const int32_t *perm, const int32_t *rowptr_dst,
                            const int32_t *perm_dst, const float *edge_attr, const float *params, const float *heu,
                            const float *grad_heu, float *grad_params, int fixed_stats, void *workspace, size_t workspace_bytes);

/* ---------------------------------------------------------------------------------------------
 * daco_cvrp_local_search -- replaces ACO.multiple_swap_star's per-ant CPU tasks
 *   cvrp_nls/aco.py:114-126 (one swapstar() call per ant through a thread pool), cvrp_nls/swapstar.py:240-271
 *   (/tmp-file hand-over) and the entry it calls in the vendored HGS-CVRP (Program/C_Interface.cpp:128-172).
 * A deterministic best-improvement search over relocate / swap / intra-route 2-opt moves (specified in
 * csrc/daco_cvrp_ls.hip, restated in oracle/cvrp_ls.py); HGS's own LocalSearch (third-party, randomised neighbourhood
 * order) is NOT reproduced move for move: results are feasible, never worse, and local optima of these neighbourhoods.
 *   dist     [B][n][n] f32 (dist_bstride elements between instances, 0 = shared); need not be symmetric
 *   demand   [B][n] f32, demand[.][0] = 0;  capacity: vehicle capacity in the same unit
 *   paths    in/out [B][Lmax][A] int64: column (b, a) is ant a's route sequence 0 a b 0 c d 0 ... zero-padded
 *            (the layout of ACO.gen_path, cvrp/aco.py:138-165); rewritten in place without empty routes
 *   max_moves  moves applied at most per solution (the reference's `count`/`limit`)
 *   lens     out [B][A] int32 or NULL: entries used by each solution (including the closing depot)
 *   moves    out [B][A] int32 or NULL: moves applied
 */
int daco_cvrp_local_search(void *stream, int B, int n, int A, int Lmax, const float *dist, long dist_bstride,
                           const float *demand, float capacity, int64_t *paths, int max_moves, int32_t *lens,
                           int32_t *moves);

/* ---------------------------------------------------------------------------------------------
 * daco_hgs_prepare / daco_hgs_local_search -- the reference's CVRP local search, ROUTE FOR ROUTE
 *   replaces cvrp_nls/aco.py:114-126 (multiple_swap_star: one swapstar() call per ant through a thread pool),
 *   cvrp_nls/swapstar.py:324-346 (the HGS set-up: demands * 1000, capacity 1000.001, num_vehicles = number of routes),
 *   swapstar.py:187-271 (/tmp-file hand-over) and the entry it binds, HGS-CVRP-main/Program/C_Interface.cpp:128-172
 *   `local_search` -> LocalSearch::run (LocalSearch.cpp:3-103): moves 1-9 (LocalSearch.cpp:134-484) under the
 *   nbGranular-nearest restriction (Params.cpp:77-103), first improvement in the order std::shuffle(std::minstd_rand)
 *   fixes (libstdc++), load penalty 10 * max(0.1, min(1000, maxDist / maxDemand)), float64 throughout.
 *   The reference's ctypes structure (swapstar.py:62-74) is 10 fields of the header's 15 (AlgorithmParameters.h:10-28):
 *   HGS reads useSwapStar beyond it, so the reference RUNS WITHOUT SWAP* and with zero coordinates (Params.cpp:40-54;
 *   the export order of LocalSearch.cpp:756-778 is then the route index order).  That is what these entries compute: the
 *   reference's routes entry for entry (oracle/hgs_ls.c use_swap_star = 0; fixtures tests/golden/g11_*).
 *
 * daco_hgs_prepare: per instance and matrix, what Params derives from the matrix alone: maxDist, the correlated vertices
 *   (nb_granular nearest by (cost, index), made symmetric, ascending), the shuffled node order of LocalSearch.cpp:9 and the
 *   generator state behind it.
 *   matrix  [B][n][n] f64 (bstride elements between instances); node 0 is the depot
 *   tables  out, B * daco_hgs_table_bytes(n, nb_granular) bytes
 * daco_hgs_local_search: nstages (1..3) calls of `local_search` in a row on every solution (cvrp_nls/aco.py:443-448
 *   neural_swapstar is three: distances / limit, heuristic-derived matrix / 10, distances / limit), the routes of a stage
 *   handed to the next as the reference's files do (non-empty routes, in export order).
 *   matrices[s], bstrides[s], tables[s], counts[s]   stage s: its matrix [B][n][n] f64, the tables daco_hgs_prepare made
 *            of it, and `count` (the loop bound of LocalSearch.cpp:17)
 *   matrices_t[s]   the transposed copy of matrices[s] (same stride), or NULL (the array itself may be NULL) for a symmetric
 *            matrix: timeCost[v][u] with u the node being improved is read as row u of the transpose, so that a wavefront's 64
 *            gathers fall on a few cache lines instead of 64 (the same numbers either way)
 *   demand   [B][n] f64 AS HGS GETS THEM (the caller multiplies by 1000, swapstar.py:335), demand[.][0] = 0
 *   capacity 1000.001 in the reference (swapstar.py:337)
 *   paths    in/out [B][Lmax][A] int64: column (b, a) is a zero-separated route sequence (cvrp/aco.py:138-165); rewritten
 *            as cvrp_nls/aco.py:22-33 merge_subroutes lays the result out ("0 c1 .. ck" per route, zero padded)
 *   status   out [B][A] int32 or NULL: 0 searched; 1 a stage was skipped because HGS throws there (distances or demands
 *            out of scale Params.cpp:106-111, fleet too small :112, infeasible input Individual.cpp:70) -- the reference
 *            then keeps that stage's input (swapstar.py:341-345); 2 the column is not a complete solution (left untouched)
 *   stats    out [B][A][4] int32 or NULL: moves applied, loops run, evaluation rounds (one per node and re-evaluation),
 *            watchdog (0; the per-stage step budget DACO_HGS_BUDGET ran out at: 1 a route walk, 2 an evaluation round)
 *   workspace daco_hgs_workspace_bytes(B, n, A, Lmax, nb_granular) bytes
 */
size_t daco_hgs_table_bytes(int n, int nb_granular);
int daco_hgs_prepare(void *stream, int B, int n, const double *matrix, long bstride, int nb_granular, void *tables);
size_t daco_hgs_workspace_bytes(int B, int n, int A, int Lmax, int nb_granular);
int daco_hgs_local_search(void *stream, int B, int n, int A, int Lmax, int nstages, const double *const *matrices,
                          const double *const *matrices_t, const long *bstrides, const void *const *tables, const int *counts, const double *demand,
                          double capacity, int nb_granular, int64_t *paths, int32_t *status, int32_t *stats,
                          void *workspace, size_t workspace_bytes);

/* ---------------------------------------------------------------------------------------------
 * daco_tsp_knn_graph -- replaces gen_distance_matrix + gen_pyg_data for a batch of instances
 *   tsp/utils.py:4-36, tsp_nls/utils.py:5-45
 * coords [B][n][2] f32 -> dist [B][n][n] (diagonal = diag, 1e9 in the reference; may be NULL),
 * edge_src / edge_dst [B][n*k] int64 (instance-local node ids; sources sorted, k per node,
 * neighbours by ascending distance, ties -> smaller index) and edge_attr [B][n*k] f32.
 */
int daco_tsp_knn_graph(void *stream, int B, int n, int k, const float *coords, float diag, float *dist,
                       int64_t *edge_src, int64_t *edge_dst, float *edge_attr);
/* The same launch, which also writes the batch as ONE block-diagonal graph in the form daco_gnn_forward takes
 * (src32 / dst32 [B*n*k] int32, node ids b*n + i; the CSR row pointer of it is arange(B*n + 1) * k): what the caller
 * would otherwise derive from edge_src / edge_dst with a handful of elementwise launches per forward. */
int daco_tsp_knn_graph_csr(void *stream, int B, int n, int k, const float *coords, float diag, float *dist,
                           int64_t *edge_src, int64_t *edge_dst, float *edge_attr, int32_t *src32, int32_t *dst32);

/* ---------------------------------------------------------------------------------------------
 * daco_heu_matrix -- replaces Net.reshape for a batch, with the "+ eps" its callers add
 *   tsp/net.py:94-102 (matrix = zeros; matrix[edge_index[0], edge_index[1]] = vector), tsp/train.ipynb:35, tsp_nls/test.py:28
 * edge_index [B][2][E] int64 (graph-local ids), heu [B][E] f32 -> out [B][n][n] f32 (16-byte aligned): `fill` everywhere,
 * heu + add at [src][dst] (fill = add = eps gives reshape(...) + eps bit for bit; an edge listed twice: one of its values,
 * as with the reference's indexed assignment).  Ids outside [0, n) are skipped and counted in *bad (may be NULL).
 */
int daco_heu_matrix(void *stream, int B, int n, int E, const int64_t *edge_index, const float *heu, float fill, float add,
                    float *out, int32_t *bad);

#ifdef __cplusplus
}
#endif
#endif /* DEEPACO_HIP_H */
