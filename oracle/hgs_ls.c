/* oracle/hgs_ls.c -- CPU restatement of the reference's CVRP local search.  TEST INFRASTRUCTURE ONLY.
 *
 * What the reference runs per ant (cvrp_nls/aco.py:114-126 -> swapstar.py:324-346 -> swapstar.py:187-271 ->
 * HGS-CVRP-main/Program/C_Interface.cpp:128-172 `local_search`): Params (Params.cpp:5-121), Individual read from the
 * routes (Individual.cpp:38-82), LocalSearch::run(indiv, 10 * penaltyCapacity, 10 * penaltyDuration, count)
 * (LocalSearch.cpp:3-103), exportIndividual (LocalSearch.cpp:756-778), exportCVRPLibFormat (Individual.cpp:85-102).
 * This file restates that path on index arrays instead of pointer-linked nodes: same doubles, same expression
 * order (compiled with -ffp-contract=off), same std::minstd_rand stream through libstdc++'s std::shuffle /
 * uniform_int_distribution (GCC 11 <bits/stl_algo.h> shuffle with the two-draws-in-one pairing, <bits/uniform_int_dist.h>
 * fallback path: scaling = range / (b - a + 1), rejection above scaling * (b - a + 1)).
 *
 * Pinned: tests/test_hgs_ls_oracle.py runs it against oracle/_ref/libhgscvrp.so (HGS built by plain g++ from the
 * reference's own sources) in this container, routes for routes, and against fixtures tests/golden/g11_hgs_ls_*.npz
 * (outputs of the reference's Python over that library) anywhere.
 *
 * What the reference ACTUALLY passes (probed, tests/golden/gen_g11_hgs_ls.py): swapstar.py's CAlgorithmParameters has
 * 10 fields where AlgorithmParameters.h:10-28 has 15, so the C side reads seed = 1 (same minstd state as seed 0: both
 * become state 1) and useSwapStar from beyond the structure (never 1 in any run here): the reference runs the nine
 * classical moves with the 20-nearest granular restriction and NO SWAP*.  use_swap_star = 0 is therefore "the reference
 * as run"; use_swap_star = 1 restates swapStar() (LocalSearch.cpp:486-573) as the sources mean it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define HGS_EPS 0.00001            /* Params.h:41 MY_EPSILON */
#define HGS_PI 3.14159265359       /* Params.h:42 */

/* ---------------------------------------------------------------------------------------------- RNG */
typedef struct { uint64_t x; } minstd_t;                     /* std::minstd_rand: x <- 48271 x mod (2^31 - 1) */
static void minstd_seed(minstd_t* g, uint32_t s) { g->x = s % 2147483647u; if (g->x == 0) g->x = 1; }
static uint64_t minstd_next(minstd_t* g) { g->x = g->x * 48271u % 2147483647u; return g->x; }

/* uniform_int_distribution<unsigned long>{0, hi}(minstd): urng range 2147483645 (min 1, max 2^31 - 2) */
static uint64_t uid(minstd_t* g, uint64_t hi, long* draws) {
    const uint64_t urngrange = 2147483645u;
    const uint64_t uerange = hi + 1, scaling = urngrange / uerange, past = uerange * scaling;
    uint64_t r;
    do { r = minstd_next(g) - 1; if (draws) ++*draws; } while (r >= past);
    return r / scaling;
}

/* std::shuffle(v, v + n, minstd) as libstdc++ does it when range / n >= n (always here: n <= 46340) */
static void shuffle_int(int* v, int n, minstd_t* g, long* draws) {
    if (n <= 0) return;
    int i = 1;
    if ((n % 2) == 0) {
        int j = (int)uid(g, 1, draws);
        int t = v[i]; v[i] = v[j]; v[j] = t; ++i;
    }
    while (i != n) {
        const uint64_t b0 = (uint64_t)i + 1, b1 = b0 + 1;
        const uint64_t x = uid(g, b0 * b1 - 1, draws);
        int j0 = (int)(x / b1), j1 = (int)(x % b1), t;
        t = v[i]; v[i] = v[j0]; v[j0] = t; ++i;
        t = v[i]; v[i] = v[j1]; v[j1] = t; ++i;
    }
}

/* exposed for the tests of the kernel's own generator */
int hgs_shuffle(int* v, int n, uint32_t seed, int skip_draws) {
    minstd_t g; minstd_seed(&g, seed);
    for (int i = 0; i < skip_draws; ++i) minstd_next(&g);
    long d = 0; shuffle_int(v, n, &g, &d);
    return (int)d;
}

/* ---------------------------------------------------------------------------------------------- state */
typedef struct {
    int nc, R, n;                       /* clients, routes (= vehicles), nodes incl. depot */
    const double* tc;                   /* timeCost [n][n] */
    const double* dem;                  /* demand [n] (depot 0) */
    double cap, penCap;                 /* vehicleCapacity, penaltyCapacityLS */
    /* nodes: 1..nc clients, nc+1+r start depot of route r, nc+1+R+r end depot */
    int *next, *prev, *route, *pos, *whenRI;
    double *cumLoad, *cumTime, *cumRev, *deltaRemoval;
    /* routes */
    int *nbCust, *whenMod, *whenSwapStar, *secStart, *secEnd;
    double *duration, *load, *revDist, *penalty, *polarBary;
    int* polar;                         /* Client::polarAngle */
    const double *cx, *cy;
    int** corr; int* corrLen;           /* correlatedVertices */
    int nbMoves, searchCompleted, loopID;
    /* SWAP* memory: [R][nc+1] */
    int* tbWhen; double* tbCost; int* tbLoc;       /* 3 entries each */
    /* current pair */
    int U, X, V, Y, rU, rV, Up, Xn, Vp, Yn, iU, iX, iV, iY, intra;
    double loadU, loadX, loadV, loadY;
    long evals;
} ls_t;

#define COUR(s, v) ((v) <= (s)->nc ? (v) : 0)
#define ISDEPOT(s, v) ((v) > (s)->nc)
#define DEP(s, r) ((s)->nc + 1 + (r))
#define DEPEND(s, r) ((s)->nc + 1 + (s)->R + (r))
#define TC(s, a, b) ((s)->tc[(size_t)(a) * (s)->n + (b)])

static double pen_load(const ls_t* s, double l) { double e = l - s->cap; return (e > 0. ? e : 0.) * s->penCap; }  /* LocalSearch.h:140 */

static int posmod(int i) { return (i % 65536 + 65536) % 65536; }                                               /* CircleSector.h:14-19 */
static int sec_enclosed(int st, int en, int p) { return posmod(p - st) <= posmod(en - st); }
static int sec_overlap(int s1, int e1, int s2, int e2) {
    return (posmod(s2 - s1) <= posmod(e1 - s1)) || (posmod(s1 - s2) <= posmod(e2 - s2));
}

/* LocalSearch.cpp:652-707 */
static void update_route(ls_t* s, int r) {
    int place = 0; double load = 0., time = 0., rev = 0., cx = 0., cy = 0.;
    int node = DEP(s, r);
    s->pos[node] = 0; s->cumLoad[node] = 0.; s->cumTime[node] = 0.; s->cumRev[node] = 0.;
    int first = 1;
    while (!ISDEPOT(s, node) || first) {
        node = s->next[node];
        ++place;
        s->pos[node] = place;
        const int c = COUR(s, node), pc = COUR(s, s->prev[node]);
        load += s->dem[c];
        time += TC(s, pc, c) + 0.;                               /* + serviceDuration (zeros, swapstar.py:204) */
        rev += TC(s, c, pc) - TC(s, pc, c);
        s->cumLoad[node] = load; s->cumTime[node] = time; s->cumRev[node] = rev;
        if (!ISDEPOT(s, node)) {
            cx += s->cx[c]; cy += s->cy[c];
            if (first) { s->secStart[r] = s->secEnd[r] = s->polar[c]; }
            else if (!sec_enclosed(s->secStart[r], s->secEnd[r], s->polar[c])) {
                if (posmod(s->polar[c] - s->secEnd[r]) <= posmod(s->secStart[r] - s->polar[c])) s->secEnd[r] = s->polar[c];
                else s->secStart[r] = s->polar[c];
            }
        }
        first = 0;
    }
    s->duration[r] = time; s->load[r] = load;
    s->penalty[r] = 0. + pen_load(s, load);                      /* penaltyExcessDuration is 0 * 10: durationLimit = DBL_MAX */
    s->nbCust[r] = place - 1;
    s->revDist[r] = rev;
    s->whenMod[r] = s->nbMoves;
    if (s->nbCust[r] == 0) s->polarBary[r] = 1.e30;
    else s->polarBary[r] = atan2(cy / (double)s->nbCust[r] - s->cy[0], cx / (double)s->nbCust[r] - s->cx[0]);
}

static int first_empty_route(const ls_t* s) {                   /* *emptyRoutes.begin(): std::set<int>, smallest index */
    for (int r = 0; r < s->R; ++r) if (s->nbCust[r] == 0) return r;
    return -1;
}

static void insert_node(ls_t* s, int U, int V) {                /* LocalSearch.cpp:617-626 */
    s->next[s->prev[U]] = s->next[U];
    s->prev[s->next[U]] = s->prev[U];
    s->prev[s->next[V]] = U;
    s->prev[U] = V;
    s->next[U] = s->next[V];
    s->next[V] = U;
    s->route[U] = s->route[V];
}

static void swap_node(ls_t* s, int U, int V) {                  /* LocalSearch.cpp:628-650 */
    int vp = s->prev[V], vn = s->next[V], up = s->prev[U], un = s->next[U], ru = s->route[U], rv = s->route[V];
    s->next[up] = V; s->prev[un] = V; s->next[vp] = U; s->prev[vn] = U;
    s->prev[U] = vp; s->next[U] = vn; s->prev[V] = up; s->next[V] = un;
    s->route[U] = rv; s->route[V] = ru;
}

static void set_u(ls_t* s) {                                    /* LocalSearch.cpp:105-117 */
    s->rU = s->route[s->U]; s->X = s->next[s->U];
    s->Xn = COUR(s, s->next[s->X]); s->iU = COUR(s, s->U); s->Up = COUR(s, s->prev[s->U]); s->iX = COUR(s, s->X);
    s->loadU = s->dem[s->iU]; s->loadX = s->dem[s->iX];
}
static void set_v(ls_t* s) {                                    /* LocalSearch.cpp:119-132 */
    s->rV = s->route[s->V]; s->Y = s->next[s->V];
    s->Yn = COUR(s, s->next[s->Y]); s->iV = COUR(s, s->V); s->Vp = COUR(s, s->prev[s->V]); s->iY = COUR(s, s->Y);
    s->loadV = s->dem[s->iV]; s->loadY = s->dem[s->iY];
    s->intra = (s->rU == s->rV);
}

static void applied(ls_t* s, int both) {
    s->nbMoves++; s->searchCompleted = 0;
    update_route(s, s->rU);
    if (both) update_route(s, s->rV);
}

/* the duration terms of LocalSearch.cpp:134-345 are penaltyExcessDuration(...) = max(0, . - DBL_MAX) * 10 = +0.0: omitted
 * (x + 0.0 == x for every x the sums can take) */
static int move1(ls_t* s) {                                     /* LocalSearch.cpp:134-162 */
    double cU = TC(s, s->Up, s->iX) - TC(s, s->Up, s->iU) - TC(s, s->iU, s->iX);
    double cV = TC(s, s->iV, s->iU) + TC(s, s->iU, s->iY) - TC(s, s->iV, s->iY);
    if (!s->intra) {
        if (cU + cV >= s->penalty[s->rU] + s->penalty[s->rV]) return 0;
        cU += pen_load(s, s->load[s->rU] - s->loadU) - s->penalty[s->rU];
        cV += pen_load(s, s->load[s->rV] + s->loadU) - s->penalty[s->rV];
    }
    if (cU + cV > -HGS_EPS) return 0;
    if (s->iU == s->iY) return 0;
    insert_node(s, s->U, s->V);
    applied(s, !s->intra);
    return 1;
}
static int move2(ls_t* s) {                                     /* LocalSearch.cpp:164-193 */
    double cU = TC(s, s->Up, s->Xn) - TC(s, s->Up, s->iU) - TC(s, s->iX, s->Xn);
    double cV = TC(s, s->iV, s->iU) + TC(s, s->iX, s->iY) - TC(s, s->iV, s->iY);
    if (!s->intra) {
        if (cU + cV >= s->penalty[s->rU] + s->penalty[s->rV]) return 0;
        cU += pen_load(s, s->load[s->rU] - s->loadU - s->loadX) - s->penalty[s->rU];
        cV += pen_load(s, s->load[s->rV] + s->loadU + s->loadX) - s->penalty[s->rV];
    }
    if (cU + cV > -HGS_EPS) return 0;
    if (s->U == s->Y || s->V == s->X || ISDEPOT(s, s->X)) return 0;
    insert_node(s, s->U, s->V);
    insert_node(s, s->X, s->U);
    applied(s, !s->intra);
    return 1;
}
static int move3(ls_t* s) {                                     /* LocalSearch.cpp:195-224 */
    double cU = TC(s, s->Up, s->Xn) - TC(s, s->Up, s->iU) - TC(s, s->iU, s->iX) - TC(s, s->iX, s->Xn);
    double cV = TC(s, s->iV, s->iX) + TC(s, s->iX, s->iU) + TC(s, s->iU, s->iY) - TC(s, s->iV, s->iY);
    if (!s->intra) {
        if (cU + cV >= s->penalty[s->rU] + s->penalty[s->rV]) return 0;
        cU += pen_load(s, s->load[s->rU] - s->loadU - s->loadX) - s->penalty[s->rU];
        cV += pen_load(s, s->load[s->rV] + s->loadU + s->loadX) - s->penalty[s->rV];
    }
    if (cU + cV > -HGS_EPS) return 0;
    if (s->U == s->Y || s->X == s->V || ISDEPOT(s, s->X)) return 0;
    insert_node(s, s->X, s->V);
    insert_node(s, s->U, s->X);
    applied(s, !s->intra);
    return 1;
}
static int move4(ls_t* s) {                                     /* LocalSearch.cpp:226-254 */
    double cU = TC(s, s->Up, s->iV) + TC(s, s->iV, s->iX) - TC(s, s->Up, s->iU) - TC(s, s->iU, s->iX);
    double cV = TC(s, s->Vp, s->iU) + TC(s, s->iU, s->iY) - TC(s, s->Vp, s->iV) - TC(s, s->iV, s->iY);
    if (!s->intra) {
        if (cU + cV >= s->penalty[s->rU] + s->penalty[s->rV]) return 0;
        cU += pen_load(s, s->load[s->rU] + s->loadV - s->loadU) - s->penalty[s->rU];
        cV += pen_load(s, s->load[s->rV] + s->loadU - s->loadV) - s->penalty[s->rV];
    }
    if (cU + cV > -HGS_EPS) return 0;
    if (s->iU == s->Vp || s->iU == s->iY) return 0;
    swap_node(s, s->U, s->V);
    applied(s, !s->intra);
    return 1;
}
static int move5(ls_t* s) {                                     /* LocalSearch.cpp:256-285 */
    double cU = TC(s, s->Up, s->iV) + TC(s, s->iV, s->Xn) - TC(s, s->Up, s->iU) - TC(s, s->iX, s->Xn);
    double cV = TC(s, s->Vp, s->iU) + TC(s, s->iX, s->iY) - TC(s, s->Vp, s->iV) - TC(s, s->iV, s->iY);
    if (!s->intra) {
        if (cU + cV >= s->penalty[s->rU] + s->penalty[s->rV]) return 0;
        cU += pen_load(s, s->load[s->rU] + s->loadV - s->loadU - s->loadX) - s->penalty[s->rU];
        cV += pen_load(s, s->load[s->rV] + s->loadU + s->loadX - s->loadV) - s->penalty[s->rV];
    }
    if (cU + cV > -HGS_EPS) return 0;
    if (s->U == s->prev[s->V] || s->X == s->prev[s->V] || s->U == s->Y || ISDEPOT(s, s->X)) return 0;
    swap_node(s, s->U, s->V);
    insert_node(s, s->X, s->U);
    applied(s, !s->intra);
    return 1;
}
static int move6(ls_t* s) {                                     /* LocalSearch.cpp:287-316 */
    double cU = TC(s, s->Up, s->iV) + TC(s, s->iY, s->Xn) - TC(s, s->Up, s->iU) - TC(s, s->iX, s->Xn);
    double cV = TC(s, s->Vp, s->iU) + TC(s, s->iX, s->Yn) - TC(s, s->Vp, s->iV) - TC(s, s->iY, s->Yn);
    if (!s->intra) {
        if (cU + cV >= s->penalty[s->rU] + s->penalty[s->rV]) return 0;
        cU += pen_load(s, s->load[s->rU] + s->loadV + s->loadY - s->loadU - s->loadX) - s->penalty[s->rU];
        cV += pen_load(s, s->load[s->rV] + s->loadU + s->loadX - s->loadV - s->loadY) - s->penalty[s->rV];
    }
    if (cU + cV > -HGS_EPS) return 0;
    if (ISDEPOT(s, s->X) || ISDEPOT(s, s->Y) || s->Y == s->prev[s->U] || s->U == s->Y || s->X == s->V || s->V == s->next[s->X]) return 0;
    swap_node(s, s->U, s->V);
    swap_node(s, s->X, s->Y);
    applied(s, !s->intra);
    return 1;
}
static int move7(ls_t* s) {                                     /* LocalSearch.cpp:318-352: 2-opt inside a route */
    if (s->pos[s->U] > s->pos[s->V]) return 0;
    double cost = TC(s, s->iU, s->iV) + TC(s, s->iX, s->iY) - TC(s, s->iU, s->iX) - TC(s, s->iV, s->iY)
                  + s->cumRev[s->V] - s->cumRev[s->X];
    if (cost > -HGS_EPS) return 0;
    if (s->next[s->U] == s->V) return 0;
    int U = s->U, X = s->X, V = s->V, Y = s->Y;
    int node = s->next[X];
    s->prev[X] = node; s->next[X] = Y;
    while (node != V) {
        int t = s->next[node];
        s->next[node] = s->prev[node]; s->prev[node] = t;
        node = t;
    }
    s->next[V] = s->prev[V]; s->prev[V] = U; s->next[U] = V; s->prev[Y] = X;
    applied(s, 0);
    return 1;
}
static int move8(ls_t* s) {                                     /* LocalSearch.cpp:354-425: 2-opt*, (U,V) and (X,Y) joined */
    double cost = TC(s, s->iU, s->iV) + TC(s, s->iX, s->iY) - TC(s, s->iU, s->iX) - TC(s, s->iV, s->iY)
                  + s->cumRev[s->V] + s->revDist[s->rU] - s->cumRev[s->X] - s->penalty[s->rU] - s->penalty[s->rV];
    if (cost >= 0) return 0;
    cost += pen_load(s, s->cumLoad[s->U] + s->cumLoad[s->V])
          + pen_load(s, s->load[s->rU] + s->load[s->rV] - s->cumLoad[s->U] - s->cumLoad[s->V]);
    if (cost > -HGS_EPS) return 0;
    const int rU = s->rU, rV = s->rV, U = s->U, X = s->X, V = s->V, Y = s->Y;
    const int depU = DEP(s, rU), depV = DEP(s, rV), depUFin = s->prev[depU], depVFin = s->prev[depV], depVNext = s->next[depV];
    int xx = X, vv = V, t;
    while (!ISDEPOT(s, xx)) { t = s->next[xx]; s->next[xx] = s->prev[xx]; s->prev[xx] = t; s->route[xx] = rV; xx = t; }
    while (!ISDEPOT(s, vv)) { t = s->prev[vv]; s->prev[vv] = s->next[vv]; s->next[vv] = t; s->route[vv] = rU; vv = t; }
    s->next[U] = V; s->prev[V] = U; s->next[X] = Y; s->prev[Y] = X;
    if (ISDEPOT(s, X)) {
        s->next[depUFin] = depU; s->prev[depUFin] = depVNext; s->next[s->prev[depUFin]] = depUFin;
        s->next[depV] = Y; s->prev[Y] = depV;
    } else if (ISDEPOT(s, V)) {
        s->next[depV] = s->prev[depUFin]; s->prev[s->next[depV]] = depV; s->prev[depV] = depVFin;
        s->prev[depUFin] = U; s->next[U] = depUFin;
    } else {
        s->next[depV] = s->prev[depUFin]; s->prev[s->next[depV]] = depV;
        s->prev[depUFin] = depVNext; s->next[s->prev[depUFin]] = depUFin;
    }
    applied(s, 1);
    return 1;
}
static int move9(ls_t* s) {                                     /* LocalSearch.cpp:427-484: 2-opt*, tails exchanged */
    double cost = TC(s, s->iU, s->iY) + TC(s, s->iV, s->iX) - TC(s, s->iU, s->iX) - TC(s, s->iV, s->iY)
                  - s->penalty[s->rU] - s->penalty[s->rV];
    if (cost >= 0) return 0;
    cost += pen_load(s, s->cumLoad[s->U] + s->load[s->rV] - s->cumLoad[s->V])
          + pen_load(s, s->cumLoad[s->V] + s->load[s->rU] - s->cumLoad[s->U]);
    if (cost > -HGS_EPS) return 0;
    const int rU = s->rU, rV = s->rV, U = s->U, X = s->X, V = s->V, Y = s->Y;
    const int depU = DEP(s, rU), depV = DEP(s, rV), depUFin = s->prev[depU], depVFin = s->prev[depV], depUpred = s->prev[depUFin];
    int c = Y;
    while (!ISDEPOT(s, c)) { s->route[c] = rU; c = s->next[c]; }
    c = X;
    while (!ISDEPOT(s, c)) { s->route[c] = rV; c = s->next[c]; }
    s->next[U] = Y; s->prev[Y] = U; s->next[V] = X; s->prev[X] = V;
    if (ISDEPOT(s, X)) {
        s->prev[depUFin] = s->prev[depVFin]; s->next[s->prev[depUFin]] = depUFin;
        s->next[V] = depVFin; s->prev[depVFin] = V;
    } else {
        s->prev[depUFin] = s->prev[depVFin]; s->next[s->prev[depUFin]] = depUFin;
        s->prev[depVFin] = depUpred; s->next[s->prev[depVFin]] = depVFin;
    }
    applied(s, 1);
    return 1;
}

/* ---------------------------------------------------------------------------------------------- SWAP* */
#define TB(s, r, c) (((size_t)(r) * ((s)->nc + 1) + (c)))
static void tb_reset(ls_t* s, size_t k) { for (int j = 0; j < 3; ++j) { s->tbCost[3 * k + j] = 1.e30; s->tbLoc[3 * k + j] = -1; } }
static void tb_add(ls_t* s, size_t k, double c, int loc) {      /* LocalSearch.h:69-89 */
    double* bc = s->tbCost + 3 * k; int* bl = s->tbLoc + 3 * k;
    if (c >= bc[2]) return;
    else if (c >= bc[1]) { bc[2] = c; bl[2] = loc; }
    else if (c >= bc[0]) { bc[2] = bc[1]; bl[2] = bl[1]; bc[1] = c; bl[1] = loc; }
    else { bc[2] = bc[1]; bl[2] = bl[1]; bc[1] = bc[0]; bl[1] = bl[0]; bc[0] = c; bl[0] = loc; }
}
static void preprocess_insertions(ls_t* s, int r1, int r2) {    /* LocalSearch.cpp:594-615 */
    for (int U = s->next[DEP(s, r1)]; !ISDEPOT(s, U); U = s->next[U]) {
        const int u = U, up = COUR(s, s->prev[U]), un = COUR(s, s->next[U]);
        s->deltaRemoval[U] = TC(s, up, un) - TC(s, up, u) - TC(s, u, un);
        const size_t k = TB(s, r2, u);
        if (s->whenMod[r2] > s->tbWhen[k]) {
            tb_reset(s, k);
            s->tbWhen[k] = s->nbMoves;
            const int f = COUR(s, s->next[DEP(s, r2)]);
            s->tbCost[3 * k] = TC(s, 0, u) + TC(s, u, f) - TC(s, 0, f);
            s->tbLoc[3 * k] = DEP(s, r2);
            for (int V = s->next[DEP(s, r2)]; !ISDEPOT(s, V); V = s->next[V]) {
                const int vn = COUR(s, s->next[V]);
                tb_add(s, k, TC(s, V, u) + TC(s, u, vn) - TC(s, V, vn), V);
            }
        }
    }
}
static double cheapest_insert_simult_removal(ls_t* s, int U, int V, int* bestPos) {     /* LocalSearch.cpp:575-592 */
    const size_t k = TB(s, s->route[V], U);
    const double* bc = s->tbCost + 3 * k; const int* bl = s->tbLoc + 3 * k;
    *bestPos = bl[0];
    double best = bc[0];
    int found = (*bestPos != V && s->next[*bestPos] != V);
    if (!found && bl[1] != -1) {
        *bestPos = bl[1]; best = bc[1];
        found = (*bestPos != V && s->next[*bestPos] != V);
        if (!found && bl[2] != -1) { *bestPos = bl[2]; best = bc[2]; found = 1; }
    }
    const int vp = COUR(s, s->prev[V]), vn = COUR(s, s->next[V]);
    const double d = TC(s, vp, U) + TC(s, U, vn) - TC(s, vp, vn);
    if (!found || d < best) { *bestPos = s->prev[V]; best = d; }
    return best;
}
static int swap_star(ls_t* s, int rU, int rV) {                 /* LocalSearch.cpp:486-573 */
    double bestCost = 1.e30; int bU = -1, bV = -1, bPU = -1, bPV = -1;
    preprocess_insertions(s, rU, rV);
    preprocess_insertions(s, rV, rU);
    for (int U = s->next[DEP(s, rU)]; !ISDEPOT(s, U); U = s->next[U])
        for (int V = s->next[DEP(s, rV)]; !ISDEPOT(s, V); V = s->next[V]) {
            const double dPU = pen_load(s, s->load[rU] + s->dem[V] - s->dem[U]) - s->penalty[rU];
            const double dPV = pen_load(s, s->load[rV] + s->dem[U] - s->dem[V]) - s->penalty[rV];
            if (dPU + s->deltaRemoval[U] + dPV + s->deltaRemoval[V] <= 0) {
                int pU, pV;
                const double extraV = cheapest_insert_simult_removal(s, U, V, &pU);
                const double extraU = cheapest_insert_simult_removal(s, V, U, &pV);
                const double mc = dPU + s->deltaRemoval[U] + extraU + dPV + s->deltaRemoval[V] + extraV + 0. + 0.;
                if (mc < bestCost) { bestCost = mc; bU = U; bV = V; bPU = pU; bPV = pV; }
            }
        }
    for (int U = s->next[DEP(s, rU)]; !ISDEPOT(s, U); U = s->next[U]) {
        const size_t k = TB(s, rV, U);
        const int up = COUR(s, s->prev[U]), un = COUR(s, s->next[U]);
        const double dU = TC(s, up, un) - TC(s, up, U) - TC(s, U, un);
        const double dV = s->tbCost[3 * k];
        const double mc = dU + dV + pen_load(s, s->load[rU] - s->dem[U]) - s->penalty[rU]
                        + pen_load(s, s->load[rV] + s->dem[U]) - s->penalty[rV] + 0. + 0.;
        if (mc < bestCost) { bestCost = mc; bU = U; bPU = s->tbLoc[3 * k]; bV = -1; bPV = -1; }
    }
    for (int V = s->next[DEP(s, rV)]; !ISDEPOT(s, V); V = s->next[V]) {
        const size_t k = TB(s, rU, V);
        const int vp = COUR(s, s->prev[V]), vn = COUR(s, s->next[V]);
        const double dU = s->tbCost[3 * k];
        const double dV = TC(s, vp, vn) - TC(s, vp, V) - TC(s, V, vn);
        const double mc = dU + dV + pen_load(s, s->load[rU] + s->dem[V]) - s->penalty[rU]
                        + pen_load(s, s->load[rV] - s->dem[V]) - s->penalty[rV] + 0. + 0.;
        if (mc < bestCost) { bestCost = mc; bV = V; bPV = s->tbLoc[3 * k]; bU = -1; bPU = -1; }
    }
    if (bestCost > -HGS_EPS) return 0;
    if (bPU != -1) insert_node(s, bU, bPU);
    if (bPV != -1) insert_node(s, bV, bPV);
    s->nbMoves++; s->searchCompleted = 0;
    update_route(s, rU); update_route(s, rV);
    return 1;
}

/* ---------------------------------------------------------------------------------------------- driver */
typedef struct { double d; int j; } prox_t;
static int prox_cmp(const void* a, const void* b) {            /* std::pair<double,int> operator< */
    const prox_t *x = a, *y = b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return (x->j > y->j) - (x->j < y->j);
}
typedef struct { double a; int r; } ang_t;
static int ang_cmp(const void* a, const void* b) {
    const ang_t *x = a, *y = b;
    if (x->a < y->a) return -1;
    if (x->a > y->a) return 1;
    return (x->r > y->r) - (x->r < y->r);
}

/* Correlated vertices (Params.cpp:80-103): for each client the nb_granular nearest by timeCost[i][.] (ties: smaller index),
 * made symmetric, ascending.  out_len[i] (i = 0..nc; 0 unused), out [ (nc+1) * nc ] row i at out + i * nc. */
void hgs_correlated(int n, const double* tc, int nb_granular, int* out, int* out_len) {
    const int nc = n - 1;
    unsigned char* m = calloc((size_t)(nc + 1) * (nc + 1), 1);
    prox_t* p = malloc(sizeof(prox_t) * (nc > 0 ? nc : 1));
    for (int i = 1; i <= nc; ++i) {
        int k = 0;
        for (int j = 1; j <= nc; ++j) if (i != j) { p[k].d = tc[(size_t)i * n + j]; p[k].j = j; ++k; }
        qsort(p, k, sizeof(prox_t), prox_cmp);
        const int g = nb_granular < nc - 1 ? nb_granular : nc - 1;
        for (int j = 0; j < g; ++j) { m[(size_t)i * (nc + 1) + p[j].j] = 1; m[(size_t)p[j].j * (nc + 1) + i] = 1; }
    }
    out_len[0] = 0;
    for (int i = 1; i <= nc; ++i) {
        int k = 0;
        for (int j = 1; j <= nc; ++j) if (m[(size_t)i * (nc + 1) + j]) out[(size_t)i * nc + k++] = j;
        out_len[i] = k;
    }
    free(p); free(m);
}

/* One call of the reference's local_search (C_Interface.cpp:128-172) on ONE solution.
 *   n nodes (depot 0), xs / ys coordinates, tc [n][n] f64, dem [n] f64 AS HGS GETS THEM (swapstar.py:335: demands * 1000),
 *   cap (1000.001), seq_in: the solution as a zero-separated node sequence of len_in entries (a column of `paths`),
 *   count: the loop bound of LocalSearch::run, nb_granular (20), seed (the value HGS reads: 1; 0 gives the same stream).
 *   seq_out [out_len]: "0 c1 .. ck" per non-empty route in export order, zero padded (cvrp_nls/aco.py:22-33 merge_subroutes).
 *   stats (may be NULL): [0] moves applied, [1] loops run, [2] RNG draws, [3] (U,V) pairs evaluated.
 * Returns 0; 1 when HGS throws (distances / demands out of scale, Params.cpp:106-114; infeasible or incomplete input,
 * Individual.cpp:68-71; fleet too small) -- the Python side then keeps the input routes (swapstar.py:262-271, :341-345),
 * which is what seq_out holds; -1 bad argument. */
int hgs_local_search(int n, const double* xs, const double* ys, const double* tc, const double* dem, double cap,
                     const int* seq_in, int len_in, int count, int nb_granular, uint32_t seed, int use_swap_star,
                     int* seq_out, int out_len, long* stats) {
    const int nc = n - 1;
    if (n < 2 || len_in < 1 || out_len < 1) return -1;
    /* routes of the input (cvrp_nls/aco.py:12-20 get_subroutes: non-empty pieces between zeros) */
    int R = 0;
    int* rstart = malloc(sizeof(int) * (len_in + 1)); int* rlen = malloc(sizeof(int) * (len_in + 1));
    for (int i = 0; i < len_in;) {
        if (seq_in[i] == 0) { ++i; continue; }
        int j = i; while (j < len_in && seq_in[j] != 0) ++j;
        rstart[R] = i; rlen[R] = j - i; ++R; i = j;
    }
    /* the unchanged answer, should HGS throw */
    { int k = 0; memset(seq_out, 0, sizeof(int) * out_len);
      for (int r = 0; r < R; ++r) { if (k + 1 + rlen[r] > out_len) { free(rstart); free(rlen); return -1; }
                                    seq_out[k++] = 0; for (int i = 0; i < rlen[r]; ++i) seq_out[k++] = seq_in[rstart[r] + i]; } }
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;

    /* Params (Params.cpp:25-121) */
    double totalDemand = 0., maxDemand = 0., maxDist = 0.;
    for (int i = 0; i <= nc; ++i) { if (dem[i] > maxDemand) maxDemand = dem[i]; totalDemand += dem[i]; }
    for (int i = 0; i <= nc; ++i) for (int j = 0; j <= nc; ++j) if (tc[(size_t)i * n + j] > maxDist) maxDist = tc[(size_t)i * n + j];
    int thrown = (maxDist < 0.1 || maxDist > 100000) || (maxDemand < 0.1 || maxDemand > 100000) || (R < ceil(totalDemand / cap));
    /* Individual(params, file) (Individual.cpp:38-82): complete and feasible, or it throws */
    {
        int cnt = 0; unsigned char* seen = calloc(nc + 1, 1);
        for (int r = 0; r < R && !thrown; ++r) {
            double load = 0.;
            for (int i = 0; i < rlen[r]; ++i) {
                int c = seq_in[rstart[r] + i];
                if (c < 1 || c > nc || seen[c]) { thrown = 1; break; }
                seen[c] = 1; ++cnt; load += dem[c];
            }
            /* capacityExcess < MY_EPSILON, summed over routes (Individual.cpp:26,32): one route over by >= 1e-5 is enough */
            if (load > cap && load - cap >= HGS_EPS) thrown = 1;
        }
        free(seen);
        if (cnt != nc) thrown = 1;
    }
    if (thrown) { free(rstart); free(rlen); return 1; }
    /* (sum of several sub-epsilon excesses reaching 1e-5 is not representable with demands k * 1000 / capacity: ignored) */

    ls_t S; ls_t* s = &S; memset(s, 0, sizeof(S));
    s->nc = nc; s->R = R; s->n = n; s->tc = tc; s->dem = dem; s->cap = cap;
    /* Params.cpp:40-54: without SWAP* the coordinates are NOT taken over (all clients at (0, 0), polar angle 0): every
     * non-empty route's barycentre angle is atan2(0, 0) = 0 and the export order below is the route index order */
    double* zeros = calloc(nc + 1, sizeof(double));
    s->cx = use_swap_star ? xs : zeros; s->cy = use_swap_star ? ys : zeros;
    xs = s->cx; ys = s->cy;
    double penaltyCapacity = maxDist / maxDemand; if (penaltyCapacity > 1000.) penaltyCapacity = 1000.; if (penaltyCapacity < 0.1) penaltyCapacity = 0.1;
    s->penCap = penaltyCapacity * 10.;
    const int N = nc + 1 + 2 * R;
    s->next = malloc(sizeof(int) * N); s->prev = malloc(sizeof(int) * N); s->route = malloc(sizeof(int) * N);
    s->pos = malloc(sizeof(int) * N); s->whenRI = malloc(sizeof(int) * N);
    s->cumLoad = malloc(sizeof(double) * N); s->cumTime = malloc(sizeof(double) * N); s->cumRev = malloc(sizeof(double) * N);
    s->deltaRemoval = malloc(sizeof(double) * N);
    s->nbCust = malloc(sizeof(int) * R); s->whenMod = malloc(sizeof(int) * R); s->whenSwapStar = malloc(sizeof(int) * R);
    s->secStart = malloc(sizeof(int) * R); s->secEnd = malloc(sizeof(int) * R);
    s->duration = malloc(sizeof(double) * R); s->load = malloc(sizeof(double) * R); s->revDist = malloc(sizeof(double) * R);
    s->penalty = malloc(sizeof(double) * R); s->polarBary = malloc(sizeof(double) * R);
    s->polar = malloc(sizeof(int) * (nc + 1));
    for (int i = 0; i <= nc; ++i)                               /* Params.cpp:42-47 (double -> int conversion truncates) */
        s->polar[i] = posmod((int)(32768. * atan2(ys[i] - ys[0], xs[i] - xs[0]) / HGS_PI));
    int* corrBuf = malloc(sizeof(int) * (size_t)(nc + 1) * (nc > 0 ? nc : 1));
    s->corrLen = malloc(sizeof(int) * (nc + 1)); s->corr = malloc(sizeof(int*) * (nc + 1));
    hgs_correlated(n, tc, nb_granular, corrBuf, s->corrLen);
    for (int i = 0; i <= nc; ++i) s->corr[i] = corrBuf + (size_t)i * nc;
    if (use_swap_star) {
        const size_t T = (size_t)R * (nc + 1);
        s->tbWhen = malloc(sizeof(int) * T); s->tbCost = malloc(sizeof(double) * 3 * T); s->tbLoc = malloc(sizeof(int) * 3 * T);
        for (size_t k = 0; k < T; ++k) tb_reset(s, k);
    }

    minstd_t g; minstd_seed(&g, seed);
    long draws = 0;
    /* Individual(params) shuffles a client permutation it then discards (Individual.cpp:30-35, :38) */
    int* tmp = malloc(sizeof(int) * (nc > R ? nc : R) + sizeof(int));
    for (int i = 0; i < nc; ++i) tmp[i] = i + 1;
    shuffle_int(tmp, nc, &g, &draws);

    /* loadIndividual (LocalSearch.cpp:709-754) */
    s->nbMoves = 0;
    for (int r = 0; r < R; ++r) {
        const int d0 = DEP(s, r), d1 = DEPEND(s, r);
        s->route[d0] = s->route[d1] = r;
        s->prev[d0] = d1; s->next[d1] = d0;
        int p = d0;
        for (int i = 0; i < rlen[r]; ++i) { int c = seq_in[rstart[r] + i]; s->route[c] = r; s->prev[c] = p; s->next[p] = c; p = c; }
        s->next[p] = d1; s->prev[d1] = p;
        update_route(s, r);
        s->whenSwapStar[r] = -1;
        if (use_swap_star) for (int i = 1; i <= nc; ++i) s->tbWhen[TB(s, r, i)] = -1;
    }
    for (int i = 1; i <= nc; ++i) s->whenRI[i] = -1;

    /* LocalSearch::run (LocalSearch.cpp:3-103) */
    int* orderNodes = malloc(sizeof(int) * (nc + 1)); int* orderRoutes = malloc(sizeof(int) * (R + 1));
    for (int i = 0; i < nc; ++i) orderNodes[i] = i + 1;
    for (int r = 0; r < R; ++r) orderRoutes[r] = r;
    shuffle_int(orderNodes, nc, &g, &draws);
    shuffle_int(orderRoutes, R, &g, &draws);
    for (int i = 1; i <= nc; ++i) {
        ++draws;
        if (minstd_next(&g) % (uint64_t)nb_granular == 0) shuffle_int(s->corr[i], s->corrLen[i], &g, &draws);
    }
    s->searchCompleted = 0;
    int loops = 0;
    for (s->loopID = 0; !s->searchCompleted && s->loopID <= count; s->loopID++) {
        ++loops;
        if (s->loopID > 1) s->searchCompleted = 1;
        for (int posU = 0; posU < nc; ++posU) {
            s->U = orderNodes[posU];
            const int lastTest = s->whenRI[s->U];
            s->whenRI[s->U] = s->nbMoves;
            const int cu = s->U;
            for (int posV = 0; posV < s->corrLen[cu]; ++posV) {
                s->V = s->corr[cu][posV];
                const int wu = s->whenMod[s->route[s->U]], wv = s->whenMod[s->route[s->V]];
                if (s->loopID == 0 || (wu > wv ? wu : wv) > lastTest) {
                    ++s->evals;
                    set_u(s); set_v(s);
                    if (move1(s)) continue;
                    if (move2(s)) continue;
                    if (move3(s)) continue;
                    if (s->iU <= s->iV && move4(s)) continue;
                    if (move5(s)) continue;
                    if (s->iU <= s->iV && move6(s)) continue;
                    if (s->intra && move7(s)) continue;
                    if (!s->intra && move8(s)) continue;
                    if (!s->intra && move9(s)) continue;
                    if (ISDEPOT(s, s->prev[s->V])) {
                        s->V = s->prev[s->V];
                        set_v(s);
                        if (move1(s)) continue;
                        if (move2(s)) continue;
                        if (move3(s)) continue;
                        if (!s->intra && move8(s)) continue;
                        if (!s->intra && move9(s)) continue;
                    }
                }
            }
            if (s->loopID > 0) {
                const int er = first_empty_route(s);
                if (er >= 0) {
                    s->V = DEP(s, er);
                    set_u(s); set_v(s);
                    if (move1(s)) continue;
                    if (move2(s)) continue;
                    if (move3(s)) continue;
                    if (move9(s)) continue;
                }
            }
        }
        if (use_swap_star) {                                    /* areCoordinatesProvided: positions are always given */
            for (int a = 0; a < R; ++a) {
                const int rU = orderRoutes[a];
                const int lastSS = s->whenSwapStar[rU];
                s->whenSwapStar[rU] = s->nbMoves;
                for (int b = 0; b < R; ++b) {
                    const int rV = orderRoutes[b];
                    if (s->nbCust[rU] > 0 && s->nbCust[rV] > 0 && rU < rV
                        && (s->loopID == 0 || (s->whenMod[rU] > s->whenMod[rV] ? s->whenMod[rU] : s->whenMod[rV]) > lastSS))
                        if (sec_overlap(s->secStart[rU], s->secEnd[rU], s->secStart[rV], s->secEnd[rV]))
                            swap_star(s, rU, rV);
                }
            }
        }
    }

    /* exportIndividual (LocalSearch.cpp:756-778) + exportCVRPLibFormat (non-empty routes, in that order) */
    ang_t* ang = malloc(sizeof(ang_t) * (R > 0 ? R : 1));
    for (int r = 0; r < R; ++r) { ang[r].a = s->polarBary[r]; ang[r].r = r; }
    qsort(ang, R, sizeof(ang_t), ang_cmp);
    int k = 0; memset(seq_out, 0, sizeof(int) * out_len);
    for (int a = 0; a < R; ++a) {
        int node = s->next[DEP(s, ang[a].r)];
        if (ISDEPOT(s, node)) continue;
        seq_out[k++] = 0;
        while (!ISDEPOT(s, node)) { seq_out[k++] = node; node = s->next[node]; }
    }
    if (stats) { stats[0] = s->nbMoves; stats[1] = loops; stats[2] = draws; stats[3] = s->evals; }

    free(ang); free(orderNodes); free(orderRoutes); free(tmp); free(corrBuf); free(s->corrLen); free(s->corr);
    free(s->next); free(s->prev); free(s->route); free(s->pos); free(s->whenRI); free(s->cumLoad); free(s->cumTime);
    free(s->cumRev); free(s->deltaRemoval); free(s->nbCust); free(s->whenMod); free(s->whenSwapStar); free(s->secStart);
    free(s->secEnd); free(s->duration); free(s->load); free(s->revDist); free(s->penalty); free(s->polarBary); free(s->polar);
    if (use_swap_star) { free(s->tbWhen); free(s->tbCost); free(s->tbLoc); }
    free(rstart); free(rlen); free(zeros);
    return 0;
}
