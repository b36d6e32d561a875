/*
 * bader_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A sequential, single-threaded CPU restatement (plain C, IEEE float64, no FMA contraction:
 * build with -ffp-contract=off) of the hot path of pybader v0.3.12 for the `threads in {0,1}` case
 * (one block == the whole grid, idx == (0,0,0); the reference's block-extension branches
 * methods.py:364-409 / refinement.py:155-198 are dead code there and are not restated).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the timed CPU baseline -- never as a fallback of the product path.
 *
 * Parity status: the reference has no tests of its own ("parity unpinned" by the reference);
 * this oracle is pinned against golden vectors captured by importing the reference in the build
 * container (tests/golden/make_golden.py -> tests/golden/ npz files; tests/test_oracle_golden.py).
 *
 * Every function cites the reference file:line it follows.  Arrays are C-order [x][y][z].
 * Labels ("volumes") are int32 here; known flags int8; indices int64.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int32_t lab_t;

typedef struct {
    int64_t n[3];
    const double *rho;
    double dist[27]; /* dist_mat[ix][iy][iz], index 2 == -1 (interface.py:242-259) */
    double T[9];     /* T_grad row-major (interface.py:285-290) */
} grid_t;

static inline int64_t lin(const grid_t *g, const int64_t p[3]) {
    return (p[0] * g->n[1] + p[1]) * g->n[2] + p[2];
}
static inline int64_t wrap1(int64_t v, int64_t n) { /* single wrap, methods.py:90-93 */
    if (v < 0) v += n; else if (v >= n) v -= n;
    return v;
}
/* round-half-away-from-zero via truncating cast, methods.py:347-350 */
static inline int64_t rha(double x) { return x > 0 ? (int64_t)(x + .5) : (int64_t)(x - .5); }

/* One neargrid gradient step from voxel p with carried remainder dr.
 * methods.py:302-363 (strict==0: tie test `<= >=`, line 324) and
 * refinement.py:89-154 (strict==1: tie test `< >`, line 111).
 * Returns 1 when max_grad < 1E-14 (no move, dr untouched), else 0 with pd/dr updated. */
static int ng_step(const grid_t *g, int strict, const int64_t p[3], double dr[3], int64_t pd[3]) {
    double grad[3], gd[3], max_grad = 0.;
    const double c = g->rho[lin(g, p)];
    for (int j = 0; j < 3; j++) {
        int64_t q[3] = {p[0], p[1], p[2]};
        q[j] = wrap1(p[j] + 1, g->n[j]);
        const double hi = g->rho[lin(g, q)];
        q[j] = wrap1(q[j] - 2, g->n[j]); /* methods.py:314-319: (p+1 wrapped) - 2, re-wrapped */
        const double lo = g->rho[lin(g, q)];
        int flat = strict ? (hi < c && c > lo) : (hi <= c && c >= lo);
        grad[j] = flat ? 0. : (hi - lo) / 2.;
    }
    for (int j = 0; j < 3; j++) { /* methods.py:332-339 */
        gd[j] = ((g->T[3 * j] * grad[0]) + (g->T[3 * j + 1] * grad[1])) + (g->T[3 * j + 2] * grad[2]);
        if (gd[j] > max_grad) max_grad = gd[j];
        else if (-gd[j] > max_grad) max_grad = -gd[j];
    }
    if (max_grad < 1E-14) { /* methods.py:341-343 */
        pd[0] = p[0]; pd[1] = p[1]; pd[2] = p[2];
        return 1;
    }
    for (int j = 0; j < 3; j++) { /* methods.py:345-363 */
        gd[j] /= max_grad;
        int64_t ig = rha(gd[j]);
        pd[j] = p[j] + ig;
        dr[j] += gd[j] - (double)ig;
        int64_t id = rha(dr[j]);
        pd[j] += id;
        dr[j] -= (double)id;
        if (pd[j] >= g->n[j]) pd[j] -= g->n[j];
        else if (pd[j] < 0) pd[j] += g->n[j];
    }
    return 0;
}

/* One ongrid step: best of the 27 neighbours, distance weighted, strict '>' first-wins in
 * (ix,iy,iz) ascending order.  methods.py:84-117, 416-447; refinement.py:204-235. */
static void og_step(const grid_t *g, const int64_t p[3], int64_t pd[3]) {
    const double ctr = g->rho[lin(g, p)];
    double max_val = ctr;
    pd[0] = p[0]; pd[1] = p[1]; pd[2] = p[2];
    for (int ix = -1; ix < 2; ix++) {
        int64_t pt[3];
        pt[0] = wrap1(p[0] + ix, g->n[0]);
        for (int iy = -1; iy < 2; iy++) {
            pt[1] = wrap1(p[1] + iy, g->n[1]);
            for (int iz = -1; iz < 2; iz++) {
                pt[2] = wrap1(p[2] + iz, g->n[2]);
                double v = g->rho[lin(g, pt)];
                v = (v - ctr) * g->dist[((ix + 3) % 3) * 9 + ((iy + 3) % 3) * 3 + ((iz + 3) % 3)];
                v += ctr;
                if (v > max_val) {
                    max_val = v;
                    pd[0] = pt[0]; pd[1] = pt[1]; pd[2] = pt[2];
                }
            }
        }
    }
}

typedef struct { int64_t *v; int64_t len, cap; } ivec;
static void ivec_push(ivec *a, int64_t x) {
    if (a->len == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 1024;
        a->v = (int64_t *)realloc(a->v, (size_t)a->cap * sizeof(int64_t));
    }
    a->v[a->len++] = x;
}

static void grid_init(grid_t *g, const double *rho, const int64_t shape[3], const double *dist, const double *T) {
    for (int j = 0; j < 3; j++) g->n[j] = shape[j];
    g->rho = rho;
    if (dist) memcpy(g->dist, dist, sizeof g->dist);
    if (T) memcpy(g->T, T, sizeof g->T);
}

/* ------------------------------------------------------------------------------------------
 * methods.ongrid, single block (methods.py:15-219).  vol: 0 unassigned / -1 vacuum on entry,
 * 1-based local labels on exit.  maxima_out: malloc'ed int64[n_max*3] (free with orc_free).
 * ---------------------------------------------------------------------------------------- */
int64_t orc_ongrid(const double *rho, const int64_t shape[3], lab_t *vol, const double *dist,
                   int64_t **maxima_out) {
    grid_t g; grid_init(&g, rho, shape, dist, NULL);
    const int64_t N = shape[0] * shape[1] * shape[2];
    ivec path = {0}, maxima = {0};
    int64_t bader_num = 0;
    int64_t i[3];
    for (i[0] = 0; i[0] < shape[0]; i[0]++) for (i[1] = 0; i[1] < shape[1]; i[1]++) for (i[2] = 0; i[2] < shape[2]; i[2]++) {
        const int64_t li = lin(&g, i);
        if (vol[li] != 0) continue; /* methods.py:73-74 */
        int64_t p[3] = {i[0], i[1], i[2]}, pd[3];
        path.len = 0; ivec_push(&path, li);
        lab_t vol_num;
        for (;;) {
            og_step(&g, p, pd);
            const int64_t lpd = lin(&g, pd);
            if (vol[lpd] != 0) { vol_num = vol[lpd]; break; }          /* methods.py:166-168 */
            if (pd[0] == p[0] && pd[1] == p[1] && pd[2] == p[2]) { vol_num = 0; break; } /* 169-177 */
            p[0] = pd[0]; p[1] = pd[1]; p[2] = pd[2];
            ivec_push(&path, lpd);
        }
        if (vol_num == 0) { /* methods.py:201-209 */
            ivec_push(&maxima, pd[0]); ivec_push(&maxima, pd[1]); ivec_push(&maxima, pd[2]);
            bader_num++;
            vol_num = (lab_t)bader_num;
        }
        for (int64_t j = 0; j < path.len; j++) vol[path.v[j]] = vol_num; /* methods.py:211-214 */
    }
    (void)N;
    free(path.v);
    *maxima_out = maxima.v;
    return bader_num;
}

/* ------------------------------------------------------------------------------------------
 * methods.neargrid, single block (methods.py:222-611): the scan-order dependent main pass.
 * ---------------------------------------------------------------------------------------- */
static inline int inb(const grid_t *g, const int64_t q[3]) {
    return q[0] >= 0 && q[0] < g->n[0] && q[1] >= 0 && q[1] < g->n[1] && q[2] >= 0 && q[2] < g->n[2];
}
/* methods.py:556-577 / 581-603: q (= path voxel +- e_k, NOT periodically wrapped) becomes known==2
 * when it is labelled (not 0/-1) and its six face neighbours are all in the block and share its label. */
static void known_check(const grid_t *g, const lab_t *vol, int8_t *known, const int64_t q[3]) {
    const lab_t t = vol[lin(g, q)];
    if (t > -2 && t < 1) return; /* not (-2 < vol_temp < 1) */
    for (int h = 0; h < 3; h++) {
        int64_t r[3] = {q[0], q[1], q[2]};
        r[h] = q[h] + 1;
        if (r[h] >= g->n[h] || vol[lin(g, r)] != t) return;
        r[h] = q[h] - 1;
        if (r[h] < 0 || vol[lin(g, r)] != t) return;
    }
    known[lin(g, q)] = 2;
}

int64_t orc_neargrid(const double *rho, const int64_t shape[3], lab_t *vol, const double *dist,
                     const double *T, int64_t **maxima_out) {
    grid_t g; grid_init(&g, rho, shape, dist, T);
    const int64_t N = shape[0] * shape[1] * shape[2];
    int8_t *known = (int8_t *)calloc((size_t)N, 1);
    ivec path = {0}, maxima = {0};
    int64_t bader_num = 0;
    int64_t i[3];
    for (i[0] = 0; i[0] < shape[0]; i[0]++) for (i[1] = 0; i[1] < shape[1]; i[1]++) for (i[2] = 0; i[2] < shape[2]; i[2]++) {
        const int64_t li = lin(&g, i);
        if (vol[li] == -1) continue;      /* methods.py:286-289 */
        if (known[li] == 2) continue;
        known[li] = 1;
        int64_t p[3] = {i[0], i[1], i[2]}, pd[3];
        double dr[3] = {0., 0., 0.};
        path.len = 0; ivec_push(&path, li);
        lab_t vol_num = 0;
        for (;;) {
            ng_step(&g, 0, p, dr, pd);
            int64_t lpv = lin(&g, pd);
            if (known[lpv] == 1) { /* methods.py:411-507: been here on this path */
                dr[0] = dr[1] = dr[2] = 0.;
                og_step(&g, p, pd);
                lpv = lin(&g, pd);
                if (pd[0] == p[0] && pd[1] == p[1] && pd[2] == p[2]) { /* break_flag */
                    vol_num = 0;
                    if (vol[lpv] != 0) vol_num = vol[lpv];
                    break;
                }
            }
            if (known[lpv] == 2) { vol_num = vol[lpv]; break; } /* methods.py:509-511 */
            p[0] = pd[0]; p[1] = pd[1]; p[2] = pd[2];              /* methods.py:513-521 */
            ivec_push(&path, lpv);
            known[lpv] = 1;
        }
        if (vol_num == 0) { /* methods.py:533-541 */
            ivec_push(&maxima, pd[0]); ivec_push(&maxima, pd[1]); ivec_push(&maxima, pd[2]);
            bader_num++;
            vol_num = (lab_t)bader_num;
        }
        for (int64_t j = 0; j < path.len; j++) { /* methods.py:543-606 */
            const int64_t lp = path.v[j];
            int64_t q[3];
            q[0] = lp / (shape[1] * shape[2]);
            q[1] = (lp / shape[2]) % shape[1];
            q[2] = lp % shape[2];
            vol[lp] = vol_num;
            if (known[lp] != 2) known[lp] = 0;
            for (int k = 0; k < 3; k++) {
                q[k] += 1;
                if (q[k] < g.n[k]) known_check(&g, vol, known, q);
                q[k] -= 2;
                if (q[k] >= 0) known_check(&g, vol, known, q);
                q[k] += 1;
            }
        }
    }
    free(known); free(path.v);
    *maxima_out = maxima.v;
    return bader_num;
}

/* ------------------------------------------------------------------------------------------
 * refinement.edge_find (refinement.py:326-405), in-place sequential sweep.
 * ---------------------------------------------------------------------------------------- */
static void classify(const grid_t *g, const lab_t *vol, const int64_t i[3], int *is_edge, int *is_max) {
    /* refinement.py:345-375 and 446-476 (identical 27-point test) */
    const lab_t vol_num = vol[lin(g, i)];
    const double max_val = g->rho[lin(g, i)];
    *is_max = 1; *is_edge = 0;
    int64_t p[3];
    for (int ix = -1; ix < 2; ix++) {
        p[0] = wrap1(i[0] + ix, g->n[0]);
        for (int iy = -1; iy < 2; iy++) {
            p[1] = wrap1(i[1] + iy, g->n[1]);
            for (int iz = -1; iz < 2; iz++) {
                p[2] = wrap1(i[2] + iz, g->n[2]);
                const int64_t lp = lin(g, p);
                if (vol[lp] == -1) continue;
                if (vol[lp] != vol_num) *is_edge = 1;
                if (g->rho[lp] > max_val) *is_max = 0;
            }
        }
    }
}
static void box_to_near(const grid_t *g, int8_t *known, const int64_t i[3]) {
    /* refinement.py:385-404 / 484-503: every known>=0 voxel of the 27-box becomes -1 */
    int64_t p[3];
    for (int ix = -1; ix < 2; ix++) {
        p[0] = wrap1(i[0] + ix, g->n[0]);
        for (int iy = -1; iy < 2; iy++) {
            p[1] = wrap1(i[1] + iy, g->n[1]);
            for (int iz = -1; iz < 2; iz++) {
                p[2] = wrap1(i[2] + iz, g->n[2]);
                const int64_t lp = lin(g, p);
                if (known[lp] >= 0) known[lp] = -1;
            }
        }
    }
}

int64_t orc_edge_find(int8_t *known, const double *rho, const int64_t shape[3], const lab_t *vol) {
    grid_t g; grid_init(&g, rho, shape, NULL, NULL);
    int64_t edge_num = 0, i[3];
    for (i[0] = 0; i[0] < shape[0]; i[0]++) for (i[1] = 0; i[1] < shape[1]; i[1]++) for (i[2] = 0; i[2] < shape[2]; i[2]++) {
        const int64_t li = lin(&g, i);
        if (known[li] == 2) continue;
        if (vol[li] == -1) continue;
        int is_edge, is_max;
        classify(&g, vol, i, &is_edge, &is_max);
        if (!is_edge || is_max) { /* refinement.py:376-381 */
            if (known[li] >= 0) known[li] = 2;
        } else {                  /* refinement.py:382-404 */
            known[li] = -2;
            edge_num++;
            box_to_near(&g, known, i);
        }
    }
    return edge_num;
}

/* refinement.edge_check (refinement.py:409-508): sequentially greedy re-classification of the
 * 27-box of every still -2 ("changed") voxel; note pe is NOT tested for vacuum. */
void orc_edge_check(int8_t *known, const double *rho, const int64_t shape[3], const lab_t *vol,
                    int64_t *checked_out, int64_t *edges_out) {
    grid_t g; grid_init(&g, rho, shape, NULL, NULL);
    const int64_t N = shape[0] * shape[1] * shape[2];
    int64_t checked = 0, edge_num = 0, i[3];
    for (i[0] = 0; i[0] < shape[0]; i[0]++) for (i[1] = 0; i[1] < shape[1]; i[1]++) for (i[2] = 0; i[2] < shape[2]; i[2]++) {
        if (known[lin(&g, i)] != -2) continue;
        int64_t pe[3];
        for (int ex = -1; ex < 2; ex++) {
            pe[0] = wrap1(i[0] + ex, g.n[0]);
            for (int ey = -1; ey < 2; ey++) {
                pe[1] = wrap1(i[1] + ey, g.n[1]);
                for (int ez = -1; ez < 2; ez++) {
                    pe[2] = wrap1(i[2] + ez, g.n[2]);
                    const int64_t lpe = lin(&g, pe);
                    int is_edge, is_max;
                    classify(&g, vol, pe, &is_edge, &is_max);
                    if (!is_edge) {              /* refinement.py:477-479 */
                        known[lpe] = -1;
                        checked++;
                    } else if (!is_max) {        /* refinement.py:480-504 */
                        if (known[lpe] != -3) {
                            known[lpe] = -3;
                            edge_num++;
                            box_to_near(&g, known, pe);
                            checked++;
                        }
                    }
                }
            }
        }
    }
    for (int64_t k = 0; k < N; k++) if (known[k] == -3) known[k] += 1; /* refinement.py:505-507 */
    *checked_out = checked; *edges_out = edge_num;
}

/* ------------------------------------------------------------------------------------------
 * refinement.neargrid, single block (refinement.py:17-322).  `rknown` is the read-only snapshot.
 * vol is modified in place at the start voxels only.  Returns `changed`.
 * ---------------------------------------------------------------------------------------- */
int64_t orc_refine_neargrid(int8_t *known, const int8_t *rknown, const double *rho, const int64_t shape[3],
                            lab_t *vol, const double *dist, const double *T) {
    grid_t g; grid_init(&g, rho, shape, dist, T);
    ivec path = {0};
    int64_t changed = 0, i[3];
    for (i[0] = 0; i[0] < shape[0]; i[0]++) for (i[1] = 0; i[1] < shape[1]; i[1]++) for (i[2] = 0; i[2] < shape[2]; i[2]++) {
        const int64_t li = lin(&g, i);
        if (known[li] != -2) continue;
        int64_t p[3] = {i[0], i[1], i[2]}, pd[3];
        double dr[3] = {0., 0., 0.};
        const lab_t vol_num = vol[li];
        known[li] += 5;
        path.len = 0; ivec_push(&path, li);
        for (;;) {
            ng_step(&g, 1, p, dr, pd);
            int64_t lpk = lin(&g, pd);
            int done = 0;
            if (known[lpk] >= 3 && known[lpk] <= 5) { /* refinement.py:200-292 */
                dr[0] = dr[1] = dr[2] = 0.;
                og_step(&g, p, pd);
                lpk = lin(&g, pd);
                if (pd[0] == p[0] && pd[1] == p[1] && pd[2] == p[2]) done = 1; /* a maximum */
            }
            if (done || rknown[lpk] == 2) { /* refinement.py:283-303 */
                const lab_t nv = vol[lpk];
                if (nv != vol_num) { vol[li] += nv - vol_num; changed++; }
                else known[li] += 1;
                break;
            }
            p[0] = pd[0]; p[1] = pd[1]; p[2] = pd[2]; /* refinement.py:305-315 */
            ivec_push(&path, lpk);
            if (known[lpk] < 2) known[lpk] += 5;
        }
        for (int64_t j = 0; j < path.len; j++) /* refinement.py:317-321 */
            if (known[path.v[j]] > 2) known[path.v[j]] -= 5;
    }
    free(path.v);
    return changed;
}

/* The "own-trajectory" map F of SURVEY.md section 7.3: every non-vacuum voxel follows its own
 * dr=0 neargrid trajectory (refinement.py:17-322 stepping rules, no early stop: rknown == 0) to
 * the maximum it reaches.  out[v] = linear index of that maximum (or -1 for vacuum voxels). */
static void own_trajectory(const double *rho, const int64_t shape[3], const lab_t *vol, const double *dist,
                           const double *T, int64_t *out, int strict) {
    grid_t g; grid_init(&g, rho, shape, dist, T);
    const int64_t N = shape[0] * shape[1] * shape[2];
    int8_t *mark = (int8_t *)calloc((size_t)N, 1);
    ivec path = {0};
    int64_t i[3];
    for (i[0] = 0; i[0] < shape[0]; i[0]++) for (i[1] = 0; i[1] < shape[1]; i[1]++) for (i[2] = 0; i[2] < shape[2]; i[2]++) {
        const int64_t li = lin(&g, i);
        if (vol[li] == -1) { out[li] = -1; continue; }
        int64_t p[3] = {i[0], i[1], i[2]}, pd[3];
        double dr[3] = {0., 0., 0.};
        mark[li] = 1;
        path.len = 0; ivec_push(&path, li);
        for (;;) {
            ng_step(&g, strict, p, dr, pd);
            int64_t lpk = lin(&g, pd);
            if (mark[lpk]) {
                dr[0] = dr[1] = dr[2] = 0.;
                og_step(&g, p, pd);
                lpk = lin(&g, pd);
                if (pd[0] == p[0] && pd[1] == p[1] && pd[2] == p[2]) { out[li] = lpk; break; }
            }
            p[0] = pd[0]; p[1] = pd[1]; p[2] = pd[2];
            ivec_push(&path, lpk);
            mark[lpk] = 1;
        }
        for (int64_t j = 0; j < path.len; j++) mark[path.v[j]] = 0;
    }
    free(mark); free(path.v);
}
void orc_own_trajectory(const double *rho, const int64_t shape[3], const lab_t *vol, const double *dist,
                        const double *T, int64_t *out) {
    own_trajectory(rho, shape, vol, dist, T, out, 1);
}
/* the same with the main pass's tie test (methods.py:324, `<= >=`): the trajectories methods.neargrid follows
 * when no known == 2 voxel cuts them short */
void orc_own_trajectory_main(const double *rho, const int64_t shape[3], const lab_t *vol, const double *dist,
                             const double *T, int64_t *out) {
    own_trajectory(rho, shape, vol, dist, T, out, 0);
}

/* The voxels of one own (dr=0) trajectory, start voxel first, maximum last -- the same stepping as
 * orc_own_trajectory.  Returns the path length (or -1 when it does not fit `cap`).  Test infrastructure for
 * the slab scheduler's remote path queries (pybader_amd/slab.py:_resolve_escaped). */
int64_t orc_trajectory_path(const double *rho, const int64_t shape[3], const double *dist, const double *T,
                            int64_t start, int64_t *out, int64_t cap) {
    grid_t g; grid_init(&g, rho, shape, dist, T);
    int64_t p[3] = {start / (shape[1] * shape[2]), (start / shape[2]) % shape[1], start % shape[2]}, pd[3];
    double dr[3] = {0., 0., 0.};
    int64_t n = 0;
    if (cap < 1) return -1;
    out[n++] = start;
    for (;;) {
        ng_step(&g, 1, p, dr, pd);
        int64_t lpk = lin(&g, pd);
        int seen = 0;
        for (int64_t j = n - 1; j >= 0 && !seen; j--) seen = (out[j] == lpk);
        if (seen) {
            dr[0] = dr[1] = dr[2] = 0.;
            og_step(&g, p, pd);
            lpk = lin(&g, pd);
            if (pd[0] == p[0] && pd[1] == p[1] && pd[2] == p[2]) return n;
        }
        if (n >= cap) return -1;
        p[0] = pd[0]; p[1] = pd[1]; p[2] = pd[2];
        out[n++] = lpk;
    }
}

/* ------------------------------------------------------------------------------------------
 * utils.* helpers on the path
 * ---------------------------------------------------------------------------------------- */
/* utils.vacuum_assign (utils.py:382-401) */
void orc_vacuum_assign(const double *reference, lab_t *vol, double vac_tol, const double *density,
                       double voxel_volume, int64_t N, double *charge_out, double *volume_out) {
    double charge = 0, volume = 0;
    for (int64_t k = 0; k < N; k++)
        if (reference[k] <= vac_tol) { vol[k] = -1; charge += density[k]; volume += voxel_volume; }
    charge *= voxel_volume;
    *charge_out = charge; *volume_out = volume;
}

/* utils.volume_offset for one block merged first (utils.py:496-510 with bader=0, edge=0):
 * 1-based local labels -> 0-based global labels. */
void orc_volume_offset(lab_t *vol, int64_t N) {
    for (int64_t k = 0; k < N; k++) if (vol[k] > 0) vol[k] -= 1;
}

/* utils.charge_sum (utils.py:235-252) */
void orc_charge_sum(double *charge, double *volume, int64_t n_labels, double voxel_volume,
                    const double *density, const lab_t *vol, int64_t N) {
    for (int64_t k = 0; k < N; k++) {
        const lab_t a = vol[k];
        if (0 <= a) { charge[a] += density[k]; volume[a] += voxel_volume; }
    }
    for (int64_t j = 0; j < n_labels; j++) charge[j] *= voxel_volume;
}

/* utils.atom_assign (utils.py:185-232): nearest atom over 27 periodic images, strict '<' */
void orc_atom_assign(const double *bader_max, int64_t n_max, const double *atoms, int64_t n_atoms,
                     const double *lattice, int64_t *atom_out, double *dist_out) {
    double pbc[3] = {0., 0., 0.};
    for (int64_t i = 0; i < n_max; i++) {
        const double *b = bader_max + 3 * i;
        /* utils.py:206-208: pbc holds the last image of the previous maximum (zeros for i == 0) */
        double d0 = b[0] - (atoms[0] + pbc[0]), d1 = b[1] - (atoms[1] + pbc[1]), d2 = b[2] - (atoms[2] + pbc[2]);
        double min_distance = (d0 * d0 + d1 * d1) + d2 * d2;
        int64_t atom_num = 0;
        for (int64_t j = 0; j < n_atoms; j++) {
            const double *a = atoms + 3 * j;
            for (int x = -1; x < 2; x++) for (int y = -1; y < 2; y++) for (int z = -1; z < 2; z++) {
                for (int k = 0; k < 3; k++)
                    pbc[k] = (lattice[k] * x + lattice[3 + k] * y) + lattice[6 + k] * z;
                d0 = b[0] - (a[0] + pbc[0]); d1 = b[1] - (a[1] + pbc[1]); d2 = b[2] - (a[2] + pbc[2]);
                const double dist = (d0 * d0 + d1 * d1) + d2 * d2;
                if (dist < min_distance) { min_distance = dist; atom_num = j; }
            }
        }
        atom_out[i] = atom_num;
        dist_out[i] = sqrt(min_distance); /* min_distance**.5 */
    }
}

/* utils.volume_assign (utils.py:404-421) */
void orc_volume_assign(lab_t *vol, const int64_t *swap, int64_t N) {
    for (int64_t k = 0; k < N; k++) if (vol[k] >= 0) vol[k] = (lab_t)swap[vol[k]];
}

/* Same density as pybader_amd/synth.py:synth_density (workload generator; see its docstring). */
void orc_synth_density(const int64_t shape[3], const double *lattice, const double *atoms, int64_t n_atoms,
                       double background, double *rho) {
    const int64_t nx = shape[0], ny = shape[1], nz = shape[2];
    for (int64_t i = 0; i < nx; i++) for (int64_t j = 0; j < ny; j++) for (int64_t k = 0; k < nz; k++) {
        const double f[3] = {(double)i / (double)nx, (double)j / (double)ny, (double)k / (double)nz};
        double r = background;
        for (int64_t a = 0; a < n_atoms; a++) {
            const double *A = atoms + 5 * a;
            double d[3];
            for (int m = 0; m < 3; m++) { d[m] = f[m] - A[m]; d[m] = d[m] - nearbyint(d[m]); }
            double r2 = 0.;
            for (int m = 0; m < 3; m++) {
                const double xm = (d[0] * lattice[m] + d[1] * lattice[3 + m]) + d[2] * lattice[6 + m];
                const double sq = xm * xm;
                r2 = (m == 0) ? sq : (r2 + sq);
            }
            double t = 1.0 - r2 / ((2048.0 * A[3]) * A[3]);
            if (!(t > 0.0)) t = 0.0;
            for (int s = 0; s < 10; s++) t = t * t;
            r = r + A[4] * t;
        }
        rho[(i * ny + j) * nz + k] = r;
    }
}

/* The density block of a CHGCAR (io/vasp.py:90-104, 147-149): whitespace separated decimal numbers in
 * Fortran order (x fastest), each converted by strtod (correctly rounded, as numpy's string -> float64),
 * stored as out[x][y][z] = value / divisor.  Returns the number of tokens converted (stops after nx*ny*nz),
 * or -1 - k when token k is malformed. */
int64_t orc_parse_density_text(const char *text, int64_t nbytes, const int64_t shape[3], double divisor, double *out) {
    const int64_t nx = shape[0], ny = shape[1], nz = shape[2], N = nx * ny * nz;
    int64_t pos = 0, k = 0;
    char buf[128];
    while (k < N) {
        while (pos < nbytes && (text[pos] == ' ' || (text[pos] >= 9 && text[pos] <= 13))) pos++;
        if (pos >= nbytes) break;
        int64_t end = pos;
        while (end < nbytes && !(text[end] == ' ' || (text[end] >= 9 && text[end] <= 13))) end++;
        if (end - pos >= (int64_t)sizeof buf) return -1 - k;
        memcpy(buf, text + pos, (size_t)(end - pos));
        buf[end - pos] = 0;
        char *stop;
        const double v = strtod(buf, &stop);
        if (stop == buf || *stop) return -1 - k;
        const int64_t x = k % nx, r = k / nx;
        out[(x * ny + r % ny) * nz + r / ny] = v / divisor;
        k++;
        pos = end;
    }
    return k;
}

void orc_free(void *p) { free(p); }
