/*
 * bader_oracle_blocks.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see bader_oracle.c, which this file includes).
 *
 * The reference's `threads > 1` path restated in C: thread_handlers.bader_calc / refine split the grid into
 * factor_3d(threads) blocks, run methods.neargrid on each block in a thread pool and merge the blocks
 * (thread_handlers.py:15-75, 128-236; utils.py:145-182, 262-317, 424-458, 479-510).  OpenMP runs the blocks in
 * parallel; they are merged in block order, i.e. the result is the reference's when its futures complete in
 * submission order (the reference merges in completion order, which makes its own numbering nondeterministic --
 * the golden vectors were captured with `as_completed` replaced by submission order, tests/golden/make_golden.py).
 * This is the all-cores leg of bench.py's cpu_baseline and the oracle of SURVEY.md section 8 rows a8 / a9.
 *
 * Block-local arrays: the reference gives every block a `volumes` / `known` array of the block's shape and GROWS
 * it when a path leaves the block (utils.volume_extend): indices >= 0 address the front of the array, negative
 * indices its tail, the middle is fresh zeros.  Here a block owns arrays of the full grid shape addressed modulo the
 * grid (index p >= 0 -> cell p, index q < 0 -> cell n + q): the same cells stay distinct as long as the reference's
 * array is not longer than the grid, which it caps (methods.py:397-399), and the bounds [negative_len, positive_len)
 * -- which decide what is "inside" -- evolve exactly as in the reference.
 */
#include "bader_oracle.c"

#ifdef _OPENMP
#include <omp.h>
#endif

/* utils.factor_3d (utils.py:283-317) */
void orc_factor_3d(int64_t x, int64_t out[3]) {
    int64_t fa[64], fb[64];
    int nf = 0;
    fa[nf] = 1; fb[nf] = x; nf++;
    for (int64_t i = 2; i < x; i++) {
        if ((double)i >= (double)x / (double)fa[nf - 1]) break;
        else if (x % i == 0) { fa[nf] = i; fb[nf] = x / i; nf++; }
    }
    const int64_t s0 = fa[nf - 1], s1 = fb[nf - 1];   /* split = fac[-1] */
    nf = 0;
    fa[nf] = 1; fb[nf] = s1; nf++;
    for (int64_t i = 2; i < s1; i++) {
        if ((double)i >= (double)s1 / (double)fa[nf - 1]) break;
        else if (s1 % i == 0) { fa[nf] = i; fb[nf] = s1 / i; nf++; }
    }
    out[0] = fb[nf - 1]; out[1] = fa[nf - 1]; out[2] = s0;
    if (fa[nf - 1] == 1) {
        /* fac.pop(0); fac.append((1, split[0])): the list is [(1, split[1])] here (a first factor of 1 means no
         * other pair was found), so it becomes [(1, split[0])] */
        nf = 0;
        fa[nf] = 1; fb[nf] = s0; nf++;
        for (int64_t i = 2; i < s0; i++) {
            if ((double)i >= (double)s0 / (double)fa[nf - 1]) break;
            else if (s0 % i == 0) { fa[nf] = i; fb[nf] = s0 / i; nf++; }
        }
        out[0] = fb[nf - 1]; out[1] = fa[nf - 1]; out[2] = s1;
    }
}

/* split[i] of thread_handlers.py:28-29: zip(density.shape, factor_3d(thread)) sorted by the shape entry
 * (stable, like Python's sorted), then the factors in that order */
static void block_split(const int64_t shape[3], int64_t threads, int64_t split[3]) {
    int64_t f[3];
    orc_factor_3d(threads, f);
    int order[3] = {0, 1, 2};
    for (int a = 1; a < 3; a++)      /* insertion sort: stable */
        for (int b = a; b > 0 && shape[order[b - 1]] > shape[order[b]]; b--) { int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t; }
    for (int k = 0; k < 3; k++) split[k] = f[order[k]];
}
/* np.array_split along one axis: the first n % parts pieces get one more */
static void split_axis(int64_t n, int64_t parts, int64_t k, int64_t *start, int64_t *len) {
    const int64_t base = n / parts, extra = n % parts;
    *len = base + (k < extra ? 1 : 0);
    *start = k * base + (k < extra ? k : extra);
}
/* Blocks in the order thread_handlers.py:30-47 produces them (np.ndindex(*split), axis 2 fastest) with their
 * origins.  The reference derives the origin from the previous block's by a wrap-around sum (lines 41-47); for
 * array_split pieces that equals the prefix sums computed here. */
typedef struct { int64_t idx[3], len[3]; } block_t;
static int64_t make_blocks(const int64_t shape[3], int64_t threads, block_t **out) {
    int64_t split[3];
    block_split(shape, threads < 1 ? 1 : threads, split);
    const int64_t nb = split[0] * split[1] * split[2];
    block_t *b = (block_t *)malloc((size_t)nb * sizeof(block_t));
    int64_t c = 0;
    for (int64_t i0 = 0; i0 < split[0]; i0++) for (int64_t i1 = 0; i1 < split[1]; i1++) for (int64_t i2 = 0; i2 < split[2]; i2++) {
        const int64_t ii[3] = {i0, i1, i2};
        for (int j = 0; j < 3; j++) split_axis(shape[j], split[j], ii[j], &b[c].idx[j], &b[c].len[j]);
        c++;
    }
    *out = b;
    return nb;
}
int64_t orc_block_table(const int64_t shape[3], int64_t threads, int64_t *idx_out, int64_t *len_out, int64_t cap) {
    block_t *b;
    const int64_t nb = make_blocks(shape, threads, &b);
    for (int64_t c = 0; c < nb && c < cap; c++)
        for (int j = 0; j < 3; j++) { idx_out[3 * c + j] = b[c].idx[j]; len_out[3 * c + j] = b[c].len[j]; }
    free(b);
    return nb;
}

/* ------------------------------------------------------------------------------------------
 * methods.neargrid on one block (methods.py:222-611) with the block-extension branches.
 * loc / known: arrays of the FULL grid shape owned by this block (see the header comment), zero outside the
 * block on entry; loc holds the block's part of `volumes` (0 / -1) in cells [0, len).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const grid_t *g;
    lab_t *vol;
    int8_t *known;
    int64_t neg[3], pos[3];          /* negative_len, positive_len */
    int64_t nneg[3], npos[3];        /* new_negative_len, new_positive_len */
    int64_t extend[3], vshape[3], idx[3];
} blk_t;
static inline int64_t cell(const blk_t *b, const int64_t pv[3]) {
    const grid_t *g = b->g;
    const int64_t c0 = pv[0] < 0 ? pv[0] + g->n[0] : pv[0], c1 = pv[1] < 0 ? pv[1] + g->n[1] : pv[1],
                  c2 = pv[2] < 0 ? pv[2] + g->n[2] : pv[2];
    return (c0 * g->n[1] + c1) * g->n[2] + c2;
}
/* methods.py:364-409 (and its copy 448-507): bring pv into the volume space or extend the space */
static void fit_or_extend(blk_t *b, int64_t pv[3]) {
    const grid_t *g = b->g;
    int extend_flag = 0;
    for (int j = 0; j < 3; j++) {
        if (pv[j] < b->neg[j]) {
            const int64_t upper = pv[j] + g->n[j] - b->pos[j] + 1, lower = (pv[j] - b->neg[j]) * -1;
            if (upper <= 0) pv[j] += g->n[j];
            else if (upper > lower) { b->nneg[j] -= b->vshape[j] / 2; b->extend[j] += b->vshape[j] / 2; extend_flag = 1; }
            else { b->npos[j] += b->vshape[j] / 2; b->extend[j] += b->vshape[j] / 2; pv[j] += g->n[j]; extend_flag = 1; }
        } else if (pv[j] >= b->pos[j]) {
            const int64_t upper = pv[j] - b->pos[j] + 1, lower = (pv[j] - g->n[j] - b->neg[j]) * -1;
            if (lower <= 0) pv[j] -= g->n[j];
            else if (upper > lower) { b->nneg[j] -= b->vshape[j] / 2; b->extend[j] += b->vshape[j] / 2; pv[j] -= g->n[j]; extend_flag = 1; }
            else { b->npos[j] += b->vshape[j] / 2; b->extend[j] += b->vshape[j] / 2; extend_flag = 1; }
        }
    }
    if (extend_flag) {   /* methods.py:396-409: volume_extend keeps every cell, only the bounds move */
        for (int j = 0; j < 3; j++) {
            if (b->extend[j] > g->n[j]) b->extend[j] = g->n[j];
            if (b->extend[j] == g->n[j]) { b->pos[j] = g->n[j]; b->neg[j] = 0; }
            else { b->pos[j] = b->npos[j]; b->neg[j] = b->nneg[j]; }
        }
    }
}
static inline int in_space(const blk_t *b, int h, int64_t v) { return b->neg[h] <= v && v < b->pos[h]; }
/* methods.py:556-577 / 581-603 for one neighbour q of a path voxel (q[k] already checked against the bounds) */
static void known_check_blk(blk_t *b, const int64_t q[3]) {
    const lab_t t = b->vol[cell(b, q)];
    if (t > -2 && t < 1) return;
    for (int h = 0; h < 3; h++) {
        int64_t r[3] = {q[0], q[1], q[2]};
        r[h] = q[h] + 1;
        if (!in_space(b, h, r[h]) || b->vol[cell(b, r)] != t) return;
        r[h] = q[h] - 1;
        if (!in_space(b, h, r[h]) || b->vol[cell(b, r)] != t) return;
    }
    b->known[cell(b, q)] = 2;
}
static void neargrid_block(const grid_t *g, const block_t *blk, lab_t *loc, int8_t *known, ivec *maxima, ivec *edges,
                           int64_t *n_max, int64_t *n_edge) {
    blk_t B;
    B.g = g; B.vol = loc; B.known = known;
    for (int j = 0; j < 3; j++) {
        B.vshape[j] = blk->len[j]; B.extend[j] = blk->len[j]; B.pos[j] = blk->len[j]; B.npos[j] = blk->len[j];
        B.neg[j] = 0; B.nneg[j] = 0; B.idx[j] = blk->idx[j];
    }
    ivec path = {0};   /* local pv triples */
    int64_t bader_num = 0, edge_num = 0;
    int64_t i[3];
    for (i[0] = 0; i[0] < blk->len[0]; i[0]++) for (i[1] = 0; i[1] < blk->len[1]; i[1]++) for (i[2] = 0; i[2] < blk->len[2]; i[2]++) {
        const int64_t ci = cell(&B, i);
        if (loc[ci] == -1) continue;          /* methods.py:286-289 */
        if (known[ci] == 2) continue;
        known[ci] = 1;
        int64_t p[3], pd[3], pv[3];
        double dr[3] = {0., 0., 0.};
        for (int j = 0; j < 3; j++) { p[j] = i[j] + B.idx[j]; pd[j] = p[j]; }
        path.len = 0;
        ivec_push(&path, i[0]); ivec_push(&path, i[1]); ivec_push(&path, i[2]);
        lab_t vol_num = 0;
        for (;;) {
            /* methods.py:302-363: a stationary gradient leaves pd == p (line 342-344: pv = pd - idx) */
            if (ng_step(g, 0, p, dr, pd)) { pd[0] = p[0]; pd[1] = p[1]; pd[2] = p[2]; }
            for (int j = 0; j < 3; j++) pv[j] = pd[j] - B.idx[j];
            fit_or_extend(&B, pv);
            if (known[cell(&B, pv)] == 1) {   /* methods.py:411-507 */
                dr[0] = dr[1] = dr[2] = 0.;
                og_step(g, p, pd);
                for (int j = 0; j < 3; j++) pv[j] = pd[j] - B.idx[j];
                const int break_flag = pd[0] == p[0] && pd[1] == p[1] && pd[2] == p[2];
                fit_or_extend(&B, pv);
                if (break_flag) {             /* methods.py:497-507 */
                    vol_num = 0;
                    if (loc[cell(&B, pv)] != 0) vol_num = loc[cell(&B, pv)];
                    else
                        for (int k = 0; k < 3; k++) {
                            if (pv[k] >= B.vshape[k]) vol_num = -2;
                            else if (pv[k] < 0) vol_num = -2;
                        }
                    break;
                }
            }
            if (known[cell(&B, pv)] == 2) { vol_num = loc[cell(&B, pv)]; break; }   /* methods.py:509-511 */
            for (int j = 0; j < 3; j++) p[j] = pd[j];                                /* methods.py:513-521 */
            ivec_push(&path, pv[0]); ivec_push(&path, pv[1]); ivec_push(&path, pv[2]);
            known[cell(&B, pv)] = 1;
        }
        if (vol_num == -2) {                   /* methods.py:523-531: a maximum outside the block */
            ivec_push(edges, pd[0]); ivec_push(edges, pd[1]); ivec_push(edges, pd[2]);
            edge_num++;
            vol_num = (lab_t)(-2 - edge_num);
        } else if (vol_num == 0) {             /* methods.py:533-541 */
            ivec_push(maxima, pd[0]); ivec_push(maxima, pd[1]); ivec_push(maxima, pd[2]);
            bader_num++;
            vol_num = (lab_t)bader_num;
        }
        for (int64_t j = 0; j < path.len / 3; j++) {   /* methods.py:543-606 */
            int64_t q[3] = {path.v[3 * j], path.v[3 * j + 1], path.v[3 * j + 2]};
            const int64_t cq = cell(&B, q);
            loc[cq] = vol_num;
            if (known[cq] != 2) known[cq] = 0;
            for (int k = 0; k < 3; k++) {
                q[k] += 1;
                if (in_space(&B, k, q[k])) known_check_blk(&B, q);
                q[k] -= 2;
                if (in_space(&B, k, q[k])) known_check_blk(&B, q);
                q[k] += 1;
            }
        }
    }
    free(path.v);
    *n_max = bader_num;
    *n_edge = edge_num;
}

/* thread_handlers.bader_calc for method 'neargrid' and threads > 1 (thread_handlers.py:15-75).
 * vol: the volumes_init map (0 / -1) on entry, the merged 0-based labels on exit.  Returns the number of maxima;
 * *maxima_out is malloc'ed [n*3].  `workers`: OpenMP threads (0: as many as blocks). */
int64_t orc_bader_calc_blocks(const double *rho, const int64_t shape[3], lab_t *vol, const double *dist, const double *T,
                              int64_t threads, int64_t workers, int64_t **maxima_out) {
    grid_t g; grid_init(&g, rho, shape, dist, T);
    const int64_t N = shape[0] * shape[1] * shape[2];
    block_t *blocks;
    const int64_t nb = make_blocks(shape, threads, &blocks);
    lab_t **locs = (lab_t **)calloc((size_t)nb, sizeof(lab_t *));
    ivec *bmax = (ivec *)calloc((size_t)nb, sizeof(ivec)), *bedge = (ivec *)calloc((size_t)nb, sizeof(ivec));
    int64_t *nmax = (int64_t *)calloc((size_t)nb, sizeof(int64_t)), *nedge = (int64_t *)calloc((size_t)nb, sizeof(int64_t));
    const int nw = (int)(workers > 0 ? workers : nb);
    (void)nw;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nw)
    for (int64_t c = 0; c < nb; c++) {
        const block_t *bk = &blocks[c];
        lab_t *loc = (lab_t *)calloc((size_t)N, sizeof(lab_t));
        int8_t *known = (int8_t *)calloc((size_t)N, 1);
        for (int64_t x = 0; x < bk->len[0]; x++) for (int64_t y = 0; y < bk->len[1]; y++)   /* vols[i]: the block's part */
            memcpy(loc + (x * shape[1] + y) * shape[2],
                   vol + ((x + bk->idx[0]) * shape[1] + y + bk->idx[1]) * shape[2] + bk->idx[2], (size_t)bk->len[2] * sizeof(lab_t));
        neargrid_block(&g, bk, loc, known, &bmax[c], &bedge[c], &nmax[c], &nedge[c]);
        free(known);
        locs[c] = loc;
    }
    /* merge in block order (thread_handlers.py:59-65): volume_offset, volume_merge, array_merge */
    ivec maxima = {0}, edges = {0};
    int64_t n_bader = 0, n_edge = 0;
    for (int64_t c = 0; c < nb; c++) {
        const block_t *bk = &blocks[c];
        for (int64_t x = 0; x < bk->len[0]; x++) for (int64_t y = 0; y < bk->len[1]; y++) for (int64_t z = 0; z < bk->len[2]; z++) {
            lab_t v = locs[c][(x * shape[1] + y) * shape[2] + z];
            if (v > 0) v += (lab_t)(n_bader - 1);          /* utils.volume_offset (utils.py:496-510) */
            else if (v < -2) v -= (lab_t)n_edge;
            vol[((x + bk->idx[0]) * shape[1] + y + bk->idx[1]) * shape[2] + z + bk->idx[2]] = v;   /* utils.volume_merge */
        }
        for (int64_t k = 0; k < 3 * nmax[c]; k++) ivec_push(&maxima, bmax[c].v[k]);               /* utils.array_merge */
        for (int64_t k = 0; k < 3 * nedge[c]; k++) ivec_push(&edges, bedge[c].v[k]);
        n_bader += nmax[c];
        n_edge += nedge[c];
        free(locs[c]); free(bmax[c].v); free(bedge[c].v);
    }
    if (n_edge > 0) {   /* utils.edge_assign (utils.py:262-280): one lookup per edge code, at the maximum's voxel */
        lab_t *swap = (lab_t *)malloc((size_t)n_edge * sizeof(lab_t));
        char *have = (char *)calloc((size_t)n_edge, 1);
        for (int64_t l = 0; l < N; l++) {
            const lab_t v = vol[l];
            if (v < -2) {
                const int64_t e = -1 * ((int64_t)v + 3);
                if (!have[e]) {
                    const int64_t *p = &edges.v[3 * e];
                    swap[e] = vol[(p[0] * shape[1] + p[1]) * shape[2] + p[2]];
                    have[e] = 1;
                }
                vol[l] = swap[e];
            }
        }
        free(swap); free(have);
    }
    free(locs); free(bmax); free(bedge); free(nmax); free(nedge); free(blocks); free(edges.v);
    *maxima_out = maxima.v;
    return n_bader;
}

/* refinement.neargrid over the blocks of thread_handlers.refine (thread_handlers.py:154-205): every block retraces
 * its own known == -2 voxels against the shared read-only snapshot `rknown` and the shared `vol` (written at the
 * start voxels only).  The reference marks a trace's path in the block's private copy of `known` (+5, removed
 * afterwards, refinement.py:155-198 grow that copy); here the path is a private list, which is the same predicate.
 * The merged `known` and the label map equal the single-block result. */
int64_t orc_refine_neargrid_blocks(int8_t *known, const int8_t *rknown, const double *rho, const int64_t shape[3], lab_t *vol,
                                   const double *dist, const double *T, int64_t threads, int64_t workers) {
    grid_t g; grid_init(&g, rho, shape, dist, T);
    block_t *blocks;
    const int64_t nb = make_blocks(shape, threads, &blocks);
    int64_t changed = 0;
    const int nw = (int)(workers > 0 ? workers : nb);
    (void)nw;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : changed) num_threads(nw)
    for (int64_t c = 0; c < nb; c++) {
        const block_t *bk = &blocks[c];
        ivec path = {0};
        int64_t i[3];
        for (i[0] = bk->idx[0]; i[0] < bk->idx[0] + bk->len[0]; i[0]++) for (i[1] = bk->idx[1]; i[1] < bk->idx[1] + bk->len[1]; i[1]++)
        for (i[2] = bk->idx[2]; i[2] < bk->idx[2] + bk->len[2]; i[2]++) {
            const int64_t li = lin(&g, i);
            if (known[li] != -2) continue;
            int64_t p[3] = {i[0], i[1], i[2]}, pd[3];
            double dr[3] = {0., 0., 0.};
            const lab_t vol_num = vol[li];
            path.len = 0; ivec_push(&path, li);
            for (;;) {
                ng_step(&g, 1, p, dr, pd);
                int64_t lpk = lin(&g, pd);
                int on_path = 0, done = 0;
                for (int64_t k = path.len - 1; k >= 0 && !on_path; k--) on_path = path.v[k] == lpk;
                if (on_path) {   /* refinement.py:200-292 */
                    dr[0] = dr[1] = dr[2] = 0.;
                    og_step(&g, p, pd);
                    lpk = lin(&g, pd);
                    if (pd[0] == p[0] && pd[1] == p[1] && pd[2] == p[2]) done = 1;
                }
                if (done || rknown[lpk] == 2) {   /* refinement.py:283-303 */
                    const lab_t nv = vol[lpk];
                    if (nv != vol_num) { vol[li] += nv - vol_num; changed++; }   /* known[li] stays -2 */
                    else known[li] = -1;                                         /* -2 + 5 + 1 - 5 */
                    break;
                }
                p[0] = pd[0]; p[1] = pd[1]; p[2] = pd[2];
                ivec_push(&path, lpk);
            }
        }
        free(path.v);
    }
    free(blocks);
    return changed;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
