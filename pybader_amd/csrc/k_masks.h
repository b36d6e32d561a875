// k_masks.h -- device kernels of libbader_hip.so: the two passes that replace the full gradient-field table on the
// single-GPU path.  Included by bader_hip.hip (one translation unit).
//
//   pass A  k_brick_masks    every voxel: which neighbour bricks can ANY possible move of the voxels of a brick reach
//                            (the input of the brick fixpoint, k_brick_grow) + the 26-neighbour maxima.  No division,
//                            no 32-byte record: the move INTERVALS of a voxel depend on the signs of its gradient
//                            components and on two threshold tests only, and the ongrid successor only matters for
//                            voxels on a brick face whose gradient interval does not already cross that face.
//   pass B  k_brick_records  the bricks OUTSIDE the trapping regions (the walk list, ~17 % of the voxels at 512^3):
//                            the full 32-byte record of every voxel (the arithmetic of k_grad_field, operation for
//                            operation).  Walkers stop as soon as they arrive in a trapping region, so no record of a
//                            certain brick is ever used; `brick_rec[b]` says which bricks hold records, and the
//                            refinement kernels derive the few records they need elsewhere from rho (make_rec_rho).
//
// Round 1 wrote a record for every voxel: 4.2 GB of stores and 354 f64 VALU instructions per voxel (1.7 ms at 512^3),
// 83 % of them for voxels inside trapping regions whose records nobody reads.
#pragma once

// ---- tile / brick geometry shared by the kernels of the brick pipeline ----
// the density tile of passes A (k_brick_masks) and of the ongrid pass (k_og_masks): GT_Z / 8 whole 8^3 bricks in a row along z
#define GT_X 8
#define GT_Y 8
#define GT_Z 32

#define BRK 8   // edge of a brick (voxels)
#define BG 8    // edge of the chunk of bricks a workgroup of the region growth iterates in LDS (k_brick_grow_dev)

__device__ __forceinline__ int wrap_any(int v, int n) { v %= n; return v < 0 ? v + n : v; }

// brick_rec[b] := flag for the bricks whose x-brick index lies in [b0, b0 + nb) modulo nb0 (the table window of a slab; the
// whole lattice on one GPU), 0 elsewhere: k_brick_records(walk == nullptr) then writes the records of exactly those bricks
__global__ void k_flag_window_bricks(int nb0, int per_plane, int b0, int nb, unsigned char flag, unsigned char *brick_rec) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb0 * per_plane) return;
    int d = b / per_plane - b0;
    if (d < 0) d += nb0;
    brick_rec[b] = d < nb ? flag : (unsigned char)0;
}


// Conservative move interval of one axis without the division (see move_ranges_raw for the exact form):
// with d = grad_dir component, m = max |component| (>= 1e-14), the reference steps int_grad + rha(dr + r),
// int_grad = rha(d/m), r = d/m - int_grad, |dr| <= 1/2:
//   d/m in [1e-12, 1-1e-12]   -> offsets {0, +1}            (int_grad 0 or 1, the correction can only go the other way)
//   d/m > 1 - 1e-12 (max axis) -> {0, +1, +2}
//   |d/m| < 1e-12              -> {-1, 0, +1}
// and mirrored for negative d.  The thresholds below are 2e-12: a superset of the exact intervals in the two slivers
// of width 1e-12, equal everywhere else (soundness only needs a superset).
#define BM_EPS 2e-12
#ifdef XB_DEBUG_COUNT
__device__ unsigned long long xb_dbg[65536];   // [0..15] pass A's counters; [16..] the trace's time probes (k_ng_trace_g)
__device__ signed char *xb_dbg_steps;          // per start voxel: the steps its walker took (tools/walk_lengths.py), else nullptr
#endif

// thread -> (y, z) column of the 8 x 32 tile face, sorted by what a column can contribute to its brick's move mask:
//   wave 0  the low-side border columns (y == 0 or z == 0) and the four mixed corners per tile row (+ 4 second-layer ones),
//   wave 1  the purely high-side border columns (y == 7 or z == 7) (+ 12 second-layer ones),
//   wave 2  second-layer columns (y or z in {1, 6}, none on a face: a move of two voxels still leaves the brick from there),
//   wave 3  the 64 DEEP columns (y and z in 2..5): no y- or z-face within reach of any move, and at x-positions 2..5 no
//           x-face either -- there wave 3 skips the gradient altogether (only the maximum test is left).
// A border wave needs the ongrid face test on its own faces only, and a field that points the same way across the tile
// leaves one of the two border waves without any face to test.
// Round 4: the 9 ds_read_b64 of a voxel are served per 32-lane half, bank pair = (address / 8) mod 32 = (35 ty + tz) mod 32
// at a row stride of 35 doubles.  The order below keeps the four wave classes and arranges every half so that its 32 columns
// fall on distinct bank pairs (waves 0 and 1: both halves; waves 2 and 3: one half, two columns per pair at most in the
// other) -- 10 LDS cycles per read of the workgroup where 8 is the floor and round 3's order took 20 (its conflict cycles
// were three times the cycles the reads themselves needed: profiles/r3_final_pmc_sq_512_neargrid.txt).  A linear layout cannot
// do better with these classes: the low-side columns want an odd stride, the deep ones a stride of 4 mod 8.
#define BM_ROW 35   // row length of the LDS tile in doubles (34 are used)
// thread -> column, (ty << 5) | tz.  Waves as above: 0 the low-side border columns (+ the mixed corners, + (1,5) of each brick),
// 1 the high-side ones (+ (1,1), (1,2), (1,4)), 2 the other second-layer columns, 3 the deep ones.
__device__ const unsigned char bm_column_tab[TPB] = {
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31,
    37, 32, 64, 96, 128, 160, 192, 224, 45, 40, 72, 104, 136, 168, 200, 232, 53, 48, 80, 112, 144, 176, 208, 240, 61, 56, 88, 120, 152, 184, 216, 248,
    225, 226, 227, 228, 229, 230, 231, 71, 233, 234, 235, 236, 237, 238, 239, 79, 241, 242, 243, 244, 245, 246, 247, 87, 249, 250, 251, 252, 253, 254, 255, 95,
    39, 103, 135, 167, 199, 33, 34, 36, 47, 111, 143, 175, 207, 41, 42, 44, 55, 119, 151, 183, 215, 49, 50, 52, 63, 127, 159, 191, 223, 57, 58, 60,
    195, 129, 197, 65, 198, 97, 194, 35, 203, 137, 205, 73, 206, 105, 202, 43, 211, 145, 213, 81, 214, 113, 210, 51, 219, 153, 221, 89, 222, 121, 218, 59,
    166, 102, 161, 134, 70, 196, 38, 193, 174, 110, 169, 142, 78, 204, 46, 201, 182, 118, 177, 150, 86, 212, 54, 209, 190, 126, 185, 158, 94, 220, 62, 217,
    98, 130, 132, 133, 162, 163, 164, 165, 106, 138, 140, 141, 170, 171, 172, 173, 114, 146, 148, 149, 178, 179, 180, 181, 122, 154, 156, 157, 186, 187, 188, 189,
    66, 67, 68, 69, 99, 100, 101, 131, 74, 75, 76, 77, 107, 108, 109, 139, 82, 83, 84, 85, 115, 116, 117, 147, 90, 91, 92, 93, 123, 124, 125, 155,
};
__device__ __forceinline__ void bm_column(int t, int &ty, int &tz) {
    const int c = bm_column_tab[t];
    ty = c >> 5; tz = c & 31;
}

// v_max_f64 without the canonicalisation fmax() adds for operands that come straight from memory (no NaNs in a density)
__device__ __forceinline__ double max_raw(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double min_raw(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// Mirror prefilter of the ongrid face test.  The ongrid pick (methods.py:87-117) maximises w_n = fl(fl((rho_n - c) d_n) + c)
// over the 26 neighbours, strict '>'.  When the distance matrix is mirror symmetric along an axis (d(-1,j,k) == d(+1,j,k):
// that lattice vector is orthogonal to the other two) every neighbour n0 of the low plane has a mirror image n2 in the
// high plane with the same weight, and fl(.) is monotone: if rho(n2) - rho(n0) > mu for all nine pairs, with
//     mu = 2^-48 (1 + 1/d_min) max|rho| over the tile  >  2^-52 B (6 + 1/d)   (three roundings per w, |rho_n - c| <= 2B),
// then w(n2) > w(n0) STRICTLY for every pair, the largest w of the low plane is smaller than that of the high plane, and
// the pick cannot lie in the low plane: the ongrid step of this voxel does not cross the low face.  A smooth field that
// points away from a face satisfies this almost everywhere (rho(+1,j,k) - rho(-1,j,k) = 2 g_x to first order for every
// j,k), so the 115-instruction plane-maximum test below only runs for the waves that hold a voxel where it does not.
// D: +1 tests the low plane against the high one, -1 the high plane against the low one.
template <int AX>
__device__ __forceinline__ double bm_pair(const double (&a)[3][3][3], int u, int v) {
    return AX == 0 ? a[2][u][v] - a[0][u][v] : (AX == 1 ? a[u][2][v] - a[u][0][v] : a[u][v][2] - a[u][v][0]);
}
template <int AX>
__device__ __forceinline__ void bm_mirror(const double (&a)[3][3][3], bool want_lo, bool want_hi, double mu, bool &nlo, bool &nhi) {
    double e[9];
#pragma unroll
    for (int u = 0; u < 3; u++)
#pragma unroll
        for (int v = 0; v < 3; v++) e[u * 3 + v] = bm_pair<AX>(a, u, v);
    if (want_lo) {   // wave uniform
        double m = min_raw(e[0], e[1]);
#pragma unroll
        for (int k = 2; k < 9; k++) m = min_raw(m, e[k]);
        nlo = nlo && !(m > mu);
    }
    if (want_hi) {
        double m = max_raw(e[0], e[1]);
#pragma unroll
        for (int k = 2; k < 9; k++) m = max_raw(m, e[k]);
        nhi = nhi && !(m < -mu);
    }
}
// One tie rule's contribution to the face-crossing booleans of a voxel at x-position K of its column.  MT: methods.py:324
// (1) or refinement.py:111 (0), compile time.  Round-3 diet (the pass is VALU-issue bound, ~140 of its 162 instructions per
// voxel were here): the masks are internal -- any SUPERSET of the possible moves is sound, and the slivers of BM_EPS leave
// 1e-12 of slack -- so nothing below has to reproduce the reference's roundings:
//   * `m_j = max(h_j, l_j)` (shared with the maximum test) gives the axis tests of both rules: flat under methods.py:324
//     iff m_j <= c, under refinement.py:111 iff m_j < c, and a tie iff m_j == c;
//   * the halving of the central difference is dropped: d' = T (h - l) = 2 d exactly (scaling by two commutes with
//     every rounding), so d' is compared against thresholds of mg' = 2 mg and `max_grad < 1E-14` reads mg' < 2E-14;
//   * DIAG: T_grad is diagonal (an orthogonal lattice): d'_j = T_jj e_j, three products instead of nine and six sums
//     (the off-diagonal terms are exact zeros in the reference's sum);
//   * the two-threshold test per face is ONE product and ONE comparison against mg' * coef, coef a per-lane constant of
//     the column: +EPS on the face (offset -1 reachable when d < tiny), -(1 - EPS) one voxel in (offset -2 when d <
//     -nearm), -inf elsewhere (no move reaches the face); mirrored for the high face.
struct BmCoef { double yl, yh, zl, zh; };
// PART (round 4: grids that are not made of whole bricks): the high face of a brick lies where the grid ends, at position
// w - 1 of its w valid voxels -- the x thresholds come as a scalar (cxh) like those of y and z come per lane (co).
template <int MT, int K, bool DIAG, bool PART, typename GT>
__device__ __forceinline__ void bm_cross(const GT &g, double c, double ex, double ey, double ez, double mx, double my, double mz,
                                         const BmCoef &co, double cxh, bool &xl, bool &xh, bool &yl, bool &yh, bool &zl, bool &zh) {
    const bool f0 = MT ? (mx <= c) : (mx < c), f1 = MT ? (my <= c) : (my < c), f2 = MT ? (mz <= c) : (mz < c);
    const double g0 = f0 ? 0. : ex, g1 = f1 ? 0. : ey, g2 = f2 ? 0. : ez;
    double d0, d1, d2;
    if (DIAG) {
        d0 = g.T[0] * g0; d1 = g.T[4] * g1; d2 = g.T[8] * g2;
    } else {
        d0 = ((g.T[0] * g0) + (g.T[1] * g1)) + (g.T[2] * g2);
        d1 = ((g.T[3] * g0) + (g.T[4] * g1)) + (g.T[5] * g2);
        d2 = ((g.T[6] * g0) + (g.T[7] * g1)) + (g.T[8] * g2);
    }
    const double mg = fmax(fmax(fabs(d0), fabs(d1)), fabs(d2));
    const bool moves = !(mg < 2E-14);   // max_grad < 1E-14 (d is doubled here): the ongrid step only
    if (K == 0) xl |= moves && d0 < mg * BM_EPS;
    if (K == 1) xl |= moves && d0 < mg * -(1. - BM_EPS);
    if (PART) xh |= moves && d0 > mg * cxh;
    else {
        if (K == GT_X - 1) xh |= moves && d0 > mg * -BM_EPS;
        if (K == GT_X - 2) xh |= moves && d0 > mg * (1. - BM_EPS);
    }
    yl |= moves && d1 < mg * co.yl;
    yh |= moves && d1 > mg * co.yh;
    zl |= moves && d2 < mg * co.zl;
    zh |= moves && d2 > mg * co.zh;
}

// one x-position K of the column: everything below the window update
template <int MT, int K, bool DIAG, bool PART, typename GT>
__device__ __forceinline__ void bm_voxel(const GT &g, const double (&a)[3][3][3], double dface, bool in, int v, int ty, int zz,
                                         int &mine, bool &any_tie, int *s_cnt, int *s_mv, double mu, int mirror, const BmCoef &co,
                                         bool deep, int wx, bool at_yh, bool at_zh) {
    // this voxel on the high x / y / z face of its brick (PART: of what the grid leaves of the brick)
    const bool at_xh = PART ? (K == wx - 1) : (K == GT_X - 1);
    const double cxh = at_xh ? -BM_EPS : (K == wx - 2 ? (1. - BM_EPS) : __builtin_huge_val());
    const double c = a[1][1][1];
    const double hx = a[2][1][1], lx = a[0][1][1], hy = a[1][2][1], ly = a[1][0][1], hz = a[1][1][2], lz = a[1][1][0];
    const double mx = max_raw(hx, lx), my = max_raw(hy, ly), mz = max_raw(hz, lz);
    // an axis tie (the two rules disagree on this voxel: methods.py:324 zeroes the axis, refinement.py:111 does not)
    const bool tie = mx == c || my == c || mz == c;
    any_tie |= in && tie;
    // can a move of this voxel leave its brick through the low / high face of each axis?  A move reaches offset
    // -1 when d < tiny and -2 when d < -nearm (mirrored upwards), so at position p of 0..7 the low face is crossed
    // iff (p == 0 and d < tiny) or (p <= 1 and d < -nearm).  Booleans throughout: they stay lane masks.
    bool xl = false, xh = false, yl = false, yh = false, zl = false, zh = false;
    // (a deep column at x-positions 2..5 has no face within reach of any move: `deep` is wave uniform)
    if (!(deep && K >= 2 && K <= GT_X - 3)) {
        const double ex = hx - lx, ey = hy - ly, ez = hz - lz;
        bm_cross<MT, K, DIAG, PART>(g, c, ex, ey, ez, mx, my, mz, co, cxh, xl, xh, yl, yh, zl, zh);
        // a voxel with a tie axis gets the union of both rules, so that a trapping region is closed for the assignment's
        // walkers AND for the refinement's retraces (k_refine_trace stops a retrace that enters a region)
        if (__any(tie)) bm_cross<!MT, K, DIAG, PART>(g, c, ex, ey, ez, mx, my, mz, co, cxh, xl, xh, yl, yh, zl, zh);
    }
    // Does this voxel need its exact ongrid successor?  (a) it lies on a face of its brick that its gradient
    // interval does not cross already (an ongrid move is one voxel long: only face voxels can leave the brick
    // by it); (b) it may be a 26-neighbour maximum: not ruled out by a face neighbour whose weighted value
    // (bounded from below with the smallest face distance; fl(.) is monotone) exceeds c.
    const double r6 = max_raw(max_raw(mx, my), mz);
    double wq = (r6 - c) * dface;
    wq += c;
    const bool not_max = wq > c;
    // the faces whose crossing is still open, minus those the mirror prefilter closes (wave-uniform branches)
    bool nxl = in && K == 0 && !xl, nxh = in && at_xh && !xh;
    bool nyl = in && ty == 0 && !yl, nyh = in && at_yh && !yh, nzl = in && zz == 0 && !zl, nzh = in && at_zh && !zh;
    if ((PART || K == 0 || K == GT_X - 1) && (mirror & 1)) {
        const bool wl = K == 0 && __any(nxl), wh = at_xh && __any(nxh);
        if (wl || wh) bm_mirror<0>(a, wl, wh, mu, nxl, nxh);
    }
    if (mirror & 2) {
        const bool wl = __any(nyl), wh = __any(nyh);
        if (wl || wh) bm_mirror<1>(a, wl, wh, mu, nyl, nyh);
    }
    if (mirror & 4) {
        const bool wl = __any(nzl), wh = __any(nzh);
        if (wl || wh) bm_mirror<2>(a, wl, wh, mu, nzl, nzh);
    }
    const bool need = nxl || nxh || nyl || nyh || nzl || nzh || !not_max;
    bool is_max = false;
#ifdef XB_DEBUG_COUNT   // diagnostic build only (tools/debug_counts.py): why does a wave-iteration run the exact test?
    {
        const bool ax = __any(nxl || nxh), ay = __any(nyl || nyh), az = __any(nzl || nzh), am = __any(in && !not_max), an = __any(need && in);
        if (threadIdx.x % XB_WAVE == 0) {
            atomicAdd(&xb_dbg[0], 1ull);
            if (ax) atomicAdd(&xb_dbg[1], 1ull);
            if (ay) atomicAdd(&xb_dbg[2], 1ull);
            if (az) atomicAdd(&xb_dbg[3], 1ull);
            if (am) atomicAdd(&xb_dbg[4], 1ull);
            if (an) atomicAdd(&xb_dbg[5], 1ull);
        }
    }
#endif
    if (__any(need && in)) {
        // methods.py:87-117 picks the FIRST neighbour (ix,iy,iz ascending) whose weighted value w = fl(fl((rho_n - c) d_n) + c)
        // is the largest and > c.  Only two things are needed of it here: whether any w exceeds c (else the voxel is a
        // maximum), and whether the pick can lie in a face plane of the 3x3x3 box.  fl(. + c) is monotone, so the largest w
        // of a set of neighbours is fl(max p + c) with p = fl((rho_n - c) d_n): one subtraction and one multiplication
        // per neighbour, maxima per z-line, per plane and overall, and the final addition for the few maxima only.  A
        // plane "can hold the pick" iff its largest w is > c and equals the overall largest (on an exact tie with a
        // neighbour outside the plane the reference takes whichever comes first: counted as a crossing, a superset).
        double pz[3][3][3];
#pragma unroll
        for (int ix = 0; ix < 3; ix++)
#pragma unroll
            for (int iy = 0; iy < 3; iy++)
#pragma unroll
                for (int iz = 0; iz < 3; iz++)
                    if (!(ix == 1 && iy == 1 && iz == 1)) pz[ix][iy][iz] = (a[ix][iy][iz] - c) * dist_at(g, ix, iy, iz);
        double line[3][3];   // per (ix,iy): max over iz
#pragma unroll
        for (int ix = 0; ix < 3; ix++)
#pragma unroll
            for (int iy = 0; iy < 3; iy++)
                line[ix][iy] = (ix == 1 && iy == 1) ? fmax(pz[1][1][0], pz[1][1][2]) : fmax(fmax(pz[ix][iy][0], pz[ix][iy][1]), pz[ix][iy][2]);
        const double ax0 = fmax(fmax(line[0][0], line[0][1]), line[0][2]);
        const double ax1 = fmax(fmax(line[1][0], line[1][1]), line[1][2]);
        const double ax2 = fmax(fmax(line[2][0], line[2][1]), line[2][2]);
        const double pm = fmax(fmax(ax0, ax1), ax2);
        const double wm = pm + c;
        is_max = !(wm > c);
        const bool some = wm > c;
        if (K == 0) xl |= some && (ax0 + c) == wm;
        if (PART ? at_xh : K == GT_X - 1) xh |= some && (ax2 + c) == wm;
        {
            const double ay0 = fmax(fmax(line[0][0], line[1][0]), line[2][0]), ay2 = fmax(fmax(line[0][2], line[1][2]), line[2][2]);
            yl |= ty == 0 && some && (ay0 + c) == wm;
            yh |= at_yh && some && (ay2 + c) == wm;
            double az0 = pz[0][0][0], az2 = pz[0][0][2];
#pragma unroll
            for (int ix = 0; ix < 3; ix++)
#pragma unroll
                for (int iy = 0; iy < 3; iy++) {
                    az0 = fmax(az0, pz[ix][iy][0]);
                    az2 = fmax(az2, pz[ix][iy][2]);
                }
            zl |= zz == 0 && some && (az0 + c) == wm;
            zh |= at_zh && some && (az2 + c) == wm;
        }
    }
    if (in) {
        if (is_max) {   // how many 26-neighbour maxima the brick holds, and one of them
            atomicAdd(s_cnt, 1);
            *s_mv = v;
        }
        // per axis the set of brick offsets {-1,0,+1} a move can reach (0 always: bit 1), then the 27-bit outer
        // product z -> y -> x by shifts (bit = (dx+1)*9 + (dy+1)*3 + (dz+1))
        const int pc = 2 | (zl ? 1 : 0) | (zh ? 4 : 0);
        const int yz = (pc << 3) | (yl ? pc : 0) | (yh ? pc << 6 : 0);
        mine |= (yz << 9) | (xl ? yz : 0) | (xh ? yz << 18 : 0);
    }
}

// (SMALL: a grid so small that the tile's halo wraps more than once -- true modulo instead of the two unsigned minima)
template <bool SMALL, bool PART, typename GT>
__device__ __forceinline__ void bm_stage(const GT &g, const double *__restrict__ rho, double (&tile)[GT_X + 2][GT_Y + 2][BM_ROW], unsigned *s_bmax,
                                         int x0, int y0, int z0, int mirror) {
    {   // Staging of the haloed 10 x 10 x 34 tile, every load of a wave in flight before the first wait.
        // Round 4: the address of a row is ONE scalar addition.  (Round 3 derived every row's (x, y) -- a division, two wraps, two
        // multiplications -- on the scalar unit and predicated every load by `lane < 34` with an exec branch of its own: ~30 SALU
        // instructions in front of each of the 25 loads of a wave, 750 of the ~1000 SALU instructions a wave issued per tile,
        // on a scalar unit the boolean algebra of the x-walk needs as well.)  Wave w stages the x-planes w, w + 4 (and w + 8
        // for w < 2) of the tile, ten rows each: the plane's byte offset is one scalar (its periodic wrap included), the ten
        // row offsets of the y halo are computed once per wave, the z wrap once per lane; lanes beyond the 34 doubles of a row
        // repeat lane 33's address (no exec change).  32-bit byte offsets inside a plane: planes up to 2^29 voxels.
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / XB_WAVE), lane = threadIdx.x % XB_WAVE;
        const int lz = min(lane, GT_Z + 1);
        int Z = z0 + lz - 1;
        unsigned Yw[GT_Y + 2];   // byte offsets of the rows inside a plane (scalars)
        Z = SMALL ? ((Z % g.nz) + g.nz) % g.nz : wrap_u(Z, g.nz);
        const unsigned zoff = (unsigned)Z * 8u;   // the lane's byte offset inside a row: scalar row base + 32-bit lane offset
#pragma unroll
        for (int ey = 0; ey < GT_Y + 2; ey++) {
            int Y = y0 + ey - 1;
            if (SMALL) Y = ((Y % g.ny) + g.ny) % g.ny;
            else if (PART || ey == 0 || ey == GT_Y + 1) Y = wrap_u(Y, g.ny);   // (PART: a tile may reach beyond the grid, its rows wrap)
            Yw[ey] = (unsigned)(Y * g.nz) * 8u;   // (interior rows of the tile never wrap: whole-brick grids; a partial tile's rows beyond the grid are not read)
        }
        constexpr int PL = (GT_X + 2 + 3) / 4;   // x-planes per wave (the last one only for the waves that have it)
        double val[PL][GT_Y + 2];
#pragma unroll
        for (int j = 0; j < PL; j++) {
            const int ex = wv + 4 * j;
            if (ex < GT_X + 2) {   // wave uniform
                int X = x0 + ex - 1;
                X = SMALL ? ((X % g.nx) + g.nx) % g.nx : wrap_u(X, g.nx);
                const char *plane = reinterpret_cast<const char *>(rho + (size_t)X * g.nyz);   // scalar
#pragma unroll
                for (int ey = 0; ey < GT_Y + 2; ey++) val[j][ey] = *reinterpret_cast<const double *>(plane + Yw[ey] + zoff);
            } else {
#pragma unroll
                for (int ey = 0; ey < GT_Y + 2; ey++) val[j][ey] = 0.;
            }
        }
        if (lane < GT_Z + 2) {
#pragma unroll
            for (int j = 0; j < PL; j++) {
                const int ex = wv + 4 * j;
                if (ex < GT_X + 2) {
#pragma unroll
                    for (int ey = 0; ey < GT_Y + 2; ey++) tile[ex][ey][lane] = val[j][ey];
                }
            }
        }
        if (mirror) {
            unsigned hi = 0;
#pragma unroll
            for (int j = 0; j < PL; j++)
#pragma unroll
                for (int ey = 0; ey < GT_Y + 2; ey++) hi = max(hi, (unsigned)__double2hiint(val[j][ey]) & 0x7fffffffu);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) hi = max(hi, (unsigned)__shfl_xor((int)hi, o));
            if (lane == 0) atomicMax(s_bmax, hi);
        }
    }
}
// PART: the grid is not made of whole 8^3 bricks.  The brick lattice is ceil(n / 8) per axis; the last brick of an axis holds
// the w = n mod 8 voxels the grid leaves of it, its high face lies at position w - 1, moves from it wrap to brick 0 and moves
// from brick 0 downwards arrive in it -- the same 27-neighbour masks.  Only a brick of width ONE is special: a move of two
// voxels from brick 0 would jump over it, so it is marked as never certifiable (bits 27 + 28), and every brick with a move
// into it dies with it in the kill iteration: sound, and a width of one is rare.
template <typename GT, int MT, bool DIAG, bool PART = false>
__global__ __launch_bounds__(TPB) void k_brick_masks(GT g, const double *__restrict__ rho, int small, int *__restrict__ bmask,
                                                     int *__restrict__ bmaxv, int *tie_count, int xbase, double mu_scale, int mirror,
                                                     int *__restrict__ bpot) {
    __shared__ double tile[GT_X + 2][GT_Y + 2][BM_ROW];
    __shared__ int s_mask[GT_Z / 8], s_cnt[GT_Z / 8], s_mv[GT_Z / 8], s_pot[GT_Z / 8];
    __shared__ unsigned s_bmax;   // the largest |rho| of the haloed tile, as the high word of its double (mirror prefilter)
    // (xbase: the first plane; a slab runs the pass over its own planes only, brick aligned)
    const int x0 = xbase + blockIdx.z * GT_X, y0 = blockIdx.y * GT_Y, z0 = blockIdx.x * GT_Z;
    if (threadIdx.x < GT_Z / 8) { s_mask[threadIdx.x] = 0; s_cnt[threadIdx.x] = 0; s_mv[threadIdx.x] = -1; s_pot[threadIdx.x] = -2147483647 - 1; }
    if (threadIdx.x == 0) s_bmax = 0;
    __syncthreads();
    if (small & 1) bm_stage<true, PART>(g, rho, tile, &s_bmax, x0, y0, z0, mirror);
    else bm_stage<false, PART>(g, rho, tile, &s_bmax, x0, y0, z0, mirror);
    __syncthreads();
    // mu: see bm_mirror; the double whose high word is s_bmax + 1 bounds every |rho| of the tile from above
    const double mu = mirror ? __hiloint2double((int)min(s_bmax + 1u, 0x7ff00000u), 0) * mu_scale : 0.;
    int ty, tz;
    bm_column(threadIdx.x, ty, tz);
    const int zz = tz & 7;
    // the valid voxels of this lane's brick along each axis
    const int wx = PART ? min(GT_X, g.nx - x0) : GT_X, wy = PART ? min(8, g.ny - y0) : 8, wz = PART ? min(8, g.nz - (z0 + (tz & ~7))) : 8;
    const bool at_yh = ty == wy - 1, at_zh = zz == wz - 1;
    // wave 3 holds the deep columns (bm_column); in a tile the grid cuts they are not deep
    const bool deep = __builtin_amdgcn_readfirstlane(threadIdx.x / XB_WAVE) == 3 && (!PART || (wx == GT_X && wy == 8 && z0 + GT_Z <= g.nz));
    const double kInf = __builtin_huge_val();
    BmCoef co;   // per-lane threshold coefficients of the y / z faces (bm_cross)
    co.yl = ty == 0 ? BM_EPS : (ty == 1 ? -(1. - BM_EPS) : -kInf);
    co.yh = at_yh ? -BM_EPS : (ty == wy - 2 ? (1. - BM_EPS) : kInf);
    co.zl = zz == 0 ? BM_EPS : (zz == 1 ? -(1. - BM_EPS) : -kInf);
    co.zh = at_zh ? -BM_EPS : (zz == wz - 2 ? (1. - BM_EPS) : kInf);
    const int y = y0 + ty, z = z0 + tz;
    const bool col_in = y < g.ny && z < g.nz;
    // the smallest of the six face distances: a lower bound of any face neighbour's weighted value (maximum test)
    const double dface = fmin(fmin(fmin(dist_at(g, 0, 1, 1), dist_at(g, 2, 1, 1)), fmin(dist_at(g, 1, 0, 1), dist_at(g, 1, 2, 1))),
                              fmin(dist_at(g, 1, 1, 0), dist_at(g, 1, 1, 2)));
    // rolling 3x3x3 window along x: 9 LDS reads per voxel instead of 27
    double a[3][3][3];
#pragma unroll
    for (int iy = 0; iy < 3; iy++)
#pragma unroll
        for (int iz = 0; iz < 3; iz++) {
            a[1][iy][iz] = tile[0][ty + iy][tz + iz];
            a[2][iy][iz] = tile[1][ty + iy][tz + iz];
        }
    int mine = 0;
    bool any_tie = false;
    double cmax = -1.7976931348623157e308;   // the largest density of this column (brick potential of the region growth)
#define BM_STEP(K)                                                                                                   \
    {                                                                                                                \
        _Pragma("unroll") for (int iy = 0; iy < 3; iy++) _Pragma("unroll") for (int iz = 0; iz < 3; iz++) {          \
            a[0][iy][iz] = a[1][iy][iz];                                                                             \
            a[1][iy][iz] = a[2][iy][iz];                                                                             \
            a[2][iy][iz] = tile[K + 2][ty + iy][tz + iz];                                                            \
        }                                                                                                            \
        const int x = x0 + K;                                                                                        \
        if (!PART || x < g.nx) cmax = max_raw(cmax, a[1][1][1]);   /* (beyond the grid the tile holds wrapped planes) */ \
        bm_voxel<MT, K, DIAG, PART>(g, a, dface, col_in && x < g.nx, (x * g.ny + y) * g.nz + z, ty, zz, mine, any_tie, \
                                    &s_cnt[tz >> 3], &s_mv[tz >> 3], mu, mirror, co, deep, wx, at_yh, at_zh);          \
    }
    BM_STEP(0) BM_STEP(1) BM_STEP(2) BM_STEP(3) BM_STEP(4) BM_STEP(5) BM_STEP(6) BM_STEP(7)
#undef BM_STEP
    static_assert(GT_X == 8, "eight x-positions per column");
    if (__any(any_tie) && threadIdx.x % XB_WAVE == 0) atomicAdd(tie_count, 1);  // only != 0 matters
    atomicOr(&s_mask[tz >> 3], mine);
    if (bpot && col_in) {   // float order as signed-int order (a potential only steers the growth's guess: precision is irrelevant)
        const int fi = __float_as_int((float)cmax);
        atomicMax(&s_pot[tz >> 3], fi >= 0 ? fi : fi ^ 0x7fffffff);
    }
    __syncthreads();
    if (threadIdx.x < GT_Z / 8 && z0 + threadIdx.x * 8 < g.nz) {
        const int nb1 = (g.ny + 7) >> 3, nb2 = (g.nz + 7) >> 3;
        if (bpot) bpot[((x0 >> 3) * nb1 + (y0 >> 3)) * nb2 + (z0 >> 3) + threadIdx.x] = s_pot[threadIdx.x];
        // bits 0-26: the neighbour bricks a move can reach; bit 27: the brick holds a 26-neighbour maximum, bit 28: two or more
        const int b = ((x0 >> 3) * nb1 + (y0 >> 3)) * nb2 + (z0 >> 3) + threadIdx.x, n = s_cnt[threadIdx.x];
        // (PART: a brick one voxel wide is never certified -- bits 27 + 28 --, see above)
        const bool dead = PART && (g.nx - x0 == 1 || g.ny - y0 == 1 || g.nz - (z0 + 8 * (int)threadIdx.x) == 1);
        bmask[b] = (s_mask[threadIdx.x] & 0x7ffdfff) | ((n >= 1 || dead) ? 1 << 27 : 0) | ((n >= 2 || dead) ? 1 << 28 : 0);
        bmaxv[b] = (n == 1 && !dead) ? s_mv[threadIdx.x] : -1;
    }
}

// pass B: the records of the voxels of the listed bricks.  One workgroup per brick and turn: the 10^3 haloed brick
// goes through LDS, every thread derives two records (k_grad_field's arithmetic).  The brick list is either the walk
// list (`walk`, length *n_list on the device) or, with walk == nullptr, every brick whose brick_rec flag is already
// set (a rebuild under the other tie rule).
// Round 4: (i) everything about a thread's four halo elements and two voxels that does not depend on the brick -- their
// offsets from the brick's corner in the density, in the tile and in the table -- is computed ONCE per thread, before the loop
// over the bricks; a brick whose halo does not touch the faces of the grid (all but a few per cent) then stages its tile with
// one addition per load.  Round 3 redid the divisions and products per element and brick: 65 quarter-rate integer
// multiplications per thread and brick, about a third of the kernel's VALU cycles.  (ii) The tile rows are 24 doubles apart:
// the 32 lanes of a half wave (4 rows of 8 voxels) then fall on 32 distinct bank pairs for every (ix, iy, iz) of the scan,
// whether the compiler reads them with ds_read_b64 or pairs them into ds_read2_b64 (16-lane groups, banks of 16 doubles);
// at a stride of 10 every read met a two-way conflict.  Measured: 0.333 -> 0.319 ms at 512^3.  Tried on top and dropped: the halo of the
// NEXT brick fetched into registers under the scan of the current one (0.336-0.348 ms: the kernel moves 48 B per voxel at
// 3.4 TB/s and is 50-60 % VALU busy -- no single latency to hide), one x-plane of the scan at a time (same registers).
#define BR_ROW 24
template <typename GT>
__global__ __launch_bounds__(TPB, 4) void k_brick_records(GT g, const double *__restrict__ rho, GradRec *__restrict__ G,
                                                       const int *__restrict__ walk, const int *n_list, int nbr, int nb1, int nb2,
                                                       unsigned char *brick_rec, int small) {
    __shared__ double tile[10][10][BR_ROW];
    const int n = walk ? *n_list : nbr;
    int goff[4], loff[4];   // halo element e = threadIdx.x + j * TPB of the 10^3 tile: offset from the corner voxel (no wrap), slot in the tile
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int e = threadIdx.x + j * TPB;
        const int ex = e / 100, ey = (e / 10) % 10, ez = e % 10;
        goff[j] = (ex * g.ny + ey) * g.nz + ez;
        loff[j] = (ex * 10 + ey) * BR_ROW + ez;
    }
    const XcdRange xr = xcd_range(n);   // (neighbouring bricks -- their haloed tiles overlap -- are staged through one L2)
    for (int item = xr.begin; item < xr.end; item += xr.step) {
        const int b = walk ? walk[item] : item;
        if (!walk && !(brick_rec[b] & 1)) continue;   // uniform per block
        const int b0 = b / (nb1 * nb2), b1 = (b / nb2) % nb1, b2 = b % nb2;
        const int x0 = b0 * 8, y0 = b1 * 8, z0 = b2 * 8;
        // the haloed brick lies inside the grid: no periodic wrap anywhere (block uniform)
        const bool inner = !(small & 1) && x0 >= 1 && x0 + 8 < g.nx && y0 >= 1 && y0 + 8 < g.ny && z0 >= 1 && z0 + 8 < g.nz;
        __syncthreads();   // the previous turn's readers are done with the tile
        {
            double val[4];
            if (inner) {
                const double *corner = rho + ((size_t)(x0 - 1) * g.ny + (y0 - 1)) * g.nz + (z0 - 1);
#pragma unroll
                for (int j = 0; j < 4; j++) val[j] = (threadIdx.x + j * TPB < 1000) ? corner[goff[j]] : 0.;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int e = threadIdx.x + j * TPB;
                    const int ex = e / 100, ey = (e / 10) % 10, ez = e % 10;
                    int X = x0 + ex - 1, Y = y0 + ey - 1, Z = z0 + ez - 1;
                    if (small & 1) {
                        X = ((X % g.nx) + g.nx) % g.nx; Y = ((Y % g.ny) + g.ny) % g.ny; Z = ((Z % g.nz) + g.nz) % g.nz;
                    } else {
                        X = wrap_u(X, g.nx); Y = wrap_u(Y, g.ny); Z = wrap_u(Z, g.nz);
                    }
                    val[j] = e < 1000 ? rho[(X * g.ny + Y) * g.nz + Z] : 0.;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (threadIdx.x + j * TPB < 1000) (&tile[0][0][0])[loff[j]] = val[j];
        }
        __syncthreads();
        const int vbase = (x0 * g.ny + y0) * g.nz + z0;
#pragma unroll 1
        for (int j = 0; j < 2; j++) {
            const int l = threadIdx.x + j * TPB;
            const int tx = l >> 6, ty = (l >> 3) & 7, tz = l & 7;
            const double c = tile[tx + 1][ty + 1][tz + 1];
            double max_val = c;
            int og = XB_OG_SELF;
#pragma unroll
            for (int ix = 0; ix < 3; ix++) {
#pragma unroll
                for (int iy = 0; iy < 3; iy++)
#pragma unroll
                    for (int iz = 0; iz < 3; iz++) {
                        double w = tile[tx + ix][ty + iy][tz + iz];
                        w = (w - c) * dist_at(g, ix, iy, iz);
                        w += c;
                        og = (w > max_val) ? ix * 9 + iy * 3 + iz : og;
                        max_val = fmax(max_val, w);
                    }
            }
            GradRec o;
            double d0, d1, d2;
            int code;
            if (ng_dir_vals(g, c, tile[tx + 2][ty + 1][tz + 1], tile[tx][ty + 1][tz + 1], tile[tx + 1][ty + 2][tz + 1],
                            tile[tx + 1][ty][tz + 1], tile[tx + 1][ty + 1][tz + 2], tile[tx + 1][ty + 1][tz], d0, d1, d2)) {
                o.r0 = o.r1 = o.r2 = 0.;
                code = XB_STAY_CODE;
            } else {
                const int i0 = rha_cs(d0), i1 = rha_cs(d1), i2 = rha_cs(d2);
                o.r0 = d0 - (double)i0;
                o.r1 = d1 - (double)i1;
                o.r2 = d2 - (double)i2;
                code = (i0 + 1) | ((i1 + 1) << 2) | ((i2 + 1) << 4);
            }
            o.key = pack_key(c, code, og);
            // (a brick the grid cuts: only its voxels inside the grid have a record)
            if (inner || (x0 + tx < g.nx && y0 + ty < g.ny && z0 + tz < g.nz)) G[rec_slot(g, vbase + (tx * g.ny + ty) * g.nz + tz)] = o;
        }
        if (threadIdx.x == 0) brick_rec[b] |= 1;
    }
}

// refinement without a table from an assignment (after ongrid / uploaded labels): records are needed where retraces
// run, i.e. in the bricks whose 3x3x3 surroundings do not carry one single label (buni3 == XB_MIXED_LABEL); bit 1
// ("may hold a maximum") is set everywhere: nothing is known about the maxima here
#define XB_MIXED_LABEL (-2147483647 - 1)
__global__ void k_flag_mixed_bricks(int nbr, const int *__restrict__ buni3, unsigned char *brick_rec) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nbr) brick_rec[b] = (unsigned char)(2 | (buni3[b] == XB_MIXED_LABEL ? 1 : 0));
}
// A refinement after an ongrid assignment (round 4): the retraces of refinement.py:283-303 stop on the first known == 2 voxel, so
// records are needed for the BAND only (known != 2: the edge voxels and the ring around them) -- the bricks that hold band
// voxels, found from the flags of the listed tiles once the edge sweep has run, instead of every brick whose 27-brick
// surroundings carry two labels (at 512^3 half as many: k_brick_records 0.90 -> 0.45 ms).  Bricks within ONE voxel of the
// band are flagged too: the next iteration's new edges lie inside the boxes of changed edge voxels (refinement.py:428-504)
// and its ring one voxel further; what walks on beyond that is redone by the from-rho kernel as before.
// (the flags go to a byte array of their own first -- plain stores of 1, nothing to wait for -- and are merged into bit 0
// afterwards: read-modify-write of the brick bytes themselves was up to 27 dependent round trips per band row, 62 us)
__global__ void k_rec_set_bit0(int nbr, unsigned char *brick_rec, const unsigned char *__restrict__ flag) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nbr) brick_rec[b] = (unsigned char)((brick_rec[b] & 2) | (flag[b] ? 1 : 0));
}
__global__ __launch_bounds__(TPB) void k_flag_band_bricks(GridL g, const int8_t *__restrict__ known, const int *__restrict__ tiles, const int *n_tiles,
                                                          unsigned char *flag) {
    const int ntz = (g.nz + 63) / 64, nty = (g.ny + 7) / 8, n = *n_tiles;   // the edge sweep's tiles: 4 x 8 x 64 voxels (k_edges.h)
    const int nb0 = (g.nx + 7) >> 3, nb1 = (g.ny + 7) >> 3, nb2 = (g.nz + 7) >> 3;
    for (int item = blockIdx.x; item < n; item += gridDim.x) {
        const int t = (int)((unsigned)tiles[item] & 0x7fffffffu);
        const int x = (t / (ntz * nty)) * 4 + threadIdx.x / 64, y = ((t / ntz) % nty) * 8 + (threadIdx.x / 8) % 8, z = (t % ntz) * 64 + 8 * (threadIdx.x % 8);
        if (x >= g.nx || y >= g.ny || z >= g.nz) continue;
        const int8_t *row = known + ((size_t)x * g.ny + y) * g.nz;
        bool any = false, lo = false, hi = false;   // band voxels in this chunk / at its first / at its last voxel
        if (g.nz % 8 == 0) {
            const unsigned long long w = *reinterpret_cast<const unsigned long long *>(row + z);
            any = w != 0x0202020202020202ull; lo = (w & 0xffull) != 2; hi = (w >> 56) != 2;
        } else
            for (int k = 0; k < 8 && z + k < g.nz; k++) {
                const bool bnd = row[z + k] != 2;
                any |= bnd;
                if (k == 0) lo = bnd;
                if (k == 7 || z + k == g.nz - 1) hi = bnd;
            }
        if (!any) continue;
        const int bx = x >> 3, by = y >> 3, bz = z >> 3;
        const bool mx = (x & 7) == 0, px = (x & 7) == 7 || x == g.nx - 1, my = (y & 7) == 0, py = (y & 7) == 7 || y == g.ny - 1;
        for (int ax = -1; ax < 2; ax++) {
            if ((ax < 0 && !mx) || (ax > 0 && !px)) continue;
            for (int ay = -1; ay < 2; ay++) {
                if ((ay < 0 && !my) || (ay > 0 && !py)) continue;
                for (int az = -1; az < 2; az++) {
                    if ((az < 0 && !lo) || (az > 0 && !hi)) continue;
                    flag[(wrapi(bx + ax, nb0) * nb1 + wrapi(by + ay, nb1)) * nb2 + wrapi(bz + az, nb2)] = 1;
                }
            }
        }
    }
}
// brick_rec[b]: bit 0 = the records of brick b exist, bit 1 = the brick holds a 26-neighbour maximum (k_grow_finish).
// Does the record of voxel (x,y,z) exist?  (nullptr: the table covers the whole grid / window)
__device__ __forceinline__ bool rec_exists(const unsigned char *__restrict__ brick_rec, const GridL &g, int x, int y, int z) {
    return !brick_rec || (brick_rec[((x >> 3) * ((g.ny + 7) >> 3) + (y >> 3)) * ((g.nz + 7) >> 3) + (z >> 3)] & 1) != 0;
}
