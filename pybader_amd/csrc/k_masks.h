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

// Conservative move interval of one axis without the division (see move_ranges_raw for the exact form):
// with d = grad_dir component, m = max |component| (>= 1e-14), the reference steps int_grad + rha(dr + r),
// int_grad = rha(d/m), r = d/m - int_grad, |dr| <= 1/2:
//   d/m in [1e-12, 1-1e-12]   -> offsets {0, +1}            (int_grad 0 or 1, the correction can only go the other way)
//   d/m > 1 - 1e-12 (max axis) -> {0, +1, +2}
//   |d/m| < 1e-12              -> {-1, 0, +1}
// and mirrored for negative d.  The thresholds below are 2e-12: a superset of the exact intervals in the two slivers
// of width 1e-12, equal everywhere else (soundness only needs a superset).
#define BM_EPS 2e-12

// thread -> (y, z) column of the 8 x 32 tile face: the columns on a y- or z-face of their brick (28 of 64 per brick)
// come first, so that waves 0-1 hold the border columns (+16 interior ones) and waves 2-3 only interior columns --
// an interior column needs the 27-point ongrid scan on the two x-faces of the brick only.
__device__ __forceinline__ void bm_column(int t, int &ty, int &tz) {
    if (t < 112) {
        const int bz = t / 28, i = t - bz * 28;
        int yy, zz;
        if (i < 8) { yy = 0; zz = i; }
        else if (i < 16) { yy = 7; zz = i - 8; }
        else { yy = 1 + ((i - 16) >> 1); zz = ((i - 16) & 1) ? 7 : 0; }
        ty = yy; tz = bz * 8 + zz;
    } else {
        const int u = t - 112;
        const int bz = u / 36, i = u - bz * 36;
        ty = 1 + i / 6; tz = bz * 8 + 1 + i % 6;
    }
}

template <typename GT>
__global__ __launch_bounds__(TPB) void k_brick_masks(GT g, const double *__restrict__ rho, int *seeds, int *seed_count,
                                                     int seed_cap, int small, int *__restrict__ bmask, int *tie_count) {
    __shared__ double tile[GT_X + 2][GT_Y + 2][GT_Z + 2];
    __shared__ int s_mask[GT_Z / 8];
    const int x0 = blockIdx.z * GT_X, y0 = blockIdx.y * GT_Y, z0 = blockIdx.x * GT_Z;
    if (threadIdx.x < GT_Z / 8) s_mask[threadIdx.x] = 0;
    {   // row-wise staging, every load of a wave in flight before the first wait (see k_grad_field)
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / XB_WAVE), lane = threadIdx.x % XB_WAVE;
        int Z = z0 + lane - 1;
        if (small & 1) Z = ((Z % g.nz) + g.nz) % g.nz;
        else Z = wrap_u(Z, g.nz);
        constexpr int ROWS = (GT_X + 2) * (GT_Y + 2) / (TPB / XB_WAVE);
        double val[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
            const int r = wv + k * (TPB / XB_WAVE);
            const int ex = r / (GT_Y + 2), ey = r - ex * (GT_Y + 2);
            int X = x0 + ex - 1, Y = y0 + ey - 1;
            if (small & 1) {
                X = ((X % g.nx) + g.nx) % g.nx; Y = ((Y % g.ny) + g.ny) % g.ny;
            } else {
                X = wrap_u(X, g.nx); Y = wrap_u(Y, g.ny);
            }
            val[k] = (lane < GT_Z + 2) ? rho[(X * g.ny + Y) * g.nz + Z] : 0.;
        }
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
            const int r = wv + k * (TPB / XB_WAVE);
            const int ex = r / (GT_Y + 2), ey = r - ex * (GT_Y + 2);
            if (lane < GT_Z + 2) tile[ex][ey][lane] = val[k];
        }
    }
    __syncthreads();
    int ty, tz;
    bm_column(threadIdx.x, ty, tz);
    const int zz = tz & 7;
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
#pragma unroll
    for (int k = 0; k < GT_X; k++) {
#pragma unroll
        for (int iy = 0; iy < 3; iy++)
#pragma unroll
            for (int iz = 0; iz < 3; iz++) {
                a[0][iy][iz] = a[1][iy][iz];
                a[1][iy][iz] = a[2][iy][iz];
                a[2][iy][iz] = tile[k + 2][ty + iy][tz + iz];
            }
        const int x = x0 + k;
        const bool in = col_in && x < g.nx;
        const double c = a[1][1][1];
        const double hx = a[2][1][1], lx = a[0][1][1], hy = a[1][2][1], ly = a[1][0][1], hz = a[1][1][2], lz = a[1][1][0];
        const bool tie = ((int)axis_tie(hx, c, lx) | (int)axis_tie(hy, c, ly) | (int)axis_tie(hz, c, lz)) != 0;
        any_tie |= in && tie;
        int lo0 = 1, hi0 = -1, lo1 = 1, hi1 = -1, lo2 = 1, hi2 = -1;   // empty: max_grad < 1E-14 moves by the ongrid step only
        // a voxel with a tie axis has TWO gradient directions (methods.py:324 zeroes the axis, refinement.py:111 does
        // not): its interval is the union, so that a trapping region is closed for the assignment's walkers AND for
        // the refinement's retraces (k_refine_trace stops a retrace that enters a region)
        const int rules = __any(tie) ? 2 : 1;
        for (int rule = 0; rule < rules; rule++) {
            const int mt = rule ? !g.main_ties : g.main_ties;
            // methods.py:324-327 / refinement.py:111-130: the gradient direction before its normalisation
            const double g0 = axis_flat(mt, hx, c, lx) ? 0. : (hx - lx) / 2.;
            const double g1 = axis_flat(mt, hy, c, ly) ? 0. : (hy - ly) / 2.;
            const double g2 = axis_flat(mt, hz, c, lz) ? 0. : (hz - lz) / 2.;
            const double d0 = ((g.T[0] * g0) + (g.T[1] * g1)) + (g.T[2] * g2);
            const double d1 = ((g.T[3] * g0) + (g.T[4] * g1)) + (g.T[5] * g2);
            const double d2 = ((g.T[6] * g0) + (g.T[7] * g1)) + (g.T[8] * g2);
            const double mg = fmax(fmax(fabs(d0), fabs(d1)), fabs(d2));
            if (!(mg < 1E-14)) {
                const double tiny = mg * BM_EPS, nearm = mg * (1. - BM_EPS);
                lo0 = min(lo0, -(int)(d0 < tiny) - (int)(d0 < -nearm)); hi0 = max(hi0, (int)(d0 > -tiny) + (int)(d0 > nearm));
                lo1 = min(lo1, -(int)(d1 < tiny) - (int)(d1 < -nearm)); hi1 = max(hi1, (int)(d1 > -tiny) + (int)(d1 > nearm));
                lo2 = min(lo2, -(int)(d2 < tiny) - (int)(d2 < -nearm)); hi2 = max(hi2, (int)(d2 > -tiny) + (int)(d2 > nearm));
            }
        }
        // Does this voxel need its exact ongrid successor?  (a) it lies on a face of its brick that its gradient
        // interval does not cross already (an ongrid move is one voxel long: only face voxels can leave the brick
        // by it); (b) it may be a 26-neighbour maximum: not ruled out by a face neighbour whose weighted value
        // (bounded from below with the smallest face distance; fl(.) is monotone) exceeds c.
        const double r6 = fmax(fmax(fmax(hx, lx), fmax(hy, ly)), fmax(hz, lz));
        double wq = (r6 - c) * dface;
        wq += c;
        const bool not_max = wq > c;
        const bool need = ((k == 0 && lo0 > -1) || (k == GT_X - 1 && hi0 < 1) || (ty == 0 && lo1 > -1) || (ty == 7 && hi1 < 1) ||
                           (zz == 0 && lo2 > -1) || (zz == 7 && hi2 < 1) || !not_max);
        int og = -1;   // unknown (and irrelevant)
        if (__any(need && in)) {
            // methods.py:87-117: strict '>' first-wins scan in (ix,iy,iz) ascending order
            double max_val = c;
            og = XB_OG_SELF;
#pragma unroll
            for (int ix = 0; ix < 3; ix++)
#pragma unroll
                for (int iy = 0; iy < 3; iy++)
#pragma unroll
                    for (int iz = 0; iz < 3; iz++) {
                        double w = a[ix][iy][iz];
                        w = (w - c) * dist_at(g, ix, iy, iz);
                        w += c;
                        og = (w > max_val) ? ix * 9 + iy * 3 + iz : og;
                        max_val = fmax(max_val, w);
                    }
            const int ox = og / 9 - 1, oy = (og / 3) % 3 - 1, oz = og % 3 - 1;
            lo0 = min(lo0, ox); hi0 = max(hi0, ox);
            lo1 = min(lo1, oy); hi1 = max(hi1, oy);
            lo2 = min(lo2, oz); hi2 = max(hi2, oz);
        }
        if (in) {
            if (og == XB_OG_SELF) {
                const int q = atomicAdd(seed_count, 1);
                if (q < seed_cap) seeds[q] = (x * g.ny + y) * g.nz + z;
                mine |= 1 << 27;
            }
            // per axis the set of brick offsets {-1,0,+1} a move can reach (0 always), then the 27-bit outer product
            const int pa = 2 | (k + lo0 < 0) | ((k + hi0 >= 8) << 2);
            const int pb = 2 | (ty + lo1 < 0) | ((ty + hi1 >= 8) << 2);
            const int pc = 2 | (zz + lo2 < 0) | ((zz + hi2 >= 8) << 2);
            const int yz = pc * (8 | (pb & 1) | ((pb & 4) << 4));
            mine |= yz * (512 | (pa & 1) | ((pa & 4) << 16));
        }
    }
    if (__any(any_tie) && threadIdx.x % XB_WAVE == 0) atomicAdd(tie_count, 1);  // only != 0 matters
    atomicOr(&s_mask[tz >> 3], mine);
    __syncthreads();
    if (threadIdx.x < GT_Z / 8 && z0 + threadIdx.x * 8 < g.nz) {
        const int nb1 = g.ny >> 3, nb2 = g.nz >> 3;
        bmask[((x0 >> 3) * nb1 + (y0 >> 3)) * nb2 + (z0 >> 3) + threadIdx.x] = s_mask[threadIdx.x] & ~(1 << 13);
    }
}

// pass B: the records of the voxels of the listed bricks.  One workgroup per brick and turn: the 10^3 haloed brick
// goes through LDS, every thread derives two records (k_grad_field's arithmetic).  The brick list is either the walk
// list (`walk`, length *n_list on the device) or, with walk == nullptr, every brick whose brick_rec flag is already
// set (a rebuild under the other tie rule).
template <typename GT>
__global__ __launch_bounds__(TPB) void k_brick_records(GT g, const double *__restrict__ rho, GradRec *__restrict__ G,
                                                       const int *__restrict__ walk, const int *n_list, int nbr, int nb1, int nb2,
                                                       unsigned char *brick_rec, int small) {
    __shared__ double tile[10][10][10];
    const int n = walk ? *n_list : nbr;
    for (int item = blockIdx.x; item < n; item += gridDim.x) {
        const int b = walk ? walk[item] : item;
        if (!walk && !(brick_rec[b] & 1)) continue;   // uniform per block
        const int b0 = b / (nb1 * nb2), b1 = (b / nb2) % nb1, b2 = b % nb2;
        const int x0 = b0 * 8, y0 = b1 * 8, z0 = b2 * 8;
        __syncthreads();   // the previous turn's readers are done with the tile
        {
            double val[4];
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
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int e = threadIdx.x + j * TPB;
                if (e < 1000) (&tile[0][0][0])[e] = val[j];
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int j = 0; j < 2; j++) {
            const int l = threadIdx.x + j * TPB;
            const int tx = l >> 6, ty = (l >> 3) & 7, tz = l & 7;
            const double c = tile[tx + 1][ty + 1][tz + 1];
            double max_val = c;
            int og = XB_OG_SELF;
#pragma unroll
            for (int ix = 0; ix < 3; ix++)
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
            G[((x0 + tx) * g.ny + (y0 + ty)) * g.nz + z0 + tz] = o;
        }
        if (threadIdx.x == 0) brick_rec[b] |= 1;
    }
}

// brick_rec[b]: bit 0 = the records of brick b exist, bit 1 = the brick holds a 26-neighbour maximum (k_grow_finish).
// Does the record of voxel (x,y,z) exist?  (nullptr: the table covers the whole grid / window)
__device__ __forceinline__ bool rec_exists(const unsigned char *__restrict__ brick_rec, const GridL &g, int x, int y, int z) {
    return !brick_rec || (brick_rec[((x >> 3) * (g.ny >> 3) + (y >> 3)) * (g.nz >> 3) + (z >> 3)] & 1) != 0;
}
