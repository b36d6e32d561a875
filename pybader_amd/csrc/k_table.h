// k_table.h -- device kernels of libbader_hip.so: gradient-field table and trapping regions (closed cubes, brick growth).
// Included by bader_hip.hip (one translation unit); see bader_kernels.h for the common device code.
#pragma once

// ---------------------------------------------------------------------------------------------
// Gradient-field table: per voxel the normalised neargrid step direction (refinement.py:89-143)
// split into integer step + remainder, and the ongrid successor (methods.py:87-117), 32 B/voxel.
// LDS-tiled: a block stages a 4x8x64 tile of rho plus a one-voxel periodic halo (6x10x66 doubles)
// and every thread derives 8 records from the staged 3x3x3 neighbourhoods.  Neither quantity
// depends on the carried remainder `dr`, so every trajectory step afterwards is ONE 32-byte gather.
// 26-neighbour maxima (ongrid successor == self) are appended to `seeds`.
// ---------------------------------------------------------------------------------------------
#define GT_X 8
#define GT_Y 8
#define GT_Z 32
// The tile is GT_Z/8 whole 8^3 bricks in a row along z; when the grid is made of whole bricks
// (`bmask` != null) the block also reduces, per brick, which neighbour bricks any possible move of
// its voxels can reach (the k_brick_* kernels below work on these masks alone).
__device__ __forceinline__ void move_ranges_raw(int code, int og, double r0, double r1, double r2, int lo[3], int hi[3]) {
    lo[0] = hi[0] = og / 9 - 1; lo[1] = hi[1] = (og / 3) % 3 - 1; lo[2] = hi[2] = og % 3 - 1;
    if (code != XB_STAY_CODE) {
        const int i0 = (code & 3) - 1, i1 = ((code >> 2) & 3) - 1, i2 = (code >> 4) - 1;
        lo[0] = min(lo[0], i0 - (r0 < 1e-12)); hi[0] = max(hi[0], i0 + (r0 > -1e-12));
        lo[1] = min(lo[1], i1 - (r1 < 1e-12)); hi[1] = max(hi[1], i1 + (r1 > -1e-12));
        lo[2] = min(lo[2], i2 - (r2 < 1e-12)); hi[2] = max(hi[2], i2 + (r2 > -1e-12));
    }
}
template <typename GT>
__global__ __launch_bounds__(TPB) void k_grad_field(GT g, const double *__restrict__ rho,
                                                    GradRec *__restrict__ G, int *seeds, int *seed_count,
                                                    int seed_cap, int small, int *__restrict__ bmask, int *tie_count) {
    __shared__ double tile[GT_X + 2][GT_Y + 2][GT_Z + 2];
    __shared__ int s_mask[GT_Z / 8];
    // plane tiles are counted from the start of the table window (brick aligned; the whole grid on one GPU)
    int x0 = g.wx0 + blockIdx.z * GT_X;
    if (x0 >= g.nx) x0 -= g.nx;
    const int y0 = blockIdx.y * GT_Y, z0 = blockIdx.x * GT_Z;
    if (threadIdx.x < GT_Z / 8) s_mask[threadIdx.x] = 0;
    {
        // row-wise staging: a wave takes whole z-rows of the haloed tile (x,y wrap is wave-uniform scalar
        // work, the z wrap is done once per lane), ~6 instructions per row instead of a div/mod chain per element
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / XB_WAVE), lane = threadIdx.x % XB_WAVE;
        int Z = z0 + lane - 1;
        if (small & 1) Z = ((Z % g.nz) + g.nz) % g.nz;
        else Z = wrap_u(Z, g.nz);
        // all of a wave's row loads are issued before the first wait (the unrolled loop keeps ROWS loads in
        // flight; issued one at a time the kernel was bound by ROWS serial HBM latencies per block)
        constexpr int ROWS = (GT_X + 2) * (GT_Y + 2) / (TPB / XB_WAVE);
        static_assert(ROWS * (TPB / XB_WAVE) == (GT_X + 2) * (GT_Y + 2), "rows must divide evenly over the waves");
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
    const int tz = threadIdx.x & (GT_Z - 1), ty = threadIdx.x / GT_Z;   // 32 x 8 threads, 8 voxels (x) each
    int mine = 0;  // move mask of this thread's voxels (all in brick tz >> 3 of the tile)
    bool any_tie = false;  // a voxel whose record depends on the tie rule (methods.py:324 vs refinement.py:111)
#pragma unroll 1
    for (int k = 0; k < GT_X; k++) {
        const int tx = k;
        const int x = x0 + tx, y = y0 + ty, z = z0 + tz;
        if (x >= g.nx || y >= g.ny || z >= g.nz) continue;
        const int v = (x * g.ny + y) * g.nz + z;
        const double c = tile[tx + 1][ty + 1][tz + 1];
        // ongrid successor: strict '>' first-wins scan in (ix,iy,iz) ascending order
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
                    max_val = fmax(max_val, w);  // no NaNs in a density: same as the conditional assignment
                }
        GradRec o;
        double d0, d1, d2;
        int code;
        any_tie |= ((int)axis_tie(tile[tx + 2][ty + 1][tz + 1], c, tile[tx][ty + 1][tz + 1]) |
                    (int)axis_tie(tile[tx + 1][ty + 2][tz + 1], c, tile[tx + 1][ty][tz + 1]) |
                    (int)axis_tie(tile[tx + 1][ty + 1][tz + 2], c, tile[tx + 1][ty + 1][tz])) != 0;
        if (ng_dir_vals(g, c, tile[tx + 2][ty + 1][tz + 1], tile[tx][ty + 1][tz + 1], tile[tx + 1][ty + 2][tz + 1],
                        tile[tx + 1][ty][tz + 1], tile[tx + 1][ty + 1][tz + 2], tile[tx + 1][ty + 1][tz], d0, d1, d2)) {
            // max_grad < 1E-14: a trajectory stays on p, p is on its path, so the reference resets dr
            // and takes the ongrid step (refinement.py:200-235) -- which is the tabulated successor
            o.r0 = o.r1 = o.r2 = 0.;
            code = XB_STAY_CODE;
        } else {
            // refinement.py:138-143: int_grad = rha(grad_dir); the remainder grad_dir - int_grad is what
            // every trajectory through p adds to its dr
            const int i0 = rha_cs(d0), i1 = rha_cs(d1), i2 = rha_cs(d2);
            o.r0 = d0 - (double)i0;
            o.r1 = d1 - (double)i1;
            o.r2 = d2 - (double)i2;
            code = (i0 + 1) | ((i1 + 1) << 2) | ((i2 + 1) << 4);
        }
        o.key = pack_key(c, code, og);
        G[rec_slot(g, v)] = o;
        if (og == XB_OG_SELF) {
            const int q = atomicAdd(seed_count, 1);
            if (q < seed_cap) seeds[q] = v;
            mine |= 1 << 27;
        }
        if (bmask) {  // which neighbour bricks can a move from this voxel reach (moves <= 2 voxels)
            int lo[3], hi[3];
            move_ranges_raw(code, og, o.r0, o.r1, o.r2, lo, hi);
            // per axis the set of brick offsets {-1,0,+1} a move can reach (0 always, as bits 0..2), then the
            // 27-bit outer product by two carry-free multiplications
            const int pa = 2 | (tx + lo[0] < 0) | ((tx + hi[0] >= 8) << 2);
            const int pb = 2 | (ty + lo[1] < 0) | ((ty + hi[1] >= 8) << 2);
            const int pc = 2 | ((tz & 7) + lo[2] < 0) | (((tz & 7) + hi[2] >= 8) << 2);
            const int yz = pc * (8 | (pb & 1) | ((pb & 4) << 4));
            mine |= yz * (512 | (pa & 1) | ((pa & 4) << 16));
        }
    }
    if (__any(any_tie) && threadIdx.x % XB_WAVE == 0) atomicAdd(tie_count, 1);  // only != 0 matters
    if (bmask) {
        atomicOr(&s_mask[tz >> 3], mine);
        __syncthreads();
        if (threadIdx.x < GT_Z / 8 && z0 + threadIdx.x * 8 < g.nz) {
            const int nb1 = g.ny >> 3, nb2 = g.nz >> 3;
            bmask[((x0 >> 3) * nb1 + (y0 >> 3)) * nb2 + (z0 >> 3) + threadIdx.x] = s_mask[threadIdx.x] & ~(1 << 13);
        }
    }
}

// The ongrid pointer of every voxel (methods.py:84-117) from the same staged tile: labels[v] = linear
// index of the best distance-weighted neighbour (v itself for a 26-neighbour maximum); vacuum voxels
// (label -1) keep their -1 (methods.py:73-74).
// With `bmask` (grids of whole 8^3 bricks) the pass also reduces, per brick, which neighbour bricks the ongrid
// move of any of its voxels enters (bit 27: the brick holds a 26-neighbour maximum) and lists the maxima, so
// that the trapping regions of the ongrid pointer field can be grown exactly like the neargrid ones.
template <typename GT>
__global__ __launch_bounds__(TPB) void k_og_pointer_tiled(GT g, const double *__restrict__ rho, int *labels,
                                                          int small, int has_vacuum, int *seeds, int *seed_count,
                                                          int seed_cap, int *__restrict__ bmask) {
    __shared__ double tile[GT_X + 2][GT_Y + 2][GT_Z + 2];
    __shared__ int s_mask[GT_Z / 8];
    if (threadIdx.x < GT_Z / 8) s_mask[threadIdx.x] = 0;
    const int x0 = blockIdx.z * GT_X, y0 = blockIdx.y * GT_Y, z0 = blockIdx.x * GT_Z;
    {
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
    const int tz = threadIdx.x & (GT_Z - 1), ty = threadIdx.x / GT_Z;
    int mine = 0;
#pragma unroll 1
    for (int tx = 0; tx < GT_X; tx++) {
        const int x = x0 + tx, y = y0 + ty, z = z0 + tz;
        if (x >= g.nx || y >= g.ny || z >= g.nz) continue;
        const int v = (x * g.ny + y) * g.nz + z;
        if (has_vacuum && labels[v] == -1) continue;
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
        const int qx = wrapi(x + og / 9 - 1, g.nx), qy = wrapi(y + (og / 3) % 3 - 1, g.ny), qz = wrapi(z + og % 3 - 1, g.nz);
        labels[v] = (qx * g.ny + qy) * g.nz + qz;
        if (bmask) {
            if (og == XB_OG_SELF) {
                const int q = atomicAdd(seed_count, 1);
                if (q < seed_cap) seeds[q] = v;
                mine |= 1 << 27;
            }
            const int a = tx + og / 9 - 1, b = ty + (og / 3) % 3 - 1, c2 = (tz & 7) + og % 3 - 1;
            const int k0 = a < 0 ? 0 : (a >= 8 ? 2 : 1), k1 = b < 0 ? 0 : (b >= 8 ? 2 : 1), k2 = c2 < 0 ? 0 : (c2 >= 8 ? 2 : 1);
            mine |= 1 << (k0 * 9 + k1 * 3 + k2);
        }
    }
    if (bmask) {
        atomicOr(&s_mask[tz >> 3], mine);
        __syncthreads();
        if (threadIdx.x < GT_Z / 8 && z0 + threadIdx.x * 8 < g.nz) {
            const int nb1 = g.ny >> 3, nb2 = g.nz >> 3;
            bmask[((x0 >> 3) * nb1 + (y0 >> 3)) * nb2 + (z0 >> 3) + threadIdx.x] = s_mask[threadIdx.x] & ~(1 << 13);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Trapping boxes.  For a 26-neighbour maximum m let B_R = {v : |v - m|_inf <= R} (minimum image).
// B_R is CLOSED when no voxel of B_R can be left by (a) a neargrid move, for ANY carried remainder dr,
// or (b) an ongrid move.  (a): per axis the move is int_grad + corr with corr = rha(dr + r),
// |dr| <= 0.5 (+1 ulp): corr can be +1 only if r >= 0 and -1 only if r <= 0 (both when |r| < 1e-12),
// so the reachable offsets are a per-axis interval read off the table record.  If B_R is closed and
// m is its only 26-neighbour maximum, every trajectory that arrives at a voxel of B_R ends at m --
// exactly, whatever its dr -- so the trace may stop there.  Moves are at most 2 voxels long, so a
// voxel at distance d whose farthest successor is at distance D only violates the boxes d <= R < D.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int min_image_abs(int t, int n) {
    int a = t < 0 ? -t : t;
    if (a >= n) a -= n;
    return min(a, n - a);
}
// Per-axis interval of the offsets any move from this voxel can have: the ongrid move plus the
// conservative set of neargrid moves (see above).
__device__ __forceinline__ void move_ranges(const GradRec &rec, int lo[3], int hi[3]) {
    const int code = key_code(rec.key), og = key_og(rec.key);
    lo[0] = hi[0] = og / 9 - 1; lo[1] = hi[1] = (og / 3) % 3 - 1; lo[2] = hi[2] = og % 3 - 1;
    if (code != XB_STAY_CODE) {
        const int i0 = (code & 3) - 1, i1 = ((code >> 2) & 3) - 1, i2 = (code >> 4) - 1;
        lo[0] = min(lo[0], i0 - (rec.r0 < 1e-12)); hi[0] = max(hi[0], i0 + (rec.r0 > -1e-12));
        lo[1] = min(lo[1], i1 - (rec.r1 < 1e-12)); hi[1] = max(hi[1], i1 + (rec.r1 > -1e-12));
        lo[2] = min(lo[2], i2 - (rec.r2 < 1e-12)); hi[2] = max(hi[2], i2 + (rec.r2 > -1e-12));
    }
}
__device__ __forceinline__ int wrap_any(int v, int n) { v %= n; return v < 0 ? v + n : v; }

// The same move intervals derived from rho directly (no table record needed): used for the seed cubes
// when the table only covers a window of the grid (slabs).
__device__ __forceinline__ void move_ranges_rho(const double *__restrict__ rho, const Grid &g, int x, int y, int z,
                                                int lo[3], int hi[3]) {
    const int v = lin3(g, x, y, z);
    const double c = rho[v];
    double max_val = c;
    int og = XB_OG_SELF;
    for (int ix = 0; ix < 3; ix++) {
        const int tx = wrapi(x + ix - 1, g.nx);
        for (int iy = 0; iy < 3; iy++) {
            const int ty = wrapi(y + iy - 1, g.ny);
            for (int iz = 0; iz < 3; iz++) {
                const int tz = wrapi(z + iz - 1, g.nz);
                double w = rho[lin3(g, tx, ty, tz)];
                w = (w - c) * g.dist[((ix + 2) % 3) * 9 + ((iy + 2) % 3) * 3 + ((iz + 2) % 3)];
                w += c;
                if (w > max_val) { max_val = w; og = ix * 9 + iy * 3 + iz; }
            }
        }
    }
    double d0, d1, d2;
    if (ng_dir(rho, g, x, y, z, v, c, d0, d1, d2)) move_ranges_raw(XB_STAY_CODE, og, 0., 0., 0., lo, hi);
    else {
        const int i0 = rha_cs(d0), i1 = rha_cs(d1), i2 = rha_cs(d2);
        move_ranges_raw((i0 + 1) | ((i1 + 1) << 2) | ((i2 + 1) << 4), og, d0 - (double)i0, d1 - (double)i1,
                        d2 - (double)i2, lo, hi);
    }
}
__global__ __launch_bounds__(TPB) void k_box_shells_rho(Grid g, const double *__restrict__ rho,
                                                        const int *__restrict__ mxyz, const int *__restrict__ rcap,
                                                        int rlo, int K, int *bad, int stride) {
    const int m = blockIdx.y;
    const int rhi = min(rlo + K, rcap[m]);
    if (rhi < rlo) return;
    const int w = 2 * rhi + 1;
    const long long t = (long long)blockIdx.x * TPB + threadIdx.x;
    if (t >= (long long)w * w * w) return;
    const int o[3] = {(int)(t / ((long long)w * w)) - rhi, (int)((t / w) % w) - rhi, (int)(t % w) - rhi};
    const int d = max(max(abs(o[0]), abs(o[1])), abs(o[2]));
    if (d < rlo) return;
    int lo[3], hi[3];
    move_ranges_rho(rho, g, wrap_any(mxyz[3 * m] + o[0], g.nx), wrap_any(mxyz[3 * m + 1] + o[1], g.ny),
                    wrap_any(mxyz[3 * m + 2] + o[2], g.nz), lo, hi);
    int D = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) D = max(D, max(abs(o[j] + lo[j]), abs(o[j] + hi[j])));
    for (int R = d; R < D; R++) bad[m * stride + R] = 1;
}

// Closed cubes around the maxima, found in batches of K shells: the launch visits, for box m, the
// voxels at L-inf distance d in [rlo, rlo+K] of the maximum.  A voxel at distance d whose farthest
// successor is at distance D violates the cubes d <= R < D (moves are at most 2 voxels long, so
// only the two outer shells of a cube can violate it).
__global__ __launch_bounds__(TPB) void k_box_shells(GridL g, const GradRec *__restrict__ G,
                                                    const int *__restrict__ mxyz, const int *__restrict__ rcap,
                                                    int rlo, int K, int *bad, int stride) {
    const int m = blockIdx.y;
    const int rhi = min(rlo + K, rcap[m]);
    if (rhi < rlo) return;
    const int w = 2 * rhi + 1;
    const long long t = (long long)blockIdx.x * TPB + threadIdx.x;
    if (t >= (long long)w * w * w) return;
    const int o[3] = {(int)(t / ((long long)w * w)) - rhi, (int)((t / w) % w) - rhi, (int)(t % w) - rhi};
    const int d = max(max(abs(o[0]), abs(o[1])), abs(o[2]));
    if (d < rlo) return;
    const int x = wrap_any(mxyz[3 * m] + o[0], g.nx), y = wrap_any(mxyz[3 * m + 1] + o[1], g.ny),
              z = wrap_any(mxyz[3 * m + 2] + o[2], g.nz);
    const GradRec rec = fetch_rec_w(g, G, (x * g.ny + y) * g.nz + z);
    int lo[3], hi[3];
    move_ranges(rec, lo, hi);
    int D = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) D = max(D, max(abs(o[j] + lo[j]), abs(o[j] + hi[j])));
    for (int R = d; R < D; R++) bad[m * stride + R] = 1;
}
// stamp box id `id` into the key of every voxel of the cube B_R(m)
__global__ __launch_bounds__(TPB) void k_box_stamp(GridL g, GradRec *G, int mx, int my, int mz, int R, int id) {
    const int w = 2 * R + 1;
    const long long n = (long long)w * w * w;
    const long long t = (long long)blockIdx.x * TPB + threadIdx.x;
    if (t >= n) return;
    const int dz = (int)(t % w), dy = (int)((t / w) % w), dx = (int)(t / ((long long)w * w));
    const int x = wrap_any(mx + dx - R, g.nx), y = wrap_any(my + dy - R, g.ny), z = wrap_any(mz + dz - R, g.nz);
    long long *kp = reinterpret_cast<long long *>(&G[rec_slot(g, (x * g.ny + y) * g.nz + z)].key);
    *kp = (*kp & ~(0x3FFLL << 11)) | ((long long)id << 11);
}

// ---------------------------------------------------------------------------------------------
// Growing the trapping regions brick by brick (8x8x8 voxels).  Let U be a union of sets certain
// for maximum m (closed boxes, earlier bricks).  A brick B without a 26-neighbour maximum whose
// every possible move (any dr) from every voxel lands in B itself or in bricks that are certain
// for the SAME m keeps U + B closed, and a trajectory cannot stay in B forever (it only ends on
// a maximum), so it must enter U: B is certain for m as well.  Mutually dependent bricks are certified
// together by a greatest-fixpoint (kill) iteration on provisional labels, see k_brick_grow.
// ---------------------------------------------------------------------------------------------
#define BRK 8
// blab: 0 unknown, id > 0 certain for box id, -1 never (holds a maximum)
__global__ void k_brick_seed(GridL g, int nb0, int nb1, int nb2, int n_boxes, const int *__restrict__ mxyz,
                             const int *__restrict__ radius, int *blab) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb0 * nb1 * nb2) return;
    const int b0 = b / (nb1 * nb2), b1 = (b / nb2) % nb1, b2 = b % nb2;
    int lab = 0;
    for (int m = 0; m < n_boxes; m++) {
        const int R = radius[m];
        // the brick [8b, 8b+7] lies inside the cube iff both ends are within R of the maximum on
        // every axis (minimum image; boxes never wrap onto themselves)
        bool in = true;
        const int n3[3] = {g.nx, g.ny, g.nz}, bb[3] = {b0, b1, b2};
#pragma unroll
        for (int j = 0; j < 3; j++) {
            int lo = bb[j] * BRK - mxyz[3 * m + j];
            lo = ((lo % n3[j]) + n3[j]) % n3[j];
            if (lo > n3[j] / 2) lo -= n3[j];
            in &= (lo >= -R) && (lo + BRK - 1 <= R);
        }
        if (in) lab = m + 1;
    }
    blab[b] = lab;
}
// bmask[K] (built by k_grad_field): bit k (k = (d0+1)*9+(d1+1)*3+(d2+1), d = brick offset) is set
// when some possible move of some voxel of brick K lands in the neighbour brick K+d; bit 27 is set
// when the brick holds a 26-neighbour maximum.
// The two iterations (k_brick_grow below):
//  * provisional labels: an unlabelled brick adopts the label of a labelled brick it can move into (smallest
//    label on ties).  Any guess is sound -- the kill iteration decides -- a good guess only makes the
//    certain regions larger;
//  * kill (greatest fixpoint): a non-seed brick stays alive for its label m only while it holds no maximum
//    and every brick it can move into is alive with the same label.  What survives, together with the seed
//    cubes, is closed under every possible move: a trapping region of m.
// Both iterations, several rounds per launch: a workgroup keeps an 8x8x8 chunk of bricks plus a one-brick
// (periodic) halo in LDS and iterates on it until nothing changes or `inner` rounds are done; the halo is
// what the previous launch left.  Labels then travel up to `inner` bricks per launch instead of one.
// The schedule does not matter for the result's soundness: a provisional label is only a guess, and the
// kill iteration is monotone (stale neighbour labels can only delay a kill), so its fixpoint -- reached
// when a whole launch changes nothing -- is the same greatest fixpoint.  phase 0: propagate, 1: kill.
#define BG 8
__global__ __launch_bounds__(BG * BG * BG) void k_brick_grow(int nb0, int nb1, int nb2, const int *__restrict__ bmask,
                                                             const int *__restrict__ seed, const int *__restrict__ in,
                                                             int *__restrict__ out, int *changed, int phase, int inner) {
    __shared__ int lab[2][BG + 2][BG + 2][BG + 2];
    const int c0 = blockIdx.z * BG, c1 = blockIdx.y * BG, c2 = blockIdx.x * BG;
    for (int i = threadIdx.x; i < (BG + 2) * (BG + 2) * (BG + 2); i += BG * BG * BG) {
        const int e2 = i % (BG + 2), e1 = (i / (BG + 2)) % (BG + 2), e0 = i / ((BG + 2) * (BG + 2));
        const int l = in[(wrap_any(c0 + e0 - 1, nb0) * nb1 + wrap_any(c1 + e1 - 1, nb1)) * nb2 + wrap_any(c2 + e2 - 1, nb2)];
        lab[0][e0][e1][e2] = l;
        lab[1][e0][e1][e2] = l;
    }
    const int t2 = threadIdx.x % BG, t1 = (threadIdx.x / BG) % BG, t0 = threadIdx.x / (BG * BG);
    const int b0 = c0 + t0, b1 = c1 + t1, b2 = c2 + t2;
    const bool active = b0 < nb0 && b1 < nb1 && b2 < nb2;
    const int b = active ? (b0 * nb1 + b1) * nb2 + b2 : 0;
    const int m = active ? bmask[b] : 0;
    // bricks that never change: a maximum inside (propagate), seed cubes (kill)
    const bool fixed = !active || (phase == 0 && (m >> 27)) || (phase == 1 && seed[b] != 0);
    __syncthreads();
    const int first = lab[0][t0 + 1][t1 + 1][t2 + 1];
    int l = first, cur = 0;
    for (int it = 0; it < inner; it++) {
        int nl = l;
        if (!fixed) {
            if (phase == 0) {
                if (l == 0) {
                    int best = 0;
                    for (int k = 0; k < 27; k++)
                        if ((m >> k) & 1) {
                            const int q = lab[cur][t0 + k / 9][t1 + (k / 3) % 3][t2 + k % 3];
                            if (q > 0 && (best == 0 || q < best)) best = q;
                        }
                    nl = best;
                }
            } else if (l > 0) {
                bool ok = !(m >> 27);
                for (int k = 0; k < 27 && ok; k++)
                    if ((m >> k) & 1) ok = (lab[cur][t0 + k / 9][t1 + (k / 3) % 3][t2 + k % 3] == l);
                if (!ok) nl = 0;
            }
        }
        lab[cur ^ 1][t0 + 1][t1 + 1][t2 + 1] = nl;
        const int any = __syncthreads_or(nl != l);
        l = nl;
        cur ^= 1;
        if (!any) break;
    }
    if (active) {
        out[b] = l;
        if (l != first) *changed = 1;
    }
}
__global__ __launch_bounds__(TPB) void k_count_positive(const int *__restrict__ a, int n, int *count) {
    int cnt = 0;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) cnt += (a[i] > 0);
    int total;
    block_scan_excl(cnt, total);
    if (threadIdx.x == 0 && total) atomicAdd(count, total);
}
