// bader_hip.hip -- libbader_hip.so: HIP kernels + C ABI (include/bader_hip.h) for gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see pybader_amd/build.py).
#include "bader_kernels.h"
#include "../../include/bader_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

// =============================================================================================
// kernels
// =============================================================================================
#define TPB 256

template <typename T>
__global__ void k_fill(T *p, T v, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// Workload generator, bit-identical to pybader_amd/synth.py (IEEE basic ops, fixed order).
__global__ __launch_bounds__(TPB) void k_synth_density(Grid g, const double *__restrict__ lat,
                                                       const double *__restrict__ atoms, int n_atoms,
                                                       double background, double *__restrict__ rho) {
    const long long N = (long long)g.nx * g.nyz;
    const long long v = (long long)blockIdx.x * TPB + threadIdx.x;
    if (v >= N) return;
    const int i = (int)(v / g.nyz);
    const int r = (int)(v - (long long)i * g.nyz);
    const int j = r / g.nz, k = r - j * g.nz;
    const double f0 = (double)i / (double)g.nx, f1 = (double)j / (double)g.ny, f2 = (double)k / (double)g.nz;
    double acc = background;
    for (int a = 0; a < n_atoms; a++) {
        const double *A = atoms + 5 * a;
        double d0 = f0 - A[0]; d0 = d0 - rint(d0);
        double d1 = f1 - A[1]; d1 = d1 - rint(d1);
        double d2 = f2 - A[2]; d2 = d2 - rint(d2);
        double r2 = 0.;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            const double xm = (d0 * lat[m] + d1 * lat[3 + m]) + d2 * lat[6 + m];
            const double sq = xm * xm;
            r2 = (m == 0) ? sq : (r2 + sq);
        }
        double t = 1.0 - r2 / ((2048.0 * A[3]) * A[3]);
        if (!(t > 0.0)) t = 0.0;
#pragma unroll
        for (int s = 0; s < 10; s++) t = t * t;
        acc = acc + A[4] * t;
    }
    rho[v] = acc;
}

// utils.vacuum_assign (utils.py:382-401): labels = -1 where rho <= tol, 0 elsewhere, over the
// whole grid; charge/volume partial sums over the owned slab only (block reduce + one atomic).
__global__ __launch_bounds__(TPB) void k_vacuum_assign(Grid g, const double *__restrict__ rho,
                                                       int *__restrict__ labels, double tol, double *sum_rho,
                                                       unsigned long long *count) {
    const long long N = (long long)g.nx * g.nyz;
    const long long v = (long long)blockIdx.x * TPB + threadIdx.x;
    double s = 0.;
    unsigned int n = 0;
    if (v < N) {
        const double r = rho[v];
        const bool vac = r <= tol;  // NaN tol (vacuum_tol=None, interface.py:459) => never
        labels[v] = vac ? -1 : 0;
        const int x = (int)(v / g.nyz);
        if (vac && x >= g.x0 && x < g.x1) { s = r; n = 1; }
    }
    __shared__ double sh[TPB / XB_WAVE];
    __shared__ unsigned int shn[TPB / XB_WAVE];
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o); n += __shfl_down(n, o); }
    const int w = threadIdx.x / XB_WAVE, l = threadIdx.x % XB_WAVE;
    if (l == 0) { sh[w] = s; shn[w] = n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.;
        unsigned int m = 0;
        for (int q = 0; q < TPB / XB_WAVE; q++) { t += sh[q]; m += shn[q]; }
        if (m) { atomicAdd(sum_rho, t); atomicAdd(count, (unsigned long long)m); }
    }
}

// Record a trajectory's maximum `m` for the numbering: first[m] = min owned voxel index reaching m;
// the thread that lowers first[m] from INT_MAX appends m to the maxima list (exactly one does).
__device__ __forceinline__ void note_maximum(int m, int v, int *first, int *max_list, int *max_count, int max_cap) {
    if (__builtin_nontemporal_load(&first[m]) <= v) return;  // already at or below v: nothing to do
    const int old = atomicMin(&first[m], v);
    if (old == XB_INT_MAX) {
        const int k = atomicAdd(max_count, 1);
        if (k < max_cap) max_list[k] = m;
    }
}

// Wave-aggregated note_maximum: per distinct maximum in the wave, one lane reports the smallest
// voxel index of the lanes that reached it.
__device__ __forceinline__ void note_maximum_wave(bool has, int m, int v, int *first, int *max_list,
                                                  int *max_count, int max_cap) {
    unsigned long long todo = __ballot(has);
    const int lane = threadIdx.x % XB_WAVE;
    while (todo) {
        const int leader = __ffsll((unsigned long long)todo) - 1;
        const int lm = __shfl(m, leader);
        const bool mine = has && m == lm;
        int vmin = mine ? v : XB_INT_MAX;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmin = min(vmin, __shfl_xor(vmin, o));
        if (lane == leader) note_maximum(lm, vmin, first, max_list, max_count, max_cap);
        todo &= ~__ballot(mine);
    }
}

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
__global__ __launch_bounds__(TPB) void k_grad_field(Grid g, const double *__restrict__ rho,
                                                    GradRec *__restrict__ G, int *seeds, int *seed_count,
                                                    int seed_cap, int small, int *__restrict__ bmask) {
    __shared__ double tile[GT_X + 2][GT_Y + 2][GT_Z + 2];
    __shared__ int s_mask[GT_Z / 8];
    // plane tiles are counted from the start of the table window (brick aligned; the whole grid on one GPU)
    int x0 = g.wx0 + blockIdx.z * GT_X;
    if (x0 >= g.nx) x0 -= g.nx;
    const int y0 = blockIdx.y * GT_Y, z0 = blockIdx.x * GT_Z;
    if (threadIdx.x < GT_Z / 8) s_mask[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < (GT_X + 2) * (GT_Y + 2) * (GT_Z + 2); i += TPB) {
        const int ez = i % (GT_Z + 2);
        const int r = i / (GT_Z + 2);
        const int ey = r % (GT_Y + 2), ex = r / (GT_Y + 2);
        int X = x0 + ex - 1, Y = y0 + ey - 1, Z = z0 + ez - 1;
        if (small) {
            X = ((X % g.nx) + g.nx) % g.nx; Y = ((Y % g.ny) + g.ny) % g.ny; Z = ((Z % g.nz) + g.nz) % g.nz;
        } else {
            X = wrap_u(X, g.nx); Y = wrap_u(Y, g.ny); Z = wrap_u(Z, g.nz);
        }
        tile[ex][ey][ez] = rho[(X * g.ny + Y) * g.nz + Z];
    }
    __syncthreads();
    const int tz = threadIdx.x & (GT_Z - 1), ty = threadIdx.x / GT_Z;   // 32 x 8 threads, 8 voxels (x) each
    int mine = 0;  // move mask of this thread's voxels (all in brick tz >> 3 of the tile)
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
                    w = (w - c) * g.dist[((ix + 2) % 3) * 9 + ((iy + 2) % 3) * 3 + ((iz + 2) % 3)];
                    w += c;
                    if (w > max_val) { max_val = w; og = ix * 9 + iy * 3 + iz; }
                }
        GradRec o;
        double d0, d1, d2;
        int code;
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
        G[v] = o;
        if (og == XB_OG_SELF) {
            const int q = atomicAdd(seed_count, 1);
            if (q < seed_cap) seeds[q] = v;
            mine |= 1 << 27;
        }
        if (bmask) {  // which neighbour bricks can a move from this voxel reach (moves <= 2 voxels)
            int lo[3], hi[3];
            move_ranges_raw(code, og, o.r0, o.r1, o.r2, lo, hi);
            const int ob[3] = {tx, ty, tz & 7};
            int k0[3], k1[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                k0[j] = (ob[j] + lo[j] < 0) ? -1 : 0;
                k1[j] = (ob[j] + hi[j] >= 8) ? 1 : 0;
            }
            for (int c0 = k0[0]; c0 <= k1[0]; c0++)
                for (int c1 = k0[1]; c1 <= k1[1]; c1++)
                    for (int c2 = k0[2]; c2 <= k1[2]; c2++) mine |= 1 << ((c0 + 1) * 9 + (c1 + 1) * 3 + (c2 + 1));
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
    const GradRec rec = fetch_rec(G, (x * g.ny + y) * g.nz + z);
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
    long long *kp = reinterpret_cast<long long *>(&G[(x * g.ny + y) * g.nz + z].key);
    *kp = (*kp & ~(0x3FFLL << 11)) | ((long long)id << 11);
}

// ---------------------------------------------------------------------------------------------
// Growing the trapping regions brick by brick (8x8x8 voxels).  Let U be a union of sets certain
// for maximum m (closed boxes, earlier bricks).  A brick B without a 26-neighbour maximum whose
// every possible move (any dr) from every voxel lands in B itself or in bricks that are certain
// for the SAME m keeps U + B closed, and a trajectory cannot stay in B forever (it only ends on
// a maximum), so it must enter U: B is certain for m as well.  One round tests the uncertain
// bricks that touch the certain region of the previous round (deterministic: reads `blab` of the
// previous round only) and stamps the ones that pass.
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
__device__ __forceinline__ int brick_nb(int b0, int b1, int b2, int k, int nb0, int nb1, int nb2) {
    return (wrap_any(b0 + k / 9 - 1, nb0) * nb1 + wrap_any(b1 + (k / 3) % 3 - 1, nb1)) * nb2 + wrap_any(b2 + k % 3 - 1, nb2);
}
// provisional labels: an unlabelled brick adopts the label of a labelled brick it can move into
// (smallest label on ties); `plab` double-buffered by the caller.  Any guess is sound -- the kill
// iterations below decide -- a good guess only makes the certain regions larger.
__global__ void k_brick_propagate(int nb0, int nb1, int nb2, const int *__restrict__ bmask,
                                  const int *__restrict__ pin, int *__restrict__ pout, int *changed) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb0 * nb1 * nb2) return;
    int l = pin[b];
    if (l == 0 && !(bmask[b] >> 27)) {
        const int b0 = b / (nb1 * nb2), b1 = (b / nb2) % nb1, b2 = b % nb2;
        const int m = bmask[b];
        int best = 0;
        for (int k = 0; k < 27; k++)
            if ((m >> k) & 1) {
                const int q = pin[brick_nb(b0, b1, b2, k, nb0, nb1, nb2)];
                if (q > 0 && (best == 0 || q < best)) best = q;
            }
        if (best) { l = best; *changed = 1; }
    }
    pout[b] = l;
}
// kill iterations (greatest fixpoint): a non-seed brick stays alive for its label m only while it
// holds no maximum and every brick it can move into is alive with the same label.  What survives,
// together with the seed cubes, is closed under every possible move: a trapping region of m.
__global__ void k_brick_kill(int nb0, int nb1, int nb2, const int *__restrict__ bmask, const int *__restrict__ seed,
                             const int *__restrict__ ain, int *__restrict__ aout, int *changed) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb0 * nb1 * nb2) return;
    int l = ain[b];
    if (l > 0 && seed[b] == 0) {
        const int b0 = b / (nb1 * nb2), b1 = (b / nb2) % nb1, b2 = b % nb2;
        const int m = bmask[b];
        bool ok = !(m >> 27);
        for (int k = 0; k < 27 && ok; k++)
            if ((m >> k) & 1) ok = (ain[brick_nb(b0, b1, b2, k, nb0, nb1, nb2)] == l);
        if (!ok) { l = 0; *changed = 1; }
    }
    aout[b] = l;
}
__global__ void k_count_positive(const int *__restrict__ a, int n, int *count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long b = __ballot(i < n && a[i] > 0);
    if (threadIdx.x % XB_WAVE == 0 && b) atomicAdd(count, __popcll(b));
}

// ---------------------------------------------------------------------------------------------
// neargrid assignment: every owned non-vacuum voxel follows its own dr=0 trajectory
// (refinement.py:17-322 stepping rules without the early stop) to the maximum it reaches.
// One lane per voxel, lanes along z (coalesced first loads).  labels: in 0/-1, out = linear index
// of the maximum (-1 vacuum, -2 = handed to the exact slow kernel).
// ---------------------------------------------------------------------------------------------
// Voxels inside a trapping region end at its maximum: fill their labels in one streaming sweep
// (vacuum voxels keep -1; a region whose maximum is vacuum hands out -1, refinement.py:286) and note
// the maxima for the numbering.  The uncertain bricks go to the work list of k_ng_trace.
__global__ __launch_bounds__(TPB) void k_fill_certain(GridL g, const int *__restrict__ blab, int nb1, int nb2,
                                                      const int *__restrict__ box_max, int *labels, int *first,
                                                      int *max_list, int *max_count, int max_cap) {
    const int vbeg = g.x0 * g.nyz, vend = g.x1 * g.nyz;
    const int v = vbeg + blockIdx.x * TPB + threadIdx.x;
    const bool in = v < vend;
    int result = -1;
    bool has = false;
    if (in) {
        const int x = v / g.nyz;
        const int r = v - x * g.nyz;
        const int y = r / g.nz, z = r - y * g.nz;
        const int b = blab[((x >> 3) * nb1 + (y >> 3)) * nb2 + (z >> 3)];
        if (b > 0 && labels[v] != -1) {
            result = box_max[b - 1];
            if (result != v && labels[result] == -1) result = -1;
            labels[v] = result;
            has = result >= 0;
        }
    }
    note_maximum_wave(has, result, in ? v : 0, first, max_list, max_count, max_cap);
}
// Without vacuum every voxel of a certain brick belongs to its maximum and the smallest voxel
// index of a brick is its corner: the numbering needs one note per owned certain brick, and the
// labels themselves are written by k_relabel_regions after the trace.
__global__ void k_note_certain_bricks(GridL g, int nb0, int nb1, int nb2, int b_lo, int b_hi,
                                      const int *__restrict__ blab, const int *__restrict__ box_max, int *first,
                                      int *max_list, int *max_count, int max_cap) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int l = (b < nb0 * nb1 * nb2 && b >= b_lo && b < b_hi) ? blab[b] : 0;
    const bool has = l > 0;
    const int b0 = b / (nb1 * nb2), b1 = (b / nb2) % nb1, b2 = b % nb2;
    // one atomic per distinct maximum per wave (a handful of maxima own all the bricks)
    note_maximum_wave(has, has ? box_max[l - 1] : 0, ((b0 * 8) * g.ny + b1 * 8) * g.nz + b2 * 8, first, max_list,
                      max_count, max_cap);
}
// labels := rank of the maximum; voxels of certain bricks take it from the brick label, the others
// from the maximum index the trace left in `labels`
__global__ __launch_bounds__(TPB) void k_relabel_regions(GridL g, int *labels, const int *__restrict__ rank,
                                                         const int *__restrict__ blab, int nb1, int nb2,
                                                         const int *__restrict__ box_max) {
    const int v = g.x0 * g.nyz + blockIdx.x * TPB + threadIdx.x;
    if (v >= g.x1 * g.nyz) return;
    const int x = v / g.nyz;
    const int r = v - x * g.nyz;
    const int y = r / g.nz, z = r - y * g.nz;
    const int b = blab[((x >> 3) * nb1 + (y >> 3)) * nb2 + (z >> 3)];
    if (b > 0) labels[v] = rank[box_max[b - 1]];
    else {
        const int m = labels[v];
        if (m >= 0) labels[v] = rank[m];
    }
}
__global__ void k_brick_walk_list(int nbr, int b_lo, int b_hi, const int *__restrict__ blab, int *walk, int *n_walk) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;   // bricks [b_lo, b_hi) are the owned slab
    const bool hit = b < nbr && b >= b_lo && b < b_hi && blab[b] <= 0;
    const unsigned long long m = __ballot(hit);
    if (!m) return;
    const int lane = threadIdx.x % XB_WAVE;
    int base = 0;
    if (lane == 0) base = atomicAdd(n_walk, __popcll(m));
    base = __shfl(base, 0);
    if (hit) walk[base + __popcll(m & ((1ull << lane) - 1ull))] = b;
}

__device__ __forceinline__ void og_offsets(int og, int &ox, int &oy, int &oz) {
    ox = og / 9 - 1; oy = (og / 3) % 3 - 1; oz = og % 3 - 1;
}

template <int K>
__global__ __launch_bounds__(TPB) void k_ng_trace(GridL g, const GradRec *__restrict__ G,
                                                  const int *__restrict__ box_max, const int *__restrict__ blab,
                                                  int nb1, int nb2, const int *__restrict__ walk, int n_walk,
                                                  int *labels, int *first,
                                                  int *max_list, int *max_count, int max_cap, int *ovf_list,
                                                  int *ovf_count, int ovf_cap, int maxsteps, int opt) {
    // XCD-aware block order (opt bit 1): blocks are dealt round-robin over the 8 XCDs, each with
    // its own L2; give XCD k the k-th contiguous eighth of the work so that spatial neighbours --
    // whose trajectories read the same table lines -- share one L2.
    int blk = blockIdx.x;
    if (opt & 2) {
        const int per = gridDim.x >> 3;
        if (blk < (per << 3)) blk = (blk & 7) * per + (blk >> 3);
    }
    const int wpb = blockDim.x / XB_WAVE;  // waves per block (launch-time choice)
    const int wave = blk * wpb + threadIdx.x / XB_WAVE;
    const int lane = threadIdx.x % XB_WAVE;
    int sx, sy, sz;
    if (walk) {     // work list of the 8^3 bricks outside the trapping regions: 8 waves (4x4x4 each) per brick
        if ((wave >> 3) >= n_walk) return;
        const int b = walk[wave >> 3], sub = wave & 7;
        const int b0 = b / (nb1 * nb2), b1 = (b / nb2) % nb1, b2 = b % nb2;
        sx = b0 * 8 + ((sub >> 2) << 2) + (lane >> 4);
        sy = b1 * 8 + (((sub >> 1) & 1) << 2) + ((lane >> 2) & 3);
        sz = b2 * 8 + ((sub & 1) << 2) + (lane & 3);
        if (sx < g.x0 || sx >= g.x1) return;  // whole wave: 4 planes of one brick half, slab edges are brick aligned or not owned
    } else if (opt & 1) {  // one wave = one 4x4x4 brick of start voxels (z fastest: 4 lanes per 128-B table line)
        const int bz_n = (g.nz + 3) >> 2, by_n = (g.ny + 3) >> 2;
        const int bx = wave / (by_n * bz_n);
        const int brem = wave - bx * (by_n * bz_n);
        const int by = brem / bz_n, bz = brem - by * bz_n;
        sx = g.x0 + bx * 4 + (lane >> 4); sy = by * 4 + ((lane >> 2) & 3); sz = bz * 4 + (lane & 3);
    } else {        // one wave = a run of 64 voxels along z
        const int rz_n = (g.nz + 63) >> 6;
        const int row = wave / rz_n;
        sz = (wave - row * rz_n) * 64 + lane;
        sx = g.x0 + row / g.ny;
        sy = row - (row / g.ny) * g.ny;
    }
    const bool valid = sx < g.x1 && sy < g.ny && sz < g.nz;
    const int v = valid ? (sx * g.ny + sy) * g.nz + sz : 0;
    bool moving = false;
    int result = -1;
    int px = 0, py = 0, pz = 0, lp = 0, steps = 0;
    double dr0 = 0., dr1 = 0., dr2 = 0.;
    GradRec rec = {0., 0., 0., 0.};
    PathWindow<K> w;
    w.init(0, 0.);
    if (valid && labels[v] != -1) {
        px = sx; py = sy; pz = sz;
        lp = v;
        // trapping regions: brick labels (grids made of whole 8^3 bricks) or box ids in the keys
        int b = blab ? blab[((sx >> 3) * nb1 + (sy >> 3)) * nb2 + (sz >> 3)] : 0;
        if (b <= 0) {
            rec = fetch_rec(G, v);
            b = key_box(rec.key);
        }
        if (b > 0) result = box_max[b - 1];  // starts inside a trapping region: ends at its maximum
        else { w.init(v, rec.key); moving = true; }
    }
    while (__any(moving)) {
        if (moving) {
            const int bits = key_bits(rec.key);
            const int code = bits & 63;
            int qx, qy, qz, lq = 0;
            // refinement.py:132-154: the gradient move (if the voxel has one)
            bool og_move = (code == XB_STAY_CODE);
            if (!og_move) {
                ng_move_t(g, px, py, pz, rec, code, dr0, dr1, dr2, qx, qy, qz);
                lq = lin3f(g, qx, qy, qz);
                og_move = w.contains(lq);  // refinement.py:200: already been here on this path
            }
            if (og_move) {  // refinement.py:201-235: dr = 0 and one ongrid step from p (tabulated)
                const int og = (bits >> 6) & 31;
                if (og == XB_OG_SELF) { result = lp; moving = false; }  // break_flag: p is the maximum
                else {
                    int ox, oy, oz;
                    og_offsets(og, ox, oy, oz);
                    dr0 = dr1 = dr2 = 0.;
                    qx = wrap_u(px + ox, g.nx); qy = wrap_u(py + oy, g.ny); qz = wrap_u(pz + oz, g.nz);
                    lq = lin3f(g, qx, qy, qz);
                }
            }
            if (moving) {
                const int bl = blab ? blab[((qx >> 3) * nb1 + (qy >> 3)) * nb2 + (qz >> 3)] : 0;
                const bool in_win = plane_in_window(g, qx);  // the table only exists inside the window (slabs)
                const GradRec nr = fetch_rec(G, in_win ? lq : lp);
                const int b = bl > 0 ? bl : (in_win ? key_box(nr.key) : 0);
                if (b) {  // arrived inside a trapping region (q cannot be an old path voxel: the
                    result = box_max[b - 1];  // trajectory would have stopped there already)
                    moving = false;
                } else if (!in_win || (!og_move && nr.key <= w.m_old) || ++steps > maxsteps) {
                    result = -2;  // left the table window / membership undecidable: exact slow kernel
                    moving = false;  // (ongrid moves are appended without a membership test, 305-315)
                } else {
                    w.push(lq, nr.key);
                    px = qx; py = qy; pz = qz; lp = lq; rec = nr;
                }
            }
        }
    }
    // a maximum that is itself vacuum hands its -1 to the start voxel (refinement.py:286)
    if (valid && result >= 0 && result != v && labels[result] == -1) result = -1;
    if (valid) labels[v] = result;
    note_maximum_wave(valid && result >= 0, result, v, first, max_list, max_count, max_cap);
    if (valid && result == -2) {
        const int k = atomicAdd(ovf_count, 1);
        if (k < ovf_cap) ovf_list[k] = v;
    }
}

// Exact slow path for the (rare) trajectories whose path membership could not be decided from the
// window: the whole path lives in global scratch and is scanned linearly.
// mode 0: assignment (write maximum index, note it); mode 1: refinement retrace.
__global__ void k_trace_slow(Grid g, const double *__restrict__ rho, int *labels, const int8_t *known_ro,
                             int8_t *known, const int *list, int n, int *path, int lmax, int refine, int *first,
                             int *max_list, int *max_count, int max_cap, int *changed, int *escaped, int *err) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int v = list[t];
    int *P = path + (size_t)t * lmax;
    int np = 0;
    int px = v / g.nyz;
    int r = v - px * g.nyz;
    int py = r / g.nz, pz = r - py * g.nz, lp = v;
    double c = rho[v], dr0 = 0., dr1 = 0., dr2 = 0.;
    const int vol_num = labels[v];
    P[np++] = v;
    int result = -3;
    for (;;) {
        int qx, qy, qz;
        const bool stay = ng_step(rho, g, px, py, pz, lp, c, dr0, dr1, dr2, qx, qy, qz);
        int lq = lin3(g, qx, qy, qz);
        bool on_path = stay;
        for (int k = np - 1; k >= 0 && !on_path; k--) on_path = (P[k] == lq);
        if (on_path) {
            dr0 = dr1 = dr2 = 0.;
            og_step(rho, g, g.dist, px, py, pz, c, qx, qy, qz);
            lq = lin3(g, qx, qy, qz);
            if (qx == px && qy == py && qz == pz) { result = lp; break; }
        }
        if (refine) {
            if (!plane_valid(g, qx)) { known[v] = -6; atomicAdd(escaped, 1); return; }
            if (known_ro[lq] == 2) { result = lq; break; }
        }
        if (np >= lmax) { atomicExch(err, 1); return; }
        P[np++] = lq;
        px = qx; py = qy; pz = qz; lp = lq; c = rho[lq];
    }
    if (refine) {
        const int nv = labels[result];
        if (nv != vol_num) { labels[v] = nv; known[v] = -2; atomicAdd(changed, 1); }
        else known[v] = -1;
    } else {
        if (result != v && labels[result] == -1) result = -1;
        labels[v] = result;
        if (result >= 0) note_maximum(result, v, first, max_list, max_count, max_cap);
    }
}

// ---------------------------------------------------------------------------------------------
// ongrid assignment (methods.py:15-219).  The ascent is memoryless, so the sequential path
// compression of the reference equals: best-neighbour pointer per voxel, then pointer jumping.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_og_pointer(Grid g, const double *__restrict__ rho, int *labels) {
    const long long N = (long long)g.nx * g.nyz;
    const long long vv = (long long)blockIdx.x * TPB + threadIdx.x;
    if (vv >= N) return;
    const int v = (int)vv;
    if (labels[v] == -1) return;  // vacuum stays -1 (methods.py:73-74)
    const int px = v / g.nyz;
    const int r = v - px * g.nyz;
    const int py = r / g.nz, pz = r - py * g.nz;
    int qx, qy, qz;
    og_step(rho, g, g.dist, px, py, pz, rho[v], qx, qy, qz);
    labels[v] = lin3(g, qx, qy, qz);
}
// A chain that steps onto a vacuum voxel inherits -1 (methods.py:166-168).  In-place and
// asynchronous: any value read is an ancestor of the root, so progress is monotone.
__global__ __launch_bounds__(TPB) void k_og_jump(Grid g, int *labels, int *not_done) {
    const long long N = (long long)g.nx * g.nyz;
    const long long vv = (long long)blockIdx.x * TPB + threadIdx.x;
    if (vv >= N) return;
    const int v = (int)vv;
    int p = labels[v];
    if (p < 0 || p == v) return;
    int q = labels[p];
    if (q == p) return;  // parent is a root
    if (q >= 0) {
        const int q2 = labels[q];  // two hops per sweep
        if (q2 >= 0) q = q2;
        else q = -1;
    }
    labels[v] = q;
    if (q >= 0) *not_done = 1;
}
__global__ __launch_bounds__(TPB) void k_note_roots(Grid g, const int *labels, int *first, int *max_list,
                                                    int *max_count, int max_cap) {
    const long long vbeg = (long long)g.x0 * g.nyz, vend = (long long)g.x1 * g.nyz;
    const long long vv = vbeg + (long long)blockIdx.x * TPB + threadIdx.x;
    const bool valid = vv < vend;
    const int v = valid ? (int)vv : 0;
    const int m = valid ? labels[v] : -1;
    note_maximum_wave(valid && m >= 0, m, v, first, max_list, max_count, max_cap);
}

// numbering helpers ---------------------------------------------------------------------------
__global__ void k_gather_first(const int *first, const int *max_list, int n, int *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = first[max_list[i]];
}
__global__ void k_set_rank(int *first, const int *max_sorted, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) first[max_sorted[i]] = i;
}
__global__ void k_reset_first(int *first, const int *max_list, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) first[max_list[i]] = XB_INT_MAX;
}
// labels[v] (maximum index) -> rank stored in first[maximum]
__global__ __launch_bounds__(TPB) void k_relabel(Grid g, int *labels, const int *__restrict__ rank) {
    const long long vbeg = (long long)g.x0 * g.nyz, vend = (long long)g.x1 * g.nyz;
    const long long v = vbeg + (long long)blockIdx.x * TPB + threadIdx.x;
    if (v >= vend) return;
    const int m = labels[v];
    if (m >= 0) labels[v] = rank[m];
}

// ---------------------------------------------------------------------------------------------
// refinement.edge_find (refinement.py:326-405) on a fresh `known`, as two order-free passes.
// Pass 1 (planes [x0-1, x1+1)): -2 if a non-vacuum neighbour carries another label and the voxel
// is not a 26-neighbour density maximum; else 2 (non-vacuum) / 0 (vacuum).
// Pass 2 (owned planes): voxels >= 0 with an edge in their 27-box become -1 (refinement.py:403-404,
// which has no vacuum test).  Together these equal the sequential in-place sweep.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void classify27(const Grid &g, const double *__restrict__ rho,
                                           const int *__restrict__ labels, int x, int y, int z, int v,
                                           bool &is_edge, bool &is_max) {
    const int vol_num = labels[v];
    is_edge = false;
    is_max = true;
    int nb[27];
    int k = 0;
#pragma unroll
    for (int ix = -1; ix < 2; ix++) {
        const int tx = wrapi(x + ix, g.nx);
#pragma unroll
        for (int iy = -1; iy < 2; iy++) {
            const int ty = wrapi(y + iy, g.ny);
#pragma unroll
            for (int iz = -1; iz < 2; iz++) {
                const int tz = wrapi(z + iz, g.nz);
                const int l = lin3(g, tx, ty, tz);
                const int nv = labels[l];
                nb[k++] = (nv == -1) ? -1 : l;
                if (nv != -1 && nv != vol_num) is_edge = true;
            }
        }
    }
    if (!is_edge) return;  // is_max only matters for edges (refinement.py:376-383)
    const double max_val = rho[v];
#pragma unroll
    for (k = 0; k < 27; k++)
        if (nb[k] >= 0 && rho[nb[k]] > max_val) is_max = false;
}

// Block-wide exclusive scan of a small per-thread count (TPB threads); returns the offset of this
// thread and the block total.
__device__ __forceinline__ int block_scan_excl(int cnt, int &total) {
    __shared__ int wsum[TPB / XB_WAVE];
    const int lane = threadIdx.x % XB_WAVE, w = threadIdx.x / XB_WAVE;
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < XB_WAVE; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == XB_WAVE - 1) wsum[w] = incl;
    __syncthreads();
    int base = 0;
    total = 0;
#pragma unroll
    for (int q = 0; q < TPB / XB_WAVE; q++) {
        if (q < w) base += wsum[q];
        total += wsum[q];
    }
    __syncthreads();
    return base + incl - cnt;
}

// buni[K] = the label shared by all 512 voxels of brick K, or INT_MIN when the brick is mixed.
// Lets the edge sweep skip tiles whose whole 3x3x3 surroundings carry one label (no edge possible).
#define XB_MIXED (-2147483647 - 1)
__global__ __launch_bounds__(TPB) void k_label_uniform(GridL g, const int *__restrict__ labels, int nb1, int nb2,
                                                       int *__restrict__ buni) {
    __shared__ int s_min, s_max;
    if (threadIdx.x == 0) { s_min = 2147483647; s_max = XB_MIXED; }
    __syncthreads();
    const int b = blockIdx.x;
    const int b0 = b / (nb1 * nb2), b1 = (b / nb2) % nb1, b2 = b % nb2;
    int lo = 2147483647, hi = XB_MIXED;
    for (int t = threadIdx.x; t < 512; t += TPB) {
        const int l = labels[((b0 * 8 + t / 64) * g.ny + b1 * 8 + (t / 8) % 8) * g.nz + b2 * 8 + t % 8];
        lo = min(lo, l); hi = max(hi, l);
    }
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
    if (threadIdx.x % XB_WAVE == 0) { atomicMin(&s_min, lo); atomicMax(&s_max, hi); }
    __syncthreads();
    if (threadIdx.x == 0) buni[b] = (s_min == s_max) ? s_min : XB_MIXED;
}
// After an assignment without vacuum every certain brick is uniform by construction (all its voxels
// carry the rank of the region's maximum): only the bricks of the walk list need the label scan.
__global__ void k_buni_from_regions(int nbr, const int *__restrict__ blab, const int *__restrict__ box_max,
                                    const int *__restrict__ rank, int *__restrict__ buni) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbr) return;
    const int l = blab[b];
    if (l > 0) buni[b] = rank[box_max[l - 1]];
}
__global__ __launch_bounds__(TPB) void k_label_uniform_list(GridL g, const int *__restrict__ labels, int nb1, int nb2,
                                                            const int *__restrict__ walk, int n_walk,
                                                            int *__restrict__ buni) {
    __shared__ int s_min, s_max;
    if ((int)blockIdx.x >= n_walk) return;
    if (threadIdx.x == 0) { s_min = 2147483647; s_max = XB_MIXED; }
    __syncthreads();
    const int b = walk[blockIdx.x];
    const int b0 = b / (nb1 * nb2), b1 = (b / nb2) % nb1, b2 = b % nb2;
    int lo = 2147483647, hi = XB_MIXED;
    for (int t = threadIdx.x; t < 512; t += TPB) {
        const int l = labels[((b0 * 8 + t / 64) * g.ny + b1 * 8 + (t / 8) % 8) * g.nz + b2 * 8 + t % 8];
        lo = min(lo, l); hi = max(hi, l);
    }
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
    if (threadIdx.x % XB_WAVE == 0) { atomicMin(&s_min, lo); atomicMax(&s_max, hi); }
    __syncthreads();
    if (threadIdx.x == 0) buni[b] = (s_min == s_max) ? s_min : XB_MIXED;
}

// refinement.py:385-404 as written there: every listed edge voxel turns the known >= 0 voxels of
// its 27-box into -1 (all -2 flags are final before this kernel starts).
__global__ __launch_bounds__(TPB) void k_edge_dilate_list(GridL g, int8_t *known, const int *__restrict__ list, int n) {
    const int t = blockIdx.x * TPB + threadIdx.x;
    if (t >= n) return;
    const int v = list[t];
    const int x = v / g.nyz;
    const int r = v - x * g.nyz;
    const int y = r / g.nz, z = r - y * g.nz;
#pragma unroll
    for (int ix = -1; ix < 2; ix++) {
        const int tx = wrapi(x + ix, g.nx);
#pragma unroll
        for (int iy = -1; iy < 2; iy++) {
            const int ty = wrapi(y + iy, g.ny);
#pragma unroll
            for (int iz = -1; iz < 2; iz++) {
                const int l = (tx * g.ny + ty) * g.nz + wrapi(z + iz, g.nz);
                if (known[l] >= 0) known[l] = -1;
            }
        }
    }
}

// LDS-tiled edge_find pass 1: a block stages the labels of a 4x8x64 tile plus a one-voxel periodic
// halo (6x10x66 ints) in LDS, every thread classifies 8 voxels from the staged 3x3x3
// neighbourhoods, and the block appends its owned edge voxels to the edge list with ONE atomic
// (the list length is the edge count edge_find returns).  rho is only read for the few voxels
// that have a foreign neighbour (the is_max test, refinement.py:374-375).
#define ET_X 4
#define ET_Y 8
#define ET_Z 64
__global__ __launch_bounds__(TPB) void k_edge_flag_tiled(GridL g, const double *__restrict__ rho,
                                                         const int *__restrict__ labels,
                                                         int8_t *__restrict__ known, int xa, int nplanes,
                                                         int *__restrict__ list, int *list_count, int small,
                                                         const int *__restrict__ buni,
                                                         const GradRec *__restrict__ G) {
    __shared__ int tile[ET_X + 2][ET_Y + 2][ET_Z + 2];
    const int tx0 = blockIdx.z * ET_X, y0 = blockIdx.y * ET_Y, z0 = blockIdx.x * ET_Z;
    if (buni) {
        // every brick that meets the tile or its one-voxel halo carries the same single label: no
        // voxel of the tile has a foreign neighbour (2 x 3 x 10 bricks, one lookup per thread)
        __shared__ int s_lab, s_mixed;
        if (threadIdx.x == 0) { s_lab = XB_MIXED; s_mixed = 0; }
        __syncthreads();
        const int nb0 = g.nx >> 3, nb1 = g.ny >> 3, nb2 = g.nz >> 3;
        const int bx_lo = (xa + tx0 - 1) >> 3, bx_n = ((xa + tx0 + ET_X) >> 3) - bx_lo + 1;  // arithmetic shift: -1 >> 3 == -1
        const int by_lo = (y0 - 1) >> 3, by_n = ((y0 + ET_Y) >> 3) - by_lo + 1;
        const int bz_lo = (z0 - 1) >> 3, bz_n = ((z0 + ET_Z) >> 3) - bz_lo + 1;
        for (int t = threadIdx.x; t < bx_n * by_n * bz_n; t += TPB) {
            const int q0 = wrap_any(bx_lo + t / (by_n * bz_n), nb0), q1 = wrap_any(by_lo + (t / bz_n) % by_n, nb1),
                      q2 = wrap_any(bz_lo + t % bz_n, nb2);
            const int l = buni[(q0 * nb1 + q1) * nb2 + q2];
            if (l == XB_MIXED) s_mixed = 1;
            else {
                const int old = atomicCAS(&s_lab, XB_MIXED, l);
                if (old != XB_MIXED && old != l) s_mixed = 1;
            }
        }
        __syncthreads();
        if (!s_mixed) {
            const int8_t o = (s_lab == -1) ? 0 : 2;  // vacuum stays 0 (refinement.py:342-343), else "known"
            const int tz = threadIdx.x & 63, tyb = threadIdx.x >> 6;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int xr = tx0 + (k >> 1), y = y0 + tyb + ((k & 1) << 2), z = z0 + tz;
                if (xr < nplanes && y < g.ny && z < g.nz) {
                    int x = xa + xr;
                    if (x >= g.nx) x -= g.nx;
                    known[(x * g.ny + y) * g.nz + z] = o;
                }
            }
            return;
        }
    }
    for (int i = threadIdx.x; i < (ET_X + 2) * (ET_Y + 2) * (ET_Z + 2); i += TPB) {
        const int ez = i % (ET_Z + 2);
        const int r = i / (ET_Z + 2);
        const int ey = r % (ET_Y + 2), ex = r / (ET_Y + 2);
        int X = xa + tx0 + ex - 1, Y = y0 + ey - 1, Z = z0 + ez - 1;
        if (small) {
            X = ((X % g.nx) + g.nx) % g.nx; Y = ((Y % g.ny) + g.ny) % g.ny; Z = ((Z % g.nz) + g.nz) % g.nz;
        } else {
            X = wrap_u(X, g.nx); X = wrap_u(X, g.nx); Y = wrap_u(Y, g.ny); Z = wrap_u(Z, g.nz);
        }
        tile[ex][ey][ez] = labels[(X * g.ny + Y) * g.nz + Z];
    }
    __syncthreads();
    const int tz = threadIdx.x & 63, tyb = threadIdx.x >> 6;
    int8_t out[8];
    int vidx[8];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int tx = k >> 1, ty = tyb + ((k & 1) << 2);
        const int xr = tx0 + tx, y = y0 + ty, z = z0 + tz;
        out[k] = 1;  // 1 = outside the grid / the plane range: nothing to store
        vidx[k] = -1;
        if (xr < nplanes && y < g.ny && z < g.nz) {
            int x = xa + xr;
            if (x >= g.nx) x -= g.nx;
            const int v = (x * g.ny + y) * g.nz + z;
            const int lab = tile[tx + 1][ty + 1][tz + 1];
            int8_t o = 0;  // vacuum voxels are not classified (refinement.py:342-343)
            if (lab != -1) {
                bool is_edge = false;
#pragma unroll
                for (int dx = 0; dx < 3; dx++)
#pragma unroll
                    for (int dy = 0; dy < 3; dy++)
#pragma unroll
                        for (int dz = 0; dz < 3; dz++) {
                            const int nv = tile[tx + dx][ty + dy][tz + dz];
                            is_edge |= (nv != -1) & (nv != lab);
                        }
                o = 2;
                if (is_edge) {  // refinement.py:374-383: an edge unless it is a 26-neighbour maximum
                    bool is_max = true, decided = false;
                    if (G) {
                        // the table knows the best distance-weighted neighbour of v; if there is one
                        // (and it is not vacuum) that neighbour is denser than v: not a maximum.
                        // (weighted > rho(v) implies rho(n) > rho(v); the converse can fail by
                        // rounding, so "no such neighbour" still takes the full test)
                        const int og = key_og(G[v].key);
                        if (og != XB_OG_SELF && tile[tx + og / 9][ty + (og / 3) % 3][tz + og % 3] != -1) {
                            is_max = false;
                            decided = true;
                        }
                    }
                    if (!decided) {
                        const double c = rho[v];
                        for (int dx = -1; dx < 2; dx++) {
                            const int X = wrapi(x + dx, g.nx);
                            for (int dy = -1; dy < 2; dy++) {
                                const int Y = wrapi(y + dy, g.ny);
                                for (int dz = -1; dz < 2; dz++) {
                                    const int Z = wrapi(z + dz, g.nz);
                                    if (tile[tx + 1 + dx][ty + 1 + dy][tz + 1 + dz] != -1 &&
                                        rho[(X * g.ny + Y) * g.nz + Z] > c)
                                        is_max = false;
                                }
                            }
                        }
                    }
                    if (!is_max) {
                        o = -2;
                        if (x >= g.x0 && x < g.x1) { vidx[k] = v; cnt++; }
                    }
                }
            }
            out[k] = o;
            known[v] = o;
        }
    }
    int total;
    const int off = block_scan_excl(cnt, total);
    __shared__ int base_s;
    if (threadIdx.x == 0) base_s = total ? atomicAdd(list_count, total) : 0;
    __syncthreads();
    int w = base_s + off;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (vidx[k] >= 0) list[w++] = vidx[k];
}

// compaction of owned voxels with known == value, 16 voxels per thread, one atomic per block
__global__ __launch_bounds__(TPB) void k_compact_known16(GridL g, const int8_t *__restrict__ known, int value,
                                                         int *__restrict__ list, int *count) {
    const long long vbeg = (long long)g.x0 * g.nyz, vend = (long long)g.x1 * g.nyz;
    const long long base = vbeg + ((long long)blockIdx.x * TPB + threadIdx.x) * 16;
    int8_t b[16];
    if (base + 16 <= vend && ((vbeg & 15) == 0)) {
        *reinterpret_cast<uint4 *>(b) = *reinterpret_cast<const uint4 *>(known + base);
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) b[k] = (base + k < vend) ? known[base + k] : (int8_t)(value + 1);
    }
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) cnt += (b[k] == value);
    int total;
    const int off = block_scan_excl(cnt, total);
    __shared__ int base_s;
    if (threadIdx.x == 0) base_s = total ? atomicAdd(count, total) : 0;
    __syncthreads();
    int w = base_s + off;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (b[k] == value) list[w++] = (int)(base + k);
}

// known >= 0 with a `flag` voxel in the 27-box -> -1.  Used by edge_find (flag=-2) and edge_check
// (flag=-3).  Reads test == flag only, writes only turn 0/2 into -1: safe in place.
__global__ __launch_bounds__(TPB) void k_edge_dilate(Grid g, int8_t *known, int xa, int nplanes, int flag) {
    const long long vv = (long long)blockIdx.x * TPB + threadIdx.x;
    if (vv >= (long long)nplanes * g.nyz) return;
    const int xr = (int)(vv / g.nyz);
    const int r = (int)(vv - (long long)xr * g.nyz);
    int x = xa + xr;
    if (x >= g.nx) x -= g.nx;
    const int y = r / g.nz, z = r - y * g.nz;
    const int v = lin3(g, x, y, z);
    if (known[v] < 0) return;
    bool near = false;
#pragma unroll
    for (int ix = -1; ix < 2; ix++) {
        const int tx = wrapi(x + ix, g.nx);
#pragma unroll
        for (int iy = -1; iy < 2; iy++) {
            const int ty = wrapi(y + iy, g.ny);
#pragma unroll
            for (int iz = -1; iz < 2; iz++) {
                const int tz = wrapi(z + iz, g.nz);
                near |= (known[lin3(g, tx, ty, tz)] == flag);
            }
        }
    }
    if (near) known[v] = -1;
}

// ---------------------------------------------------------------------------------------------
// refinement.neargrid (refinement.py:17-322): retrace the listed edge voxels (known == -2).
// Traces only read `known` for the == 2 test and `labels` at known==2 voxels / maxima, and only
// write their own start voxel, so they are independent -- exactly as in the reference, where the
// +5 marks are per-trace scratch.  `known` therefore doubles as `rknown`.
// ---------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(TPB) void k_refine_trace(GridL g, const GradRec *__restrict__ G, int *labels,
                                                      int8_t *known, const int *__restrict__ list, int n,
                                                      int *changed, int *escaped, int *ovf_list, int *ovf_count,
                                                      int ovf_cap, int maxsteps) {
    const int t = blockIdx.x * TPB + threadIdx.x;
    const bool valid = t < n;
    const int v = valid ? list[t] : 0;
    bool moving = false;
    int result = -3;  // terminal voxel index; -2 overflow; -4 escaped
    int px = 0, py = 0, pz = 0, lp = 0, steps = 0, vol_num = 0;
    double dr0 = 0., dr1 = 0., dr2 = 0.;
    GradRec rec = {0., 0., 0., 0.};
    PathWindow<K> w;
    w.init(0, 0.);
    if (valid) {
        px = v / g.nyz;
        const int r = v - px * g.nyz;
        py = r / g.nz;
        pz = r - py * g.nz;
        lp = v;
        rec = fetch_rec(G, v);
        vol_num = labels[v];
        w.init(v, rec.key);
        moving = true;
    }
    while (__any(moving)) {
        if (moving) {
            const int bits = key_bits(rec.key);
            const int code = bits & 63;
            int qx, qy, qz, lq = 0;
            bool og_move = (code == XB_STAY_CODE);
            if (!og_move) {
                ng_move_t(g, px, py, pz, rec, code, dr0, dr1, dr2, qx, qy, qz);
                lq = lin3f(g, qx, qy, qz);
                og_move = w.contains(lq);  // refinement.py:200
            }
            if (og_move) {  // refinement.py:201-235
                const int og = (bits >> 6) & 31;
                if (og == XB_OG_SELF) { result = lp; moving = false; }  // a maximum: refinement.py:283-292
                else {
                    int ox, oy, oz;
                    og_offsets(og, ox, oy, oz);
                    dr0 = dr1 = dr2 = 0.;
                    qx = wrap_u(px + ox, g.nx); qy = wrap_u(py + oy, g.ny); qz = wrap_u(pz + oz, g.nz);
                    lq = lin3f(g, qx, qy, qz);
                }
            }
            if (moving) {
                const bool in_win = plane_in_window(g, qx);
                const GradRec nr = fetch_rec(G, in_win ? lq : lp);
                if (!plane_valid(g, qx)) { result = -4; moving = false; }
                else if (!in_win || (!og_move && nr.key <= w.m_old) || ++steps > maxsteps) { result = -2; moving = false; }
                else if (known[lq] == 2) { result = lq; moving = false; }  // refinement.py:294-303
                else {
                    w.push(lq, nr.key);
                    px = qx; py = qy; pz = qz; lp = lq; rec = nr;
                }
            }
        }
    }
    int ch = 0, es = 0;
    if (valid) {
        if (result >= 0) {
            const int nv = labels[result];
            if (nv != vol_num) { labels[v] = nv; known[v] = -2; ch = 1; }  // refinement.py:288-289
            else known[v] = -1;                                             // refinement.py:291 (+5 +1 -5)
        } else if (result == -2) {
            const int k = atomicAdd(ovf_count, 1);
            if (k < ovf_cap) ovf_list[k] = v;
        } else if (result == -4) { known[v] = -6; es = 1; }  // left the valid slab: parked for the fallback
    }
    const unsigned long long bc = __ballot(ch), be = __ballot(es);
    if (threadIdx.x % XB_WAVE == 0) {
        if (bc) atomicAdd(changed, __popcll(bc));
        if (be) atomicAdd(escaped, __popcll(be));
    }
}

// ---------------------------------------------------------------------------------------------
// refinement.edge_check (refinement.py:409-508).  The sequential scan re-classifies the 27-box of
// every voxel that is still -2 when the scan reaches it; an earlier processed neighbour j < i
// rewrites i to -1 / -3 unless i is an (edge & maximum) voxel, in which case i is processed too.
// So the processed set P is the lexicographically-first greedy choice:
//     i in P  <=>  class(i) == edge&max  or  no j in P with j < i, j in box(i).
// P is resolved in rounds (a voxel decides once all earlier changed neighbours have decided);
// the final `known` is then a pure function of P and the static classes.
// temp codes in `known`: -2 undecided, -4 processed, -5 skipped.
// ---------------------------------------------------------------------------------------------
// One round: every still-undecided entry looks at its (at most 13) earlier neighbours.  Work
// lists live on the device (`in` -> survivors appended to `out`), so rounds are queued
// back-to-back without a host round trip; the host only polls the survivor count now and then.
//   edge&max voxel            -> processed at once (earlier boxes leave it -2, refinement.py:480)
//   an earlier neighbour is P -> skipped
//   no earlier neighbour left undecided -> processed
__global__ __launch_bounds__(TPB) void k_ec_decide(Grid g, const double *__restrict__ rho,
                                                   const int *__restrict__ labels, int8_t *known,
                                                   const int *__restrict__ list, int8_t *st,
                                                   const int *__restrict__ in, const int *n_in, int *out,
                                                   int *n_out, int first_round) {
    const int n = *n_in;
    for (int e = blockIdx.x * TPB + threadIdx.x; e < n; e += gridDim.x * TPB) {
        const int t = first_round ? e : in[e];
        const int v = list[t];
        const int x = v / g.nyz;
        const int r = v - x * g.nyz;
        const int y = r / g.nz, z = r - y * g.nz;
        bool blocked = false, has_proc = false;
#pragma unroll
        for (int ix = -1; ix < 2; ix++) {
            const int tx = wrapi(x + ix, g.nx);
#pragma unroll
            for (int iy = -1; iy < 2; iy++) {
                const int ty = wrapi(y + iy, g.ny);
#pragma unroll
                for (int iz = -1; iz < 2; iz++) {
                    const int tz = wrapi(z + iz, g.nz);
                    const int l = lin3(g, tx, ty, tz);
                    if (l < v) {
                        const int8_t k = __builtin_nontemporal_load(&known[l]);
                        blocked |= (k == -2);
                        has_proc |= (k == -4);
                    }
                }
            }
        }
        int decision = 0;  // 0 wait, 1 processed, 2 skipped
        if (!blocked && !has_proc) decision = 1;
        else {
            int8_t cls = st[t] >> 2;  // cached class: 1 = edge&max, 2 = other
            if (cls == 0) {
                bool is_edge, is_max;
                classify27(g, rho, labels, x, y, z, v, is_edge, is_max);
                cls = (is_edge && is_max) ? 1 : 2;
            }
            if (cls == 1) decision = 1;
            else if (has_proc) decision = 2;
            else st[t] = (int8_t)(cls << 2);  // still waiting: remember the class
        }
        if (decision) {
            st[t] = (int8_t)decision;
            known[v] = decision == 1 ? (int8_t)-4 : (int8_t)-5;
        } else {
            out[atomicAdd(n_out, 1)] = t;
        }
    }
}
// apply: every processed voxel re-classifies its 27-box (refinement.py:428-504)
__global__ __launch_bounds__(TPB) void k_ec_apply(Grid g, const double *__restrict__ rho,
                                                  const int *__restrict__ labels, int8_t *known,
                                                  const int *__restrict__ list, int n, const int8_t *st,
                                                  unsigned long long *checked) {
    const int t = blockIdx.x * TPB + threadIdx.x;
    if (t >= n || st[t] != 1) return;
    const int v = list[t];
    const int x = v / g.nyz;
    const int r = v - x * g.nyz;
    const int y = r / g.nz, z = r - y * g.nz;
    unsigned int nchk = 0;
    for (int ex = -1; ex < 2; ex++) {
        const int tx = wrapi(x + ex, g.nx);
        for (int ey = -1; ey < 2; ey++) {
            const int ty = wrapi(y + ey, g.ny);
            for (int ez = -1; ez < 2; ez++) {
                const int tz = wrapi(z + ez, g.nz);
                const int l = lin3(g, tx, ty, tz);
                // NB no vacuum test on the box voxel (bug-compatible)
                bool is_edge, is_max;
                classify27(g, rho, labels, tx, ty, tz, l, is_edge, is_max);
                if (!is_edge) { known[l] = -1; nchk++; }
                else if (!is_max) known[l] = -3;
            }
        }
    }
    if (nchk) atomicAdd(checked, (unsigned long long)nchk);
}
// restore processed edge&max voxels (untouched by their own box) to -2
__global__ void k_ec_restore(int8_t *known, const int *list, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int8_t k = known[list[t]];
    if (k == -4 || k == -5) known[list[t]] = -2;
}
// count -3 and turn them into -2 (refinement.py:505-507); 16 voxels per thread
__global__ __launch_bounds__(TPB) void k_ec_finish(int8_t *known, long long N, unsigned long long *edges) {
    const long long base = ((long long)blockIdx.x * TPB + threadIdx.x) * 16;
    unsigned int cnt = 0;
    if (base + 16 <= N) {
        uint4 w = *reinterpret_cast<const uint4 *>(known + base);
        int8_t *b = reinterpret_cast<int8_t *>(&w);
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (b[k] == -3) { b[k] = -2; cnt++; }
        if (cnt) *reinterpret_cast<uint4 *>(known + base) = w;
    } else {
        for (long long k = base; k < N; k++)
            if (known[k] == -3) { known[k] = -2; cnt++; }
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if (threadIdx.x % XB_WAVE == 0 && cnt) atomicAdd(edges, (unsigned long long)cnt);
}

// ---------------------------------------------------------------------------------------------
// utils.charge_sum (utils.py:235-252): per-label sums over the owned slab.  LDS-privatised bins
// per block when the label count is small, global atomics otherwise.
// ---------------------------------------------------------------------------------------------
#define CS_BINS 1024
__global__ __launch_bounds__(TPB) void k_charge_sum_lds(Grid g, const double *__restrict__ rho,
                                                        const int *__restrict__ labels, int n_labels,
                                                        double *charge, unsigned long long *count, int per_thread) {
    __shared__ double sc[CS_BINS];
    __shared__ unsigned int sn[CS_BINS];
    for (int i = threadIdx.x; i < n_labels; i += TPB) { sc[i] = 0.; sn[i] = 0; }
    __syncthreads();
    const long long vbeg = (long long)g.x0 * g.nyz, vend = (long long)g.x1 * g.nyz;
    long long v = vbeg + (long long)blockIdx.x * TPB * per_thread + threadIdx.x;
    for (int k = 0; k < per_thread; k++, v += TPB) {
        if (v < vend) {
            const int a = labels[v];
            if (a >= 0 && a < n_labels) { atomicAdd(&sc[a], rho[v]); atomicAdd(&sn[a], 1u); }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_labels; i += TPB)
        if (sn[i]) { atomicAdd(&charge[i], sc[i]); atomicAdd(&count[i], (unsigned long long)sn[i]); }
}
__global__ __launch_bounds__(TPB) void k_charge_sum_glb(Grid g, const double *__restrict__ rho,
                                                        const int *__restrict__ labels, int n_labels,
                                                        double *charge, unsigned long long *count) {
    const long long vbeg = (long long)g.x0 * g.nyz, vend = (long long)g.x1 * g.nyz;
    const long long v = vbeg + (long long)blockIdx.x * TPB + threadIdx.x;
    if (v >= vend) return;
    const int a = labels[v];
    if (a >= 0 && a < n_labels) { atomicAdd(&charge[a], rho[v]); atomicAdd(&count[a], 1ull); }
}

// utils.surface_dist (utils.py:320-379) over the edge list: squared minimum-image distance of every
// edge voxel to the atom that owns it, reduced per atom with an integer atomicMin on the bit
// pattern (non-negative doubles order like their bits), so the minimum is exact and order-free.
__global__ __launch_bounds__(TPB) void k_surface_dist(GridL g, const int *__restrict__ labels,
                                                      const int *__restrict__ list, int n,
                                                      const double *__restrict__ lat, const double *__restrict__ atoms,
                                                      int n_atoms, unsigned long long *min_d2) {
    const int t = blockIdx.x * TPB + threadIdx.x;
    if (t >= n) return;
    const int v = list[t];
    const int a = labels[v];
    if (a < 0 || a >= n_atoms) return;
    const int p0 = v / g.nyz;
    const int r = v - p0 * g.nyz;
    const int p1 = r / g.nz, p2 = r - p1 * g.nz;
    double pc[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {  // utils.py:357-359
        pc[j] = lat[j] * (double)p0 / (double)g.nx;
        pc[j] += lat[3 + j] * (double)p1 / (double)g.ny;
        pc[j] += lat[6 + j] * (double)p2 / (double)g.nz;
    }
    double best = 1.7976931348623157e308;
    for (int x = -1; x < 2; x++)
        for (int y = -1; y < 2; y++)
            for (int z = -1; z < 2; z++) {
                double d2 = 0.;
#pragma unroll
                for (int j = 0; j < 3; j++) {  // utils.py:369-374
                    const double pbc = (lat[j] * (double)x + lat[3 + j] * (double)y) + lat[6 + j] * (double)z;
                    const double e = pc[j] - (atoms[3 * a + j] + pbc);
                    d2 = (j == 0) ? e * e : d2 + e * e;
                }
                if (d2 < best) best = d2;
            }
    atomicMin(&min_d2[a], (unsigned long long)__double_as_longlong(best));
}
// utils.volume_mask (utils.py:461-476)
__global__ __launch_bounds__(TPB) void k_volume_mask(const double *__restrict__ rho, const int *__restrict__ labels,
                                                     int vol_num, double *__restrict__ out, long long N) {
    const long long v = (long long)blockIdx.x * TPB + threadIdx.x;
    if (v < N) out[v] = (labels[v] == vol_num) ? rho[v] : 0.;
}
// sum of rho and count over the owned voxels whose label equals `value` (vacuum sums with a
// separate reference density, utils.py:396-400)
__global__ __launch_bounds__(TPB) void k_label_sum(Grid g, const double *__restrict__ rho, const int *__restrict__ labels,
                                                   int value, double *sum, unsigned long long *count) {
    const long long vbeg = (long long)g.x0 * g.nyz, vend = (long long)g.x1 * g.nyz;
    const long long v = vbeg + (long long)blockIdx.x * TPB + threadIdx.x;
    double s = 0.;
    unsigned int n = 0;
    if (v < vend && labels[v] == value) { s = rho[v]; n = 1; }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o); n += __shfl_down(n, o); }
    if (threadIdx.x % XB_WAVE == 0 && n) { atomicAdd(sum, s); atomicAdd(count, (unsigned long long)n); }
}

// utils.volume_assign (utils.py:404-421)
__global__ __launch_bounds__(TPB) void k_volume_assign(Grid g, int *labels, const int *__restrict__ swap, int n_swap) {
    const long long vbeg = (long long)g.x0 * g.nyz, vend = (long long)g.x1 * g.nyz;
    const long long v = vbeg + (long long)blockIdx.x * TPB + threadIdx.x;
    if (v >= vend) return;
    const int a = labels[v];
    if (a >= 0 && a < n_swap) labels[v] = swap[a];
}

// utils.dtype_change (utils.py:255-259): widen / narrow between the boundary dtype and int32
template <typename T>
__global__ void k_widen(const T *__restrict__ in, int *__restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int)in[i];
}
template <typename T>
__global__ void k_narrow(const int *__restrict__ in, T *__restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (T)in[i];
}

// =============================================================================================
// host side
// =============================================================================================
static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) return fail(XB_E_HIP, "%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
    } while (0)

struct TimedKernel {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double ms = 0.;
    long long launches = 0;
};

struct xb_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    Grid g{};
    bool has_grid = false;
    long long N = 0;
    int halo = 0;
    double *rho = nullptr;
    GradRec *grad = nullptr;   // gradient-field table, 32 B per voxel
    double *dist_dev = nullptr; // dist_mat on the device
    int *boxbuf = nullptr;      // seeds / box tables of the table build (BB_* layout)
    int n_boxes = 0;
    long long box_voxels = 0;
    int opt_boxes = 1;
    int opt_bricks = 1;
    int opt_dbg = 0;
    bool has_vacuum = true;    // false only when volumes_init proved there is no -1 label
    bool regions_pending = false;  // labels of certain bricks are written by the relabel pass
    bool buni_valid = false;       // per-brick label uniformity (in `st`) matches the resident labels
    int n_walk = 0;                // bricks on the walk list of the last assignment
    int table_margin = -1;         // planes of table each side of the slab (slabs); -1: whole grid
    bool table_prebuilt = false;   // xb_table_finish done: the next xb_assign_trace must not rebuild
    int table_stage = 0;           // windowed build: 1 = records + masks done, 2 = trapping regions done
    std::vector<int> window_seeds; // maxima found in the owned planes (windowed build)
    long long stat_ovf_assign = 0, stat_ovf_refine = 0;   // trajectories handed to the exact slow kernel
    int *blab = nullptr;        // brick labels of the trapping regions (inside `list`), or null
    int nbk[3] = {0, 0, 0};
    int opt_trace_tpb = 64;   // one wave per block: a finished wave frees its slot at once
    bool grad_valid = false;
    int *labels = nullptr;
    int8_t *known = nullptr;
    int *first = nullptr;      // n^3: min voxel per maximum, then rank per maximum
    int *list = nullptr;       // n^3 ints: compaction list (edges / changed voxels)
    int8_t *st = nullptr;      // per-list-entry status for edge_check
    void *stage = nullptr;     // staging for dtype conversion (N * 8 bytes max)
    size_t stage_bytes = 0;
    int *max_list = nullptr;   // maxima discovered (linear indices)
    int *max_aux = nullptr;
    int max_cap = 0;
    int *ovf_list = nullptr;
    int ovf_cap = 0;
    int *counters = nullptr;   // small device scratch: ints
    unsigned long long *counters64 = nullptr;
    double *dsum = nullptr;
    int *host_ints = nullptr;  // pinned
    std::vector<int> maxima_sorted;  // global, label order
    std::vector<int> local_max, local_first;
    bool first_clean = false;
    int list_n = 0;            // entries of `list` that hold the owned known == -2 voxels ...
    bool list_valid = false;   // ... when this is set (by xb_edge_find)
    bool timing = false;
    int opt_trace = 1;   // bit0: 4x4x4 brick per wave, bit1: XCD-aware block order
    TimedKernel tk[6];
    long long n_alloc = 0;
};

static inline unsigned nblocks(long long n) { return (unsigned)((n + TPB - 1) / TPB); }
static GridL light(const Grid &g) {
    GridL l;
    l.nx = g.nx; l.ny = g.ny; l.nz = g.nz; l.nyz = g.nyz;
    l.x0 = g.x0; l.x1 = g.x1; l.vx0 = g.vx0; l.vlen = g.vlen;
    l.wx0 = g.wx0; l.wlen = g.wlen;
    l.use24 = ((long long)g.nx * g.ny < (1 << 24)) && g.nz < (1 << 24);
    return l;
}

struct ScopedTimer {
    xb_ctx *c;
    int which;
    hipEvent_t a = nullptr, b = nullptr;
    ScopedTimer(xb_ctx *c_, int w) : c(c_), which(w) {
        if (c->timing) {
            hipEventCreate(&a);
            hipEventCreate(&b);
            hipEventRecord(a, c->stream);
        }
    }
    ~ScopedTimer() {
        if (c->timing) {
            hipEventRecord(b, c->stream);
            c->tk[which].pending.push_back({a, b});
        }
    }
};

extern "C" {

const char *xb_last_error(void) { return g_err.c_str(); }

int xb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int xb_create(int device, xb_ctx **out) {
    if (!out) return fail(XB_E_ARG, "xb_create: null out");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(XB_E_HIP, "xb_create: no HIP device visible (%s); libbader_hip has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(XB_E_ARG, "xb_create: device %d out of range [0,%d)", device, n);
    HIPCHK(hipSetDevice(device));
    xb_ctx *c = new xb_ctx();
    c->device = device;
    HIPCHK(hipStreamCreate(&c->stream));
    HIPCHK(hipMalloc(&c->counters, 64 * sizeof(int)));
    HIPCHK(hipMalloc(&c->counters64, 16 * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&c->dsum, 16 * sizeof(double)));
    HIPCHK(hipMalloc(&c->dist_dev, 27 * sizeof(double)));
    HIPCHK(hipMalloc(&c->boxbuf, (size_t)(1 << 20) * sizeof(int)));
    HIPCHK(hipHostMalloc(&c->host_ints, 64 * sizeof(long long)));
    *out = c;
    return XB_OK;
}

static void free_grid(xb_ctx *c) {
    hipFree(c->rho); hipFree(c->grad); hipFree(c->labels); hipFree(c->known); hipFree(c->first); hipFree(c->list);
    hipFree(c->st); hipFree(c->stage); hipFree(c->max_list); hipFree(c->max_aux); hipFree(c->ovf_list);
    c->rho = nullptr; c->grad = nullptr; c->grad_valid = false; c->labels = nullptr; c->known = nullptr; c->first = nullptr; c->list = nullptr;
    c->st = nullptr; c->stage = nullptr; c->max_list = nullptr; c->max_aux = nullptr; c->ovf_list = nullptr;
    c->n_alloc = 0; c->stage_bytes = 0;
}

void xb_destroy(xb_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (auto &t : c->tk)
        for (auto &p : t.pending) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    free_grid(c);
    hipFree(c->counters); hipFree(c->counters64); hipFree(c->dsum); hipFree(c->dist_dev); hipFree(c->boxbuf);
    hipHostFree(c->host_ints);
    hipStreamDestroy(c->stream);
    delete c;
}

int xb_sync(xb_ctx *c) {
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}
void *xb_stream(xb_ctx *c) { return (void *)c->stream; }

static void set_valid_range(xb_ctx *c) {
    Grid &g = c->g;
    const int own = g.x1 - g.x0;
    if (own + 2 * c->halo >= g.nx) { g.vx0 = 0; g.vlen = g.nx; }
    else {
        // labels valid on [x0-H, x1+H); known (flag + dilate) on [x0-H+2, x1+H-2)
        const int hv = c->halo - 2;
        g.vx0 = ((g.x0 - hv) % g.nx + g.nx) % g.nx;
        g.vlen = own + 2 * hv;
    }
}

int xb_set_grid(xb_ctx *c, const int64_t shape[3], const double dist_mat[27], const double T_grad[9],
                int64_t x0, int64_t x1) {
    if (!c || !shape) return fail(XB_E_ARG, "xb_set_grid: null argument");
    for (int j = 0; j < 3; j++)
        if (shape[j] < 3) return fail(XB_E_ARG, "xb_set_grid: every axis needs >= 3 voxels (got %lld)", (long long)shape[j]);
    const long long N = (long long)shape[0] * shape[1] * shape[2];
    if (N >= 2147483647LL) return fail(XB_E_LIMIT, "xb_set_grid: %lld voxels exceed the int32 index range", N);
    if (x0 < 0 || x1 > shape[0] || x0 >= x1) return fail(XB_E_ARG, "xb_set_grid: bad slab [%lld,%lld)", (long long)x0, (long long)x1);
    HIPCHK(hipSetDevice(c->device));
    if (N != c->n_alloc) {
        free_grid(c);
        HIPCHK(hipMalloc(&c->rho, N * sizeof(double)));
        HIPCHK(hipMalloc(&c->grad, N * sizeof(GradRec)));
        HIPCHK(hipMalloc(&c->labels, N * sizeof(int)));
        HIPCHK(hipMalloc(&c->known, N));
        HIPCHK(hipMalloc(&c->first, N * sizeof(int)));
        HIPCHK(hipMalloc(&c->list, N * sizeof(int)));
        HIPCHK(hipMalloc(&c->st, N));
        c->stage_bytes = (size_t)N * 8;
        HIPCHK(hipMalloc(&c->stage, c->stage_bytes));
        c->max_cap = (int)std::min<long long>(N, 1 << 22);
        HIPCHK(hipMalloc(&c->max_list, c->max_cap * sizeof(int)));
        HIPCHK(hipMalloc(&c->max_aux, c->max_cap * sizeof(int)));
        c->ovf_cap = (int)std::min<long long>(N, 1 << 22);
        HIPCHK(hipMalloc(&c->ovf_list, c->ovf_cap * sizeof(int)));
        c->n_alloc = N;
        c->first_clean = false;
    }
    Grid &g = c->g;
    if (g.nx != (int)shape[0] || g.ny != (int)shape[1] || g.nz != (int)shape[2]) c->grad_valid = false;
    g.nx = (int)shape[0]; g.ny = (int)shape[1]; g.nz = (int)shape[2];
    g.nyz = g.ny * g.nz;
    g.x0 = (int)x0; g.x1 = (int)x1;
    if (dist_mat) {
        memcpy(g.dist, dist_mat, sizeof g.dist);
        HIPCHK(hipMemcpyAsync(c->dist_dev, dist_mat, sizeof g.dist, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (T_grad && memcmp(g.T, T_grad, sizeof g.T) != 0) { memcpy(g.T, T_grad, sizeof g.T); c->grad_valid = false; }
    c->N = N;
    c->halo = (x0 == 0 && x1 == shape[0]) ? g.nx : 0;
    set_valid_range(c);
    g.wx0 = 0; g.wlen = g.nx;      // table window: whole grid unless xb_set_table_window says otherwise
    c->table_margin = -1;
    c->table_stage = 0;
    c->has_grid = true;
    c->maxima_sorted.clear();
    if (!c->first_clean) {
        k_fill<int><<<4096, TPB, 0, c->stream>>>(c->first, XB_INT_MAX, N);
        HIPCHK(hipGetLastError());
        c->first_clean = true;
    }
    return XB_OK;
}

int xb_set_halo(xb_ctx *c, int64_t halo) {
    if (!c || !c->has_grid) return fail(XB_E_STATE, "xb_set_halo: no grid");
    if (halo < 2) return fail(XB_E_ARG, "xb_set_halo: halo must be >= 2 planes");
    c->halo = (int)halo;
    set_valid_range(c);
    return XB_OK;
}

#define NEED_GRID(name) \
    if (!c || !c->has_grid) return fail(XB_E_STATE, name ": call xb_set_grid first"); \
    HIPCHK(hipSetDevice(c->device))

int xb_upload_density(xb_ctx *c, const double *rho_host) {
    NEED_GRID("xb_upload_density");
    c->grad_valid = false;
    HIPCHK(hipMemcpyAsync(c->rho, rho_host, c->N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}
int xb_download_density(xb_ctx *c, double *rho_host) {
    NEED_GRID("xb_download_density");
    HIPCHK(hipMemcpyAsync(rho_host, c->rho, c->N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}

int xb_synth_density(xb_ctx *c, const double lattice[9], const double *atoms5, int64_t n_atoms, double background) {
    NEED_GRID("xb_synth_density");
    if (n_atoms < 0 || n_atoms > 4096) return fail(XB_E_ARG, "xb_synth_density: bad atom count");
    c->grad_valid = false;
    double *tmp = (double *)c->stage;
    HIPCHK(hipMemcpyAsync(tmp, lattice, 9 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(tmp + 16, atoms5, n_atoms * 5 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    k_synth_density<<<nblocks(c->N), TPB, 0, c->stream>>>(c->g, tmp, tmp + 16, (int)n_atoms, background, c->rho);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}

static size_t dtype_size(int dtype) { return (dtype == XB_I8 || dtype == XB_I16 || dtype == XB_I32 || dtype == XB_I64) ? (size_t)dtype : 0; }

int xb_upload_labels(xb_ctx *c, const void *labels_host, int dtype) {
    NEED_GRID("xb_upload_labels");
    c->list_valid = false;
    c->has_vacuum = true;
    c->buni_valid = false;
    const size_t sz = dtype_size(dtype);
    if (!sz) return fail(XB_E_ARG, "xb_upload_labels: bad dtype code %d", dtype);
    if (dtype == XB_I32) {
        HIPCHK(hipMemcpyAsync(c->labels, labels_host, c->N * 4, hipMemcpyHostToDevice, c->stream));
    } else {
        HIPCHK(hipMemcpyAsync(c->stage, labels_host, c->N * sz, hipMemcpyHostToDevice, c->stream));
        if (dtype == XB_I8) k_widen<int8_t><<<nblocks(c->N), TPB, 0, c->stream>>>((const int8_t *)c->stage, c->labels, c->N);
        else if (dtype == XB_I16) k_widen<int16_t><<<nblocks(c->N), TPB, 0, c->stream>>>((const int16_t *)c->stage, c->labels, c->N);
        else k_widen<long long><<<nblocks(c->N), TPB, 0, c->stream>>>((const long long *)c->stage, c->labels, c->N);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}
int xb_download_labels(xb_ctx *c, void *labels_host, int dtype) {
    NEED_GRID("xb_download_labels");
    const size_t sz = dtype_size(dtype);
    if (!sz) return fail(XB_E_ARG, "xb_download_labels: bad dtype code %d", dtype);
    if (dtype == XB_I32) {
        HIPCHK(hipMemcpyAsync(labels_host, c->labels, c->N * 4, hipMemcpyDeviceToHost, c->stream));
    } else {
        if (dtype == XB_I8) k_narrow<int8_t><<<nblocks(c->N), TPB, 0, c->stream>>>(c->labels, (int8_t *)c->stage, c->N);
        else if (dtype == XB_I16) k_narrow<int16_t><<<nblocks(c->N), TPB, 0, c->stream>>>(c->labels, (int16_t *)c->stage, c->N);
        else k_narrow<long long><<<nblocks(c->N), TPB, 0, c->stream>>>(c->labels, (long long *)c->stage, c->N);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(labels_host, c->stage, c->N * sz, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}
int xb_upload_known(xb_ctx *c, const int8_t *known_host) {
    NEED_GRID("xb_upload_known");
    c->list_valid = false;
    HIPCHK(hipMemcpyAsync(c->known, known_host, c->N, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}
int xb_download_known(xb_ctx *c, int8_t *known_host) {
    NEED_GRID("xb_download_known");
    HIPCHK(hipMemcpyAsync(known_host, c->known, c->N, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}

int xb_vacuum_assign(xb_ctx *c, double vac_tol, double voxel_volume, double *vac_charge, double *vac_volume) {
    NEED_GRID("xb_vacuum_assign");
    c->buni_valid = false;
    if (vac_tol != vac_tol) {
        // vacuum_tol=None reaches the reference's sweep as NaN (interface.py:459): `rho <= NaN` is never
        // true, so the result is all-zero labels and zero vacuum charge/volume -- no need to read rho
        HIPCHK(hipMemsetAsync(c->labels, 0, c->N * sizeof(int), c->stream));
        c->has_vacuum = false;
        if (vac_charge) *vac_charge = 0.;
        if (vac_volume) *vac_volume = 0.;
        return XB_OK;
    }
    HIPCHK(hipMemsetAsync(c->dsum, 0, sizeof(double), c->stream));
    HIPCHK(hipMemsetAsync(c->counters64, 0, sizeof(unsigned long long), c->stream));
    k_vacuum_assign<<<nblocks(c->N), TPB, 0, c->stream>>>(c->g, c->rho, c->labels, vac_tol, c->dsum, c->counters64);
    HIPCHK(hipGetLastError());
    double s;
    unsigned long long n;
    HIPCHK(hipMemcpyAsync(&s, c->dsum, sizeof s, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&n, c->counters64, sizeof n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->has_vacuum = true;   // (the count covers the owned slab only: stay conservative)
    if (vac_charge) *vac_charge = s * voxel_volume;  // utils.py:400
    if (vac_volume) *vac_volume = (double)n * voxel_volume;
    return XB_OK;
}

static int read_counter(xb_ctx *c, int idx, int *out);
static GridL light(const Grid &g);

// layout of the small device int buffer used by the table build (c->boxbuf)
enum { BB_SEEDS = 0, BB_SEED_CAP = 4096, BB_MXYZ = 4096, BB_RCAP = 4352, BB_BOXMAX = 4608, BB_EXT = 5632, BB_BAD = 8192,
       BB_TOTAL = 1 << 20, XB_BOX_SEEDS_MAX = 64 };

// (re)build the gradient-field table from the resident density; with `boxes`, also find and stamp
// the trapping boxes around the 26-neighbour maxima (k_box_scan)
static bool table_windowed(const xb_ctx *c) { return c->g.wlen < c->g.nx; }
static int table_regions(xb_ctx *c, std::vector<int> seeds, bool bricks);

static int ensure_grad(xb_ctx *c, bool force, bool boxes) {
    if (c->grad_valid && !force) return XB_OK;
    const Grid &g = c->g;
    ScopedTimer t(c, 4);
    HIPCHK(hipMemsetAsync(c->counters + 9, 0, sizeof(int), c->stream));
    // brick growth needs a grid made of whole 8^3 bricks; its scratch is carved from `list`
    const bool bricks = boxes && c->opt_boxes && c->opt_bricks && g.nx % BRK == 0 && g.ny % BRK == 0 &&
                        g.nz % BRK == 0 && 5LL * (c->N / (BRK * BRK * BRK)) <= c->N;
    const int nbr_all = (int)(c->N / (BRK * BRK * BRK));
    if (table_windowed(c) && boxes && c->opt_boxes && !bricks)
        return fail(XB_E_STATE, "a table window needs brick growth (grid of whole 8^3 bricks)");
    {
        const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
        dim3 grid((g.nz + GT_Z - 1) / GT_Z, (g.ny + GT_Y - 1) / GT_Y, (g.wlen + GT_X - 1) / GT_X);
        ScopedTimer tk(c, 5);
        k_grad_field<<<grid, TPB, 0, c->stream>>>(g, c->rho, c->grad, c->boxbuf + BB_SEEDS, c->counters + 9,
                                                 BB_SEED_CAP, small, bricks ? c->list + nbr_all : nullptr);
    }
    HIPCHK(hipGetLastError());
    c->grad_valid = true;
    c->n_boxes = 0;
    c->box_voxels = 0;
    c->blab = nullptr;
    c->table_stage = 1;
    if (!boxes || !c->opt_boxes) return XB_OK;
    int ns = 0;
    if (int rc = read_counter(c, 9, &ns)) return rc;
    if (ns > BB_SEED_CAP) ns = XB_BOX_SEEDS_MAX + 1;  // list overflowed: far too many maxima for boxes anyway
    std::vector<int> seeds(std::max(ns, 0));
    if (ns > 0 && ns <= XB_BOX_SEEDS_MAX) {
        HIPCHK(hipMemcpyAsync(seeds.data(), c->boxbuf + BB_SEEDS, ns * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (table_windowed(c)) {
        // slabs: the trapping regions need the maxima and brick masks of ALL ranks; keep what this
        // rank owns and let the scheduler exchange (xb_table_local_seeds / xb_brick_masks / xb_table_finish)
        c->window_seeds.clear();
        if (ns > XB_BOX_SEEDS_MAX) c->window_seeds.assign(XB_BOX_SEEDS_MAX + 1, -1);  // "too many" marker
        else
            for (int v : seeds)
                if (v / g.nyz >= g.x0 && v / g.nyz < g.x1) c->window_seeds.push_back(v);
        return XB_OK;
    }
    if (ns < 1 || ns > XB_BOX_SEEDS_MAX) { c->table_stage = 2; return XB_OK; }  // many maxima (noisy data): plain tracing
    const int rc = table_regions(c, seeds, bricks);
    c->table_stage = 2;
    return rc;
}

// trapping regions from the list of all 26-neighbour maxima: closed seed cubes, then brick growth
static int table_regions(xb_ctx *c, std::vector<int> seeds, bool bricks) {
    const Grid &g = c->g;
    const int ns = (int)seeds.size();
    const int nbr_all = (int)(c->N / (BRK * BRK * BRK));
    (void)nbr_all;
    std::vector<int> mxyz(3 * ns), rcap(ns);
    std::sort(seeds.begin(), seeds.end());  // atomic append order is arbitrary: make box ids deterministic
    for (int m = 0; m < ns; m++) {
        mxyz[3 * m] = seeds[m] / g.nyz;
        mxyz[3 * m + 1] = (seeds[m] % g.nyz) / g.nz;
        mxyz[3 * m + 2] = seeds[m] % g.nz;
    }
    auto mi = [](int t, int n) { int a = std::abs(t) % n; return std::min(a, n - a); };
    const int rmax = std::min(std::min(g.nx, g.ny), g.nz) / 2 - 2;  // a box must not wrap onto itself
    int stride = 4;
    for (int m = 0; m < ns; m++) {
        int cap = rmax;
        for (int o = 0; o < ns; o++)
            if (o != m) {  // exactly one maximum per box: stay clear of the nearest other maximum
                const int d = std::max(std::max(mi(mxyz[3 * m] - mxyz[3 * o], g.nx), mi(mxyz[3 * m + 1] - mxyz[3 * o + 1], g.ny)),
                                       mi(mxyz[3 * m + 2] - mxyz[3 * o + 2], g.nz));
                cap = std::min(cap, d - 1);
            }
        rcap[m] = std::max(cap, 0);
        stride = std::max(stride, rcap[m] + 4);
    }
    if ((long long)ns * stride > BB_TOTAL - BB_BAD) return XB_OK;
    HIPCHK(hipMemcpyAsync(c->boxbuf + BB_MXYZ, mxyz.data(), 3 * ns * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->boxbuf + BB_RCAP, rcap.data(), ns * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->boxbuf + BB_BAD, 0, (size_t)ns * stride * sizeof(int), c->stream));
    // shells in batches of K radii; a box stops growing after a batch without any closed radius.
    // With brick growth available the cubes are only seeds: one batch (R <= K) is enough.
    const int K = 32;
    std::vector<int> best(ns, 0), cap_now(rcap), bad((size_t)ns * stride);
    int rcap_max = 0;
    for (int m = 0; m < ns; m++) rcap_max = std::max(rcap_max, rcap[m]);
    for (int rlo = 0; rlo <= rcap_max; rlo += K + 1) {
        int rtop = 0;
        for (int m = 0; m < ns; m++) rtop = std::max(rtop, std::min(rlo + K, cap_now[m]));
        if (rtop < rlo) break;
        const long long w = 2LL * rtop + 1;
        dim3 grid(nblocks(w * w * w), ns);
        if (table_windowed(c))  // a seed cube may lie outside this rank's table window: ranges from rho
            k_box_shells_rho<<<grid, TPB, 0, c->stream>>>(g, c->rho, c->boxbuf + BB_MXYZ, c->boxbuf + BB_RCAP, rlo, K,
                                                          c->boxbuf + BB_BAD, stride);
        else
            k_box_shells<<<grid, TPB, 0, c->stream>>>(light(g), c->grad, c->boxbuf + BB_MXYZ, c->boxbuf + BB_RCAP, rlo, K,
                                                      c->boxbuf + BB_BAD, stride);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(bad.data(), c->boxbuf + BB_BAD, bad.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        bool any = false;
        for (int m = 0; m < ns; m++) {
            if (cap_now[m] < rlo) continue;
            bool found = false;
            for (int R = std::max(rlo, 1); R <= std::min(rlo + K, cap_now[m]); R++)
                if (!bad[(size_t)m * stride + R]) { best[m] = R; found = true; }
            if (found) any = true;
            else cap_now[m] = rlo - 1;  // stop growing this box
        }
        if (!any || bricks) break;
        HIPCHK(hipMemcpyAsync(c->boxbuf + BB_RCAP, cap_now.data(), ns * sizeof(int), hipMemcpyHostToDevice, c->stream));
    }
    std::vector<int> box_max, bx, br;
    for (int m = 0; m < ns; m++) {
        if (best[m] < 1 || (int)box_max.size() >= XB_MAX_BOXES) continue;
        box_max.push_back(seeds[m]);
        for (int k = 0; k < 3; k++) bx.push_back(mxyz[3 * m + k]);
        br.push_back(best[m]);
        if (!bricks) {  // no brick labels: the cube is stamped into the keys
            const long long w = 2LL * best[m] + 1, nvox = w * w * w;
            k_box_stamp<<<nblocks(nvox), TPB, 0, c->stream>>>(light(g), c->grad, mxyz[3 * m], mxyz[3 * m + 1], mxyz[3 * m + 2],
                                                             best[m], (int)box_max.size());
            c->box_voxels += nvox;
        }
    }
    HIPCHK(hipGetLastError());
    const int nbx = (int)box_max.size();
    if (nbx)
        HIPCHK(hipMemcpyAsync(c->boxbuf + BB_BOXMAX, box_max.data(), nbx * sizeof(int), hipMemcpyHostToDevice, c->stream));
    // grow the certain regions brick by brick from the bricks inside the seed cubes
    if (nbx && bricks) {
        const int nb0 = g.nx / BRK, nb1 = g.ny / BRK, nb2 = g.nz / BRK, nbr = nb0 * nb1 * nb2;
        // scratch carved from `list` (free during an assignment): seed labels, brick masks, two label buffers
        int *seed = c->list, *bmask = c->list + nbr, *buf[2] = {c->list + 2 * nbr, c->list + 3 * nbr};
        HIPCHK(hipMemcpyAsync(c->boxbuf + BB_MXYZ, bx.data(), bx.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->boxbuf + BB_RCAP, br.data(), br.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        k_brick_seed<<<(nbr + 255) / 256, 256, 0, c->stream>>>(light(g), nb0, nb1, nb2, nbx, c->boxbuf + BB_MXYZ,
                                                             c->boxbuf + BB_RCAP, seed);
        HIPCHK(hipMemcpyAsync(buf[0], seed, nbr * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
        int cur = 0;
        const int max_rounds = 8 * ((2 * (nb0 + nb1 + nb2) + 15) / 8);  // a multiple of the polling period
        bool kill_converged = false;
        for (int phase = 0; phase < 2; phase++) {  // 0: propagate provisional labels, 1: kill violators
            for (int round = 1; round <= max_rounds; round++) {
                if ((round & 7) == 1) HIPCHK(hipMemsetAsync(c->counters + 11, 0, sizeof(int), c->stream));
                if (phase == 0)
                    k_brick_propagate<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nb0, nb1, nb2, bmask, buf[cur], buf[1 - cur], c->counters + 11);
                else
                    k_brick_kill<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nb0, nb1, nb2, bmask, seed, buf[cur], buf[1 - cur], c->counters + 11);
                cur = 1 - cur;
                if ((round & 7) == 0) {  // poll the change flag of the last eight rounds
                    HIPCHK(hipGetLastError());
                    int ch = 0;
                    if (int rc = read_counter(c, 11, &ch)) return rc;
                    if (!ch) {
                        if (phase == 1) kill_converged = true;
                        break;
                    }
                }
            }
        }
        // only a FIXPOINT of the kill iteration is closed under every move; without it fall back to
        // the seed cubes, which are trapping regions on their own
        int *blab = kill_converged ? buf[cur] : seed;
        HIPCHK(hipMemsetAsync(c->counters + 11, 0, sizeof(int), c->stream));
        k_count_positive<<<(nbr + 255) / 256, 256, 0, c->stream>>>(blab, nbr, c->counters + 11);
        int ncertain = 0;
        if (int rc = read_counter(c, 11, &ncertain)) return rc;
        c->box_voxels = (long long)ncertain * BRK * BRK * BRK;
        c->blab = blab;
        c->nbk[0] = nb0; c->nbk[1] = nb1; c->nbk[2] = nb2;
    }
    HIPCHK(hipStreamSynchronize(c->stream));  // host vectors must outlive the copies
    c->n_boxes = (int)box_max.size();
    return XB_OK;
}

static int read_counter(xb_ctx *c, int idx, int *out) {
    HIPCHK(hipMemcpyAsync(c->host_ints, c->counters + idx, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *out = c->host_ints[0];
    return XB_OK;
}

// run the exact slow kernel over ovf_list[0..n) in chunks
static int run_slow(xb_ctx *c, int n, int refine) {
    const int lmax = 1 << 15, chunk = 2048;
    int *path = nullptr;
    HIPCHK(hipMalloc(&path, (size_t)chunk * lmax * sizeof(int)));
    HIPCHK(hipMemsetAsync(c->counters + 8, 0, sizeof(int), c->stream));  // err
    for (int o = 0; o < n; o += chunk) {
        const int m = std::min(chunk, n - o);
        k_trace_slow<<<(m + 63) / 64, 64, 0, c->stream>>>(c->g, c->rho, c->labels, c->known, c->known,
                                                         c->ovf_list + o, m, path, lmax, refine, c->first,
                                                         c->max_list, c->counters + 0, c->max_cap,
                                                         c->counters + 2, c->counters + 3, c->counters + 8);
    }
    hipError_t e = hipGetLastError();
    int err = 0;
    int rc = read_counter(c, 8, &err);
    hipFree(path);
    if (e != hipSuccess) return fail(XB_E_HIP, "k_trace_slow: %s", hipGetErrorString(e));
    if (rc) return rc;
    if (err) return fail(XB_E_LIMIT, "trajectory longer than %d voxels", lmax);
    return XB_OK;
}

int xb_assign_trace(xb_ctx *c, int method, int64_t *n_local) {
    NEED_GRID("xb_assign_trace");
    const Grid &g = c->g;
    const long long own = (long long)(g.x1 - g.x0) * g.nyz;
    HIPCHK(hipMemsetAsync(c->counters, 0, 16 * sizeof(int), c->stream));
    if (!c->first_clean) {  // a previous assignment did not finish: `first` may hold stale minima
        k_fill<int><<<4096, TPB, 0, c->stream>>>(c->first, XB_INT_MAX, c->N);
        HIPCHK(hipGetLastError());
    }
    c->first_clean = false;
    c->regions_pending = false;
    if (method == XB_METHOD_NEARGRID) {
        const int maxsteps = 8 * (g.nx + g.ny + g.nz) + 64;
        // the table is a pure function of the resident density, but it is part of the assignment
        // work: rebuilt on every call, never carried over from a previous assignment
        if (c->table_prebuilt) c->table_prebuilt = false;   // built by xb_table_build/xb_table_finish just now
        else {
            if (table_windowed(c)) return fail(XB_E_STATE, "windowed table: call xb_table_build / xb_table_finish first");
            if (int rc = ensure_grad(c, true, true)) return rc;
        }
        {
            ScopedTimer t(c, 0);
            const int opt = c->opt_trace;
            const int tpb = c->opt_trace_tpb;
            const bool slab_bricks = (g.x0 % 8 == 0) && (g.x1 % 8 == 0);
            if (c->blab && slab_bricks) {
                // trapping regions known per brick: fill them in one sweep, trace only the rest
                const int nbr = c->nbk[0] * c->nbk[1] * c->nbk[2];
                int *walk = c->blab + nbr;  // next scratch slice of `list` (see ensure_grad)
                HIPCHK(hipMemsetAsync(c->counters + 13, 0, sizeof(int), c->stream));
                k_brick_walk_list<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, (g.x0 / 8) * c->nbk[1] * c->nbk[2],
                                                                         (g.x1 / 8) * c->nbk[1] * c->nbk[2], c->blab, walk, c->counters + 13);
                if (c->has_vacuum) {
                    k_fill_certain<<<nblocks(own), TPB, 0, c->stream>>>(light(g), c->blab, c->nbk[1], c->nbk[2],
                                                                        c->boxbuf + BB_BOXMAX, c->labels, c->first, c->max_list,
                                                                        c->counters + 0, c->max_cap);
                } else {
                    k_note_certain_bricks<<<(nbr + 255) / 256, 256, 0, c->stream>>>(
                        light(g), c->nbk[0], c->nbk[1], c->nbk[2], (g.x0 / 8) * c->nbk[1] * c->nbk[2],
                        (g.x1 / 8) * c->nbk[1] * c->nbk[2], c->blab, c->boxbuf + BB_BOXMAX, c->first, c->max_list,
                        c->counters + 0, c->max_cap);
                    c->regions_pending = true;
                }
                int nwalk = 0;
                if (int rc = read_counter(c, 13, &nwalk)) return rc;
                c->n_walk = nwalk;
                if (nwalk) {
                    const long long waves = 8LL * nwalk;
                    k_ng_trace<2><<<(unsigned)((waves + tpb / XB_WAVE - 1) / (tpb / XB_WAVE)), tpb, 0, c->stream>>>(
                        light(g), c->grad, c->boxbuf + BB_BOXMAX, c->blab, c->nbk[1], c->nbk[2], walk, nwalk, c->labels,
                        c->first, c->max_list, c->counters + 0, c->max_cap, c->ovf_list, c->counters + 1, c->ovf_cap,
                        maxsteps, opt);
                }
            } else {
                const long long waves = (opt & 1)
                    ? (long long)((g.x1 - g.x0 + 3) / 4) * ((g.ny + 3) / 4) * ((g.nz + 3) / 4)
                    : (long long)(g.x1 - g.x0) * g.ny * ((g.nz + 63) / 64);
                k_ng_trace<2><<<(unsigned)((waves + tpb / XB_WAVE - 1) / (tpb / XB_WAVE)), tpb, 0, c->stream>>>(
                    light(g), c->grad, c->boxbuf + BB_BOXMAX, c->blab, c->nbk[1], c->nbk[2], nullptr, 0, c->labels,
                    c->first, c->max_list, c->counters + 0, c->max_cap, c->ovf_list, c->counters + 1, c->ovf_cap,
                    maxsteps, opt);
            }
        }
        HIPCHK(hipGetLastError());
        int novf = 0;
        if (int rc = read_counter(c, 1, &novf)) return rc;
        if (novf > c->ovf_cap) return fail(XB_E_LIMIT, "%d trajectories need the slow path (cap %d)", novf, c->ovf_cap);
        c->stat_ovf_assign += novf;
        if (novf > 0)
            if (int rc = run_slow(c, novf, 0)) return rc;
    } else if (method == XB_METHOD_ONGRID) {
        {
            ScopedTimer t(c, 1);
            k_og_pointer<<<nblocks(c->N), TPB, 0, c->stream>>>(g, c->rho, c->labels);
        }
        HIPCHK(hipGetLastError());
        for (int it = 0; it < 64; it++) {
            HIPCHK(hipMemsetAsync(c->counters + 4, 0, sizeof(int), c->stream));
            k_og_jump<<<nblocks(c->N), TPB, 0, c->stream>>>(g, c->labels, c->counters + 4);
            HIPCHK(hipGetLastError());
            int nd = 0;
            if (int rc = read_counter(c, 4, &nd)) return rc;
            if (!nd) break;
            if (it == 63) return fail(XB_E_LIMIT, "ongrid pointer jumping did not converge");
        }
        k_note_roots<<<nblocks(own), TPB, 0, c->stream>>>(g, c->labels, c->first, c->max_list, c->counters + 0, c->max_cap);
        HIPCHK(hipGetLastError());
    } else
        return fail(XB_E_ARG, "xb_assign: unknown method %d", method);
    int nmax = 0;
    if (int rc = read_counter(c, 0, &nmax)) return rc;
    if (nmax > c->max_cap) return fail(XB_E_LIMIT, "%d maxima exceed the table capacity %d", nmax, c->max_cap);
    c->local_max.resize(nmax);
    c->local_first.resize(nmax);
    if (nmax) {
        k_gather_first<<<(nmax + 255) / 256, 256, 0, c->stream>>>(c->first, c->max_list, nmax, c->max_aux);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(c->local_max.data(), c->max_list, nmax * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(c->local_first.data(), c->max_aux, nmax * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (n_local) *n_local = nmax;
    return XB_OK;
}

int xb_assign_local_table(xb_ctx *c, int64_t *max_idx, int64_t *first_idx, int64_t capacity) {
    NEED_GRID("xb_assign_local_table");
    if ((int64_t)c->local_max.size() > capacity) return fail(XB_E_ARG, "xb_assign_local_table: capacity too small");
    for (size_t i = 0; i < c->local_max.size(); i++) { max_idx[i] = c->local_max[i]; first_idx[i] = c->local_first[i]; }
    return XB_OK;
}

int xb_assign_finish(xb_ctx *c, const int64_t *max_idx_sorted, int64_t n_global) {
    NEED_GRID("xb_assign_finish");
    if (n_global > c->max_cap) return fail(XB_E_LIMIT, "xb_assign_finish: too many maxima");
    c->maxima_sorted.resize(n_global);
    for (int64_t i = 0; i < n_global; i++) c->maxima_sorted[i] = (int)max_idx_sorted[i];
    const Grid &g = c->g;
    const long long own = (long long)(g.x1 - g.x0) * g.nyz;
    if (n_global) {
        HIPCHK(hipMemcpyAsync(c->max_aux, c->maxima_sorted.data(), n_global * sizeof(int), hipMemcpyHostToDevice, c->stream));
        k_set_rank<<<(unsigned)((n_global + 255) / 256), 256, 0, c->stream>>>(c->first, c->max_aux, (int)n_global);
        HIPCHK(hipGetLastError());
    }
    c->buni_valid = false;
    if (c->regions_pending && c->blab) {
        k_relabel_regions<<<nblocks(own), TPB, 0, c->stream>>>(light(g), c->labels, c->first, c->blab, c->nbk[1], c->nbk[2],
                                                               c->boxbuf + BB_BOXMAX);
        if (g.x1 - g.x0 == g.nx) {  // one slab: the per-brick label uniformity edge_find wants comes for free
            const int nbr = c->nbk[0] * c->nbk[1] * c->nbk[2];
            int *buni = reinterpret_cast<int *>(c->st);
            k_buni_from_regions<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, c->blab, c->boxbuf + BB_BOXMAX, c->first, buni);
            if (c->n_walk)
                k_label_uniform_list<<<c->n_walk, TPB, 0, c->stream>>>(light(g), c->labels, c->nbk[1], c->nbk[2],
                                                                      c->blab + nbr, c->n_walk, buni);
            c->buni_valid = true;
        }
    } else
        k_relabel<<<nblocks(own), TPB, 0, c->stream>>>(g, c->labels, c->first);
    c->regions_pending = false;
    HIPCHK(hipGetLastError());
    if (n_global) {  // leave `first` clean (INT_MAX everywhere) for the next assignment
        k_reset_first<<<(unsigned)((n_global + 255) / 256), 256, 0, c->stream>>>(c->first, c->max_aux, (int)n_global);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->first_clean = true;
    return XB_OK;
}

int xb_assign(xb_ctx *c, int method, int64_t *n_maxima) {
    int64_t n = 0;
    if (int rc = xb_assign_trace(c, method, &n)) return rc;
    // numbering: rank of the smallest voxel index reaching each maximum (thread_handlers.py:59-65
    // numbers maxima in the order the C-order scan discovers them)
    std::vector<int> order(n);
    for (int i = 0; i < n; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return c->local_first[a] < c->local_first[b]; });
    std::vector<int64_t> sorted(n);
    for (int i = 0; i < n; i++) sorted[i] = c->local_max[order[i]];
    if (int rc = xb_assign_finish(c, sorted.data(), n)) return rc;
    if (n_maxima) *n_maxima = n;
    return XB_OK;
}

int xb_get_maxima(xb_ctx *c, int64_t *maxima_out, int64_t capacity) {
    NEED_GRID("xb_get_maxima");
    if ((int64_t)c->maxima_sorted.size() > capacity) return fail(XB_E_ARG, "xb_get_maxima: capacity too small");
    const Grid &g = c->g;
    for (size_t i = 0; i < c->maxima_sorted.size(); i++) {
        const int m = c->maxima_sorted[i];
        const int x = m / g.nyz, r = m - x * g.nyz;
        maxima_out[3 * i] = x; maxima_out[3 * i + 1] = r / g.nz; maxima_out[3 * i + 2] = r % g.nz;
    }
    return XB_OK;
}

// planes [x0-ext, x1+ext) clipped to the grid size; returns start plane (mod nx) and count
static void plane_range(const Grid &g, int ext, int &xa, int &np) {
    const int own = g.x1 - g.x0;
    if (own + 2 * ext >= g.nx) { xa = 0; np = g.nx; }
    else { xa = ((g.x0 - ext) % g.nx + g.nx) % g.nx; np = own + 2 * ext; }
}

int xb_edge_find(xb_ctx *c, int64_t *edges) {
    NEED_GRID("xb_edge_find");
    const Grid &g = c->g;
    const bool whole = (g.x1 - g.x0 == g.nx);
    if (!whole && c->halo < 2) return fail(XB_E_STATE, "xb_edge_find: slab needs a label halo (xb_set_halo)");
    int xa, np, xb_, npd;
    // flags need labels one plane further out, the dilation needs flags one plane further out;
    // a halo that wraps the whole grid makes every plane valid
    const bool all = whole || (g.x1 - g.x0) + 2 * c->halo >= g.nx;
    plane_range(g, all ? g.nx : c->halo - 1, xa, np);
    plane_range(g, all ? g.nx : c->halo - 2, xb_, npd);
    HIPCHK(hipMemsetAsync(c->counters + 5, 0, sizeof(int), c->stream));
    {
        ScopedTimer t(c, 2);
        const GridL gl = light(g);
        const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
        int *buni = nullptr;
        if (g.nx % 8 == 0 && g.ny % 8 == 0 && g.nz % 8 == 0) {  // whole bricks: per-brick label uniformity first
            buni = reinterpret_cast<int *>(c->st);             // N bytes >= N/512 ints; edge_check reuses st later
            if (!c->buni_valid)
                k_label_uniform<<<(unsigned)(c->N / 512), TPB, 0, c->stream>>>(gl, c->labels, g.ny / 8, g.nz / 8, buni);
        }
        dim3 grid((g.nz + ET_Z - 1) / ET_Z, (g.ny + ET_Y - 1) / ET_Y, (np + ET_X - 1) / ET_X);
        k_edge_flag_tiled<<<grid, TPB, 0, c->stream>>>(gl, c->rho, c->labels, c->known, xa, np, c->list,
                                                       c->counters + 5, small, buni, c->grad_valid ? c->grad : nullptr);
        if (!whole)
            k_edge_dilate<<<nblocks((long long)npd * g.nyz), TPB, 0, c->stream>>>(g, c->known, xb_, npd, -2);
    }
    HIPCHK(hipGetLastError());
    int n = 0;
    if (int rc = read_counter(c, 5, &n)) return rc;
    if (whole && n) {  // one slab: the list holds every edge, dilate from it
        ScopedTimer t(c, 2);
        k_edge_dilate_list<<<nblocks(n), TPB, 0, c->stream>>>(light(g), c->known, c->list, n);
        HIPCHK(hipGetLastError());
    }
    c->list_n = n;           // the edge list stays valid until `known` changes
    c->list_valid = true;
    if (edges) *edges = (int64_t)n;
    return XB_OK;
}

static int compact(xb_ctx *c, int value, int *n_out) {
    const Grid &g = c->g;
    const long long own = (long long)(g.x1 - g.x0) * g.nyz;
    HIPCHK(hipMemsetAsync(c->counters + 5, 0, sizeof(int), c->stream));
    k_compact_known16<<<nblocks((own + 15) / 16), TPB, 0, c->stream>>>(light(g), c->known, value, c->list, c->counters + 5);
    HIPCHK(hipGetLastError());
    return read_counter(c, 5, n_out);
}

static int refine_trace_impl(xb_ctx *c, int flag, int64_t *changed, int64_t *escaped);
int xb_refine_trace(xb_ctx *c, int64_t *changed, int64_t *escaped) { return refine_trace_impl(c, -2, changed, escaped); }
int xb_refine_trace_escaped(xb_ctx *c, int64_t *changed, int64_t *escaped) { return refine_trace_impl(c, -6, changed, escaped); }
static int refine_trace_impl(xb_ctx *c, int flag, int64_t *changed, int64_t *escaped) {
    NEED_GRID("xb_refine_trace");
    const Grid &g = c->g;
    int n = 0;
    if (c->list_valid && flag == -2) n = c->list_n;
    else if (int rc = compact(c, flag, &n)) return rc;
    c->list_valid = false;  // the retrace rewrites known
    c->buni_valid = false;  // ... and may relabel edge voxels; st is also edge_check's scratch
    HIPCHK(hipMemsetAsync(c->counters, 0, 4 * sizeof(int), c->stream));
    if (n) {
        const int maxsteps = 8 * (g.nx + g.ny + g.nz) + 64;
        if (int rc = ensure_grad(c, false, false)) return rc;
        {
            ScopedTimer t(c, 3);
            k_refine_trace<2><<<nblocks(n), TPB, 0, c->stream>>>(light(g), c->grad, c->labels, c->known, c->list, n,
                                                                c->counters + 2, c->counters + 3, c->ovf_list,
                                                                c->counters + 1, c->ovf_cap, maxsteps);
        }
        HIPCHK(hipGetLastError());
        int novf = 0;
        if (int rc = read_counter(c, 1, &novf)) return rc;
        if (novf > c->ovf_cap) return fail(XB_E_LIMIT, "%d retraces need the slow path (cap %d)", novf, c->ovf_cap);
        c->stat_ovf_refine += novf;
        if (novf > 0)
            if (int rc = run_slow(c, novf, 1)) return rc;
    }
    HIPCHK(hipMemcpyAsync(c->host_ints, c->counters + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (changed) *changed = c->host_ints[0];
    if (escaped) *escaped = c->host_ints[1];
    return XB_OK;
}

int xb_edge_check(xb_ctx *c, int64_t *checked, int64_t *edges) {
    NEED_GRID("xb_edge_check");
    const Grid &g = c->g;
    c->list_valid = false;
    c->buni_valid = false;
    if (g.x1 - g.x0 != g.nx) return fail(XB_E_STATE, "xb_edge_check: 'changed' mode is single-slab only; slabs use mode 'all'");
    int n = 0;
    if (int rc = compact(c, -2, &n)) return rc;
    if (checked) *checked = 0;
    if (edges) *edges = 0;
    if (!n) return XB_OK;
    HIPCHK(hipMemsetAsync(c->st, 0, n, c->stream));
    {
        // work lists: two halves of `first` would clash with the numbering table, so borrow the
        // dtype staging buffer (N*8 bytes >= 2 lists of n ints); counters 6/7 ping-pong
        int *wl[2] = {(int *)c->stage, (int *)c->stage + n};
        HIPCHK(hipMemcpyAsync(c->counters + 6, &n, sizeof(int), hipMemcpyHostToDevice, c->stream));
        int cur = 0;
        const unsigned grid = (unsigned)std::min<long long>(nblocks(n), 2048);
        for (int round = 0;; round++) {
            HIPCHK(hipMemsetAsync(c->counters + 6 + (1 - cur), 0, sizeof(int), c->stream));
            k_ec_decide<<<grid, TPB, 0, c->stream>>>(g, c->rho, c->labels, c->known, c->list, c->st, wl[cur],
                                                     c->counters + 6 + cur, wl[1 - cur], c->counters + 6 + (1 - cur),
                                                     round == 0);
            cur = 1 - cur;
            if ((round & 31) == 31 || round < 2) {
                HIPCHK(hipGetLastError());
                int und = 0;
                if (int rc = read_counter(c, 6 + cur, &und)) return rc;
                if (!und) break;
            }
            if (round > n + 64 || round > 200000) return fail(XB_E_LIMIT, "xb_edge_check: greedy resolution did not converge");
        }
    }
    HIPCHK(hipMemsetAsync(c->counters64, 0, 2 * sizeof(unsigned long long), c->stream));
    k_ec_apply<<<nblocks(n), TPB, 0, c->stream>>>(g, c->rho, c->labels, c->known, c->list, n, c->st, c->counters64 + 1);
    k_ec_restore<<<nblocks(n), TPB, 0, c->stream>>>(c->known, c->list, n);
    k_edge_dilate<<<nblocks(c->N), TPB, 0, c->stream>>>(g, c->known, 0, g.nx, -3);
    k_ec_finish<<<nblocks((c->N + 15) / 16), TPB, 0, c->stream>>>(c->known, c->N, c->counters64);
    HIPCHK(hipGetLastError());
    unsigned long long r[2];
    HIPCHK(hipMemcpyAsync(r, c->counters64, sizeof r, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (edges) *edges = (int64_t)r[0];
    if (checked) *checked = (int64_t)(r[1] + r[0]);  // refinement.py:479 + 504
    return XB_OK;
}

int xb_refine(xb_ctx *c, int mode, int64_t iters, int64_t *log, int64_t log_capacity, int64_t *n_iters) {
    NEED_GRID("xb_refine");
    if (n_iters) *n_iters = 0;
    if (iters == 0) return XB_OK;  // thread_handlers.py:146-147
    int64_t edges = 0, changed = 0, esc = 0, checked = 0;
    if (int rc = xb_edge_find(c, &edges)) return rc;
    if (edges == 0) return XB_OK;  // thread_handlers.py:151-153
    int64_t k = 0;
    auto put = [&](int64_t e, int64_t ch) {
        if (log && 2 * k + 1 < log_capacity) { log[2 * k] = e; log[2 * k + 1] = ch; }
        k++;
        if (n_iters) *n_iters = k;
    };
    if (int rc = xb_refine_trace(c, &changed, &esc)) return rc;
    if (esc) return fail(XB_E_STATE, "xb_refine: %lld traces left the valid slab", (long long)esc);
    put(edges, changed);
    for (int64_t it = 2; iters < 0 || it <= iters; it++) {  // thread_handlers.py:194-236
        if (mode == XB_REFINE_ALL) {
            if (int rc = xb_edge_find(c, &edges)) return rc;
        } else {
            if (int rc = xb_edge_check(c, &checked, &edges)) return rc;
        }
        if (int rc = xb_refine_trace(c, &changed, &esc)) return rc;
        if (esc) return fail(XB_E_STATE, "xb_refine: %lld traces left the valid slab", (long long)esc);
        put(edges, changed);
        if (changed == 0) break;
    }
    return XB_OK;
}

int xb_charge_sum(xb_ctx *c, double voxel_volume, int64_t n_labels, double *charge, double *volume) {
    NEED_GRID("xb_charge_sum");
    if (n_labels <= 0) return XB_OK;
    const Grid &g = c->g;
    const long long own = (long long)(g.x1 - g.x0) * g.nyz;
    double *dch = nullptr;
    unsigned long long *dcn = nullptr;
    HIPCHK(hipMalloc(&dch, n_labels * sizeof(double)));
    HIPCHK(hipMalloc(&dcn, n_labels * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(dch, 0, n_labels * sizeof(double), c->stream));
    HIPCHK(hipMemsetAsync(dcn, 0, n_labels * sizeof(unsigned long long), c->stream));
    if (n_labels <= CS_BINS) {
        const int per_thread = 16;
        k_charge_sum_lds<<<nblocks((own + per_thread - 1) / per_thread), TPB, 0, c->stream>>>(g, c->rho, c->labels, (int)n_labels, dch, dcn, per_thread);
    } else {
        k_charge_sum_glb<<<nblocks(own), TPB, 0, c->stream>>>(g, c->rho, c->labels, (int)n_labels, dch, dcn);
    }
    hipError_t e = hipGetLastError();
    std::vector<unsigned long long> cn(n_labels);
    if (e == hipSuccess) e = hipMemcpyAsync(charge, dch, n_labels * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(cn.data(), dcn, n_labels * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(dch);
    hipFree(dcn);
    if (e != hipSuccess) return fail(XB_E_HIP, "xb_charge_sum: %s", hipGetErrorString(e));
    for (int64_t i = 0; i < n_labels; i++) {
        charge[i] *= voxel_volume;  // utils.py:251-252
        volume[i] = (double)cn[i] * voxel_volume;
    }
    return XB_OK;
}

int xb_volume_assign(xb_ctx *c, const int64_t *swap, int64_t n_swap) {
    NEED_GRID("xb_volume_assign");
    c->buni_valid = false;
    if (n_swap <= 0) return XB_OK;
    if (n_swap > c->max_cap) return fail(XB_E_LIMIT, "xb_volume_assign: swap table too long");
    std::vector<int> s(n_swap);
    for (int64_t i = 0; i < n_swap; i++) s[i] = (int)swap[i];
    const Grid &g = c->g;
    const long long own = (long long)(g.x1 - g.x0) * g.nyz;
    HIPCHK(hipMemcpyAsync(c->max_aux, s.data(), n_swap * sizeof(int), hipMemcpyHostToDevice, c->stream));
    k_volume_assign<<<nblocks(own), TPB, 0, c->stream>>>(g, c->labels, c->max_aux, (int)n_swap);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}

// utils.atom_assign (utils.py:185-232), host side: N_maxima x N_atoms x 27 -- tiny.
int xb_atom_assign(const double *b_max, int64_t n_max, const double *atoms, int64_t n_atoms, const double lattice[9],
                   int64_t *atom_out, double *dist_out) {
    if (n_atoms <= 0) return fail(XB_E_ARG, "xb_atom_assign: no atoms");
    double pbc[3] = {0., 0., 0.};  // utils.py:199: persists across maxima (206-208 read it before the loops)
    for (int64_t i = 0; i < n_max; i++) {
        const double *b = b_max + 3 * i;
        double e0 = b[0] - (atoms[0] + pbc[0]), e1 = b[1] - (atoms[1] + pbc[1]), e2 = b[2] - (atoms[2] + pbc[2]);
        double best = (e0 * e0 + e1 * e1) + e2 * e2;
        int64_t who = 0;
        for (int64_t j = 0; j < n_atoms; j++) {
            const double *a = atoms + 3 * j;
            for (int x = -1; x < 2; x++)
                for (int y = -1; y < 2; y++)
                    for (int z = -1; z < 2; z++) {
                        for (int k = 0; k < 3; k++) pbc[k] = (lattice[k] * x + lattice[3 + k] * y) + lattice[6 + k] * z;
                        e0 = b[0] - (a[0] + pbc[0]); e1 = b[1] - (a[1] + pbc[1]); e2 = b[2] - (a[2] + pbc[2]);
                        const double d = (e0 * e0 + e1 * e1) + e2 * e2;
                        if (d < best) { best = d; who = j; }
                    }
        }
        atom_out[i] = who;
        dist_out[i] = std::sqrt(best);
    }
    return XB_OK;
}

// thread_handlers.surface_distance (thread_handlers.py:239-297) on the resident atom map: edge_find
// on a fresh `known`, then the per-atom minimum squared distance of the edge voxels (+inf: no edge).
int xb_surface_distance(xb_ctx *c, const double lattice[9], const double *atoms_cart, int64_t n_atoms,
                        double *min_d2, int64_t *edges_out) {
    NEED_GRID("xb_surface_distance");
    if (n_atoms <= 0 || n_atoms > 100000) return fail(XB_E_ARG, "xb_surface_distance: bad atom count");
    int64_t edges = 0;
    if (int rc = xb_edge_find(c, &edges)) return rc;
    if (edges_out) *edges_out = edges;
    std::vector<unsigned long long> init(n_atoms, 0x7FF0000000000000ULL);  // +inf
    double *dbuf = (double *)c->stage;  // lattice (9), atoms (3n), minima (n as u64)
    unsigned long long *dmin = (unsigned long long *)(dbuf + 16 + 3 * n_atoms);
    HIPCHK(hipMemcpyAsync(dbuf, lattice, 9 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dbuf + 16, atoms_cart, 3 * n_atoms * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(dmin, init.data(), n_atoms * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
    if (edges)
        k_surface_dist<<<nblocks(edges), TPB, 0, c->stream>>>(light(c->g), c->labels, c->list, (int)edges, dbuf, dbuf + 16,
                                                             (int)n_atoms, dmin);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(min_d2, dmin, n_atoms * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}

int xb_volume_mask(xb_ctx *c, int64_t vol_num, double *out_host) {
    NEED_GRID("xb_volume_mask");
    double *tmp = (double *)c->stage;  // N*8 bytes
    k_volume_mask<<<nblocks(c->N), TPB, 0, c->stream>>>(c->rho, c->labels, (int)vol_num, tmp, c->N);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_host, tmp, c->N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}

int xb_label_sum(xb_ctx *c, int64_t value, double *sum, int64_t *count) {
    NEED_GRID("xb_label_sum");
    const Grid &g = c->g;
    HIPCHK(hipMemsetAsync(c->dsum, 0, sizeof(double), c->stream));
    HIPCHK(hipMemsetAsync(c->counters64, 0, sizeof(unsigned long long), c->stream));
    k_label_sum<<<nblocks((long long)(g.x1 - g.x0) * g.nyz), TPB, 0, c->stream>>>(g, c->rho, c->labels, (int)value, c->dsum, c->counters64);
    HIPCHK(hipGetLastError());
    double s;
    unsigned long long n;
    HIPCHK(hipMemcpyAsync(&s, c->dsum, sizeof s, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&n, c->counters64, sizeof n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (sum) *sum = s;
    if (count) *count = (int64_t)n;
    return XB_OK;
}

int xb_set_table_window(xb_ctx *c, int64_t margin) {
    NEED_GRID("xb_set_table_window");
    Grid &g = c->g;
    c->grad_valid = false;
    c->table_stage = 0;
    const int own = g.x1 - g.x0;
    if (margin < 0 || own == g.nx) { g.wx0 = 0; g.wlen = g.nx; c->table_margin = -1; return XB_OK; }
    if (g.nx % 8 || g.ny % 8 || g.nz % 8 || g.x0 % 8 || g.x1 % 8)
        return fail(XB_E_ARG, "xb_set_table_window: grid and slab must be made of whole 8^3 bricks");
    const int m8 = (int)((std::max<int64_t>(margin, c->halo) + 7) / 8) * 8;
    if (own + 2 * m8 >= g.nx) { g.wx0 = 0; g.wlen = g.nx; c->table_margin = -1; return XB_OK; }
    g.wx0 = ((g.x0 - m8) % g.nx + g.nx) % g.nx;
    g.wlen = own + 2 * m8;
    c->table_margin = m8;
    return XB_OK;
}
int xb_table_build(xb_ctx *c, int64_t *n_local_seeds) {
    NEED_GRID("xb_table_build");
    if (int rc = ensure_grad(c, true, true)) return rc;
    if (n_local_seeds) *n_local_seeds = table_windowed(c) ? (int64_t)c->window_seeds.size() : 0;
    return XB_OK;
}
int xb_table_local_seeds(xb_ctx *c, int64_t *out, int64_t capacity) {
    NEED_GRID("xb_table_local_seeds");
    if ((int64_t)c->window_seeds.size() > capacity) return fail(XB_E_ARG, "xb_table_local_seeds: capacity too small");
    for (size_t i = 0; i < c->window_seeds.size(); i++) out[i] = c->window_seeds[i];
    return XB_OK;
}
int xb_brick_masks(xb_ctx *c, void **dev_ptr, int64_t *n_bricks, int64_t *own_first, int64_t *own_count) {
    NEED_GRID("xb_brick_masks");
    const Grid &g = c->g;
    if (g.nx % 8 || g.ny % 8 || g.nz % 8) return fail(XB_E_STATE, "xb_brick_masks: grid is not made of whole bricks");
    const int64_t nbr = c->N / 512, per_plane = (int64_t)(g.ny / 8) * (g.nz / 8);
    if (dev_ptr) *dev_ptr = (void *)(c->list + nbr);
    if (n_bricks) *n_bricks = nbr;
    if (own_first) *own_first = (g.x0 / 8) * per_plane;
    if (own_count) *own_count = ((g.x1 - g.x0) / 8) * per_plane;
    return XB_OK;
}
int xb_table_finish(xb_ctx *c, const int64_t *seeds, int64_t n_seeds) {
    NEED_GRID("xb_table_finish");
    if (c->table_stage < 1 || !c->grad_valid) return fail(XB_E_STATE, "xb_table_finish: call xb_table_build first");
    int rc = XB_OK;
    if (n_seeds >= 1 && n_seeds <= XB_BOX_SEEDS_MAX) {
        std::vector<int> sv(n_seeds);
        for (int64_t i = 0; i < n_seeds; i++) sv[i] = (int)seeds[i];
        ScopedTimer t(c, 4);
        rc = table_regions(c, sv, true);
    }
    c->table_stage = 2;
    c->table_prebuilt = true;
    return rc;
}

void *xb_labels_ptr(xb_ctx *c) { return c ? (void *)c->labels : nullptr; }
void *xb_known_ptr(xb_ctx *c) { return c ? (void *)c->known : nullptr; }
void *xb_density_ptr(xb_ctx *c) { return c ? (void *)c->rho : nullptr; }
int64_t xb_plane_elems(xb_ctx *c) { return c ? c->g.nyz : 0; }

int xb_copy_planes(xb_ctx *c, int which, int to_device, void *host, int64_t xa, int64_t xb) {
    NEED_GRID("xb_copy_planes");
    if (xa < 0 || xb > c->g.nx || xa > xb) return fail(XB_E_ARG, "xb_copy_planes: bad plane range");
    const size_t es = which == 0 ? 4 : 1;
    char *dev = which == 0 ? (char *)c->labels : (char *)c->known;
    const size_t off = (size_t)xa * c->g.nyz * es, bytes = (size_t)(xb - xa) * c->g.nyz * es;
    if (to_device) { c->list_valid = false; c->has_vacuum = true; c->buni_valid = false; }
    if (to_device) HIPCHK(hipMemcpyAsync(dev + off, host, bytes, hipMemcpyHostToDevice, c->stream));
    else HIPCHK(hipMemcpyAsync(host, dev + off, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}

int xb_set_option(xb_ctx *c, int key, int value) {
    if (!c) return fail(XB_E_ARG, "null ctx");
    if (key == 0) c->opt_trace = value;
    else if (key == 1) { c->opt_boxes = value & 1; c->opt_bricks = (value >> 1) & 1; }
    else if (key == 3) c->opt_dbg = value;
    else if (key == 2 && (value == 64 || value == 128 || value == 256)) c->opt_trace_tpb = value;
    else return fail(XB_E_ARG, "xb_set_option: unknown key %d", key);
    return XB_OK;
}
int xb_slow_path_stats(xb_ctx *c, int64_t *assign_total, int64_t *refine_total) {
    if (!c) return fail(XB_E_ARG, "null ctx");
    if (assign_total) *assign_total = c->stat_ovf_assign;
    if (refine_total) *refine_total = c->stat_ovf_refine;
    return XB_OK;
}
int xb_box_stats(xb_ctx *c, int64_t *n_boxes, int64_t *box_voxels) {
    if (!c) return fail(XB_E_ARG, "null ctx");
    if (n_boxes) *n_boxes = c->n_boxes;
    if (box_voxels) *box_voxels = c->box_voxels;
    return XB_OK;
}
int xb_enable_timing(xb_ctx *c, int on) {
    if (!c) return fail(XB_E_ARG, "null ctx");
    c->timing = on != 0;
    return XB_OK;
}
int xb_kernel_time_reset(xb_ctx *c) {
    if (!c) return fail(XB_E_ARG, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (auto &t : c->tk) {
        for (auto &p : t.pending) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
        t.pending.clear();
        t.ms = 0.;
        t.launches = 0;
    }
    return XB_OK;
}
int xb_kernel_time(xb_ctx *c, int which, double *ms_total, int64_t *launches) {
    if (!c || which < 0 || which > 5) return fail(XB_E_ARG, "xb_kernel_time: bad argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    TimedKernel &t = c->tk[which];
    for (auto &p : t.pending) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, p.first, p.second));
        t.ms += ms;
        t.launches++;
        hipEventDestroy(p.first);
        hipEventDestroy(p.second);
    }
    t.pending.clear();
    if (ms_total) *ms_total = t.ms;
    if (launches) *launches = t.launches;
    return XB_OK;
}

}  // extern "C"
