// k_edges.h -- device kernels of libbader_hip.so: refinement: edge_find, dilation, compaction, retrace, edge_check.
// Included by bader_hip.hip (one translation unit); see bader_kernels.h for the common device code.
#pragma once

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
    // unconditional loads (a vacuum neighbour reads rho[v] instead): with a branch per neighbour the compiler
    // waits for each load before it issues the next, 27 serial round trips
    double nr[27];
#pragma unroll
    for (k = 0; k < 27; k++) nr[k] = rho[nb[k] >= 0 ? nb[k] : v];
#pragma unroll
    for (k = 0; k < 27; k++) is_max &= !(nr[k] > max_val);
}

// buni[K] = the label shared by all 512 voxels of brick K, or INT_MIN when the brick is mixed.
// Lets the edge sweep skip tiles whose whole 3x3x3 surroundings carry one label (no edge possible).
#define XB_MIXED (-2147483647 - 1)
// After an assignment without vacuum every certain brick is uniform by construction (all its voxels
// carry the rank of the region's maximum): only the bricks of the walk list need the label scan.
// The bricks' label uniformity behind the relabel of an assignment: the regions' bricks (uniform by construction) and -- when
// the walkers left their verdicts (bres) -- the walk-list bricks.  Element-wise, disjoint bricks.
__global__ __launch_bounds__(256) void k_buni_after_relabel(int nbr, const int *__restrict__ blab, const int *__restrict__ box_max,
                                                            const int *__restrict__ rank, int *__restrict__ buni, const int *gate,
                                                            const int *__restrict__ walk, const int *n_walk, const int *__restrict__ bres) {
    if (gate && !*gate) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (int b = t; b < nbr; b += nt) {
        const int l = blab[b];
        if (l > 0) buni[b] = rank[box_max[l - 1]];
    }
    if (!bres) return;   // (no verdicts from the walkers: the caller scans the walk-list bricks' labels, k_label_uniform_list)
    const int n = *n_walk;
    for (int e = t; e < n; e += nt) {
        const int b = walk[e], r = bres[b];
        buni[b] = r == XB_MIXED ? XB_MIXED : rank[r];
    }
}

// buni3[K] = the label when brick K AND its 26 neighbour bricks all carry that one label, else XB_MIXED: a tile
// of the edge sweep then needs a handful of lookups instead of one per brick of its surroundings
__global__ void k_buni3(int nb0, int nb1, int nb2, const int *__restrict__ buni, int *__restrict__ buni3) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb0 * nb1 * nb2) return;
    const int b0 = b / (nb1 * nb2), b1 = (b / nb2) % nb1, b2 = b % nb2;
    const int l = buni[b];
    bool ok = l != XB_MIXED;
#pragma unroll
    for (int k = 0; k < 27; k++) {
        const int q = buni[(wrap_any(b0 + k / 9 - 1, nb0) * nb1 + wrap_any(b1 + (k / 3) % 3 - 1, nb1)) * nb2 + wrap_any(b2 + k % 3 - 1, nb2)];
        ok &= (q == l);
    }
    buni3[b] = ok ? l : XB_MIXED;
}
// one WAVE per brick (8 labels per lane), bricks strided over the grid.  The bricks: the entries of `walk` (their number may live on
// the device, n_dev), or -- walk == nullptr -- the n_walk bricks b_off, b_off + 1, ... modulo nbr (a slab scans the bricks of its
// planes only).  `gate` (when given) must be non-zero for the kernel to do anything.
__global__ __launch_bounds__(TPB) void k_label_uniform_list(GridL g, const int *__restrict__ labels, int nb1, int nb2,
                                                            const int *__restrict__ walk, int n_walk, const int *n_dev,
                                                            const int *gate, int *__restrict__ buni, int b_off = 0, int nbr = 0) {
    if (gate && !*gate) return;
    const int n = n_dev ? *n_dev : n_walk;
    const int lane = threadIdx.x % XB_WAVE;
    for (int e = blockIdx.x * (TPB / XB_WAVE) + threadIdx.x / XB_WAVE; e < n; e += gridDim.x * (TPB / XB_WAVE)) {
        int b;
        if (walk) b = walk[e];
        else { b = e + b_off; if (b >= nbr) b -= nbr; }
        const int b0 = b / (nb1 * nb2), b1 = (b / nb2) % nb1, b2 = b % nb2;
        int lo = 2147483647, hi = XB_MIXED;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int t = lane + k * XB_WAVE;
            const int x = b0 * 8 + t / 64, y = b1 * 8 + (t / 8) % 8, z = b2 * 8 + t % 8;
            if (x < g.nx && y < g.ny && z < g.nz) {   // (a brick the grid cuts: its voxels inside the grid)
                const int l = labels[(x * g.ny + y) * g.nz + z];
                lo = min(lo, l); hi = max(hi, l);
            }
        }
        for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
        if (lane == 0) buni[b] = (lo == hi) ? lo : XB_MIXED;
    }
}

// refinement.py:385-404 as written there: every listed edge voxel turns the known >= 0 voxels of
// its 27-box into -1 (all -2 flags are final before this kernel starts).
__global__ __launch_bounds__(TPB) void k_edge_dilate_list(GridL g, int8_t *known, const int *__restrict__ list, int n_host,
                                                          const int *n_dev) {
    const int n = n_dev ? *n_dev : n_host;   // the list length may live on the device: the grid strides over it
  for (int t = blockIdx.x * TPB + threadIdx.x; t < n; t += gridDim.x * TPB) {
    const int v = list[t];
    const int x = v / g.nyz;
    const int r = v - x * g.nyz;
    const int y = r / g.nz, z = r - y * g.nz;
    // all loads first (with a load-test-store per neighbour the compiler serialises 27 round trips); away from the z faces
    // the three flags of a row come with ONE unaligned 4-byte load (round 4: 9 loads instead of 27)
    int row[9];
#pragma unroll
    for (int j = 0; j < 9; j++) row[j] = (wrapi(x + j / 3 - 1, g.nx) * g.ny + wrapi(y + j % 3 - 1, g.ny)) * g.nz;
    if (z >= 1 && z + 2 < g.nz) {
        unsigned int w[9];
#pragma unroll
        for (int j = 0; j < 9; j++) w[j] = *reinterpret_cast<const unsigned int *>(known + row[j] + z - 1);
#pragma unroll
        for (int j = 0; j < 27; j++)
            if (!((w[j / 3] >> (8 * (j % 3) + 7)) & 1u)) known[row[j / 3] + z - 1 + j % 3] = -1;
    } else {
        int8_t k[27];
#pragma unroll
        for (int j = 0; j < 27; j++) k[j] = known[row[j / 3] + wrapi(z + j % 3 - 1, g.nz)];
#pragma unroll
        for (int j = 0; j < 27; j++)
            if (k[j] >= 0) known[row[j / 3] + wrapi(z + j % 3 - 1, g.nz)] = -1;
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
// one tile at (tx0 planes from xa, y0, z0); `buni` null: no uniformity shortcut (the caller knows the tile is mixed)
__device__ __forceinline__ void edge_tile(const GridL &g, const double *__restrict__ rho, const int *__restrict__ labels,
                                          int8_t *__restrict__ known, int xa, int nplanes, int *__restrict__ list,
                                          int *list_count, int small, const int *__restrict__ buni,
                                          const GradRec *__restrict__ G, const unsigned char *__restrict__ brick_rec,
                                          int no_vacuum, int tx0, int y0, int z0, int salt = 0) {
    // (salt: always 0, but opaque to the compiler when the caller loops over tiles -- everything derived from the thread index
    // then counts as loop variant, and the ~60 LDS row addresses are not carried in registers across iterations)
    const int tid = (int)threadIdx.x + salt;
    __shared__ int tile[ET_X + 2][ET_Y + 2][ET_Z + 2];
    if (buni) {
        // `buni` here is buni3 (k_buni3): a brick entry says that the brick and all its 26 neighbours carry one
        // label, so the bricks the tile itself lies in (1-2 in x, one in y, ET_Z/8 in z) settle the tile plus
        // its one-voxel halo: no voxel of it has a foreign neighbour
        __shared__ int s_lab, s_mixed;
        if (tid == 0) { s_lab = XB_MIXED; s_mixed = 0; }
        __syncthreads();
        const int nb1 = (g.ny + 7) >> 3, nb2 = (g.nz + 7) >> 3;
        int X0 = xa + tx0, X1 = xa + min(tx0 + ET_X, nplanes) - 1;
        if (X0 >= g.nx) X0 -= g.nx;
        if (X1 >= g.nx) X1 -= g.nx;
        const int bxa = X0 >> 3, bxb = X1 >> 3, nbz = min(ET_Z / 8, nb2 - (z0 >> 3));
        if ((int)tid < 2 * nbz) {
            const int bx = (tid & 1) ? bxb : bxa;
            const int l = buni[(bx * nb1 + (y0 >> 3)) * nb2 + (z0 >> 3) + (tid >> 1)];
            if (l == XB_MIXED) s_mixed = 1;
            else {
                const int old = atomicCAS(&s_lab, XB_MIXED, l);
                if (old != XB_MIXED && old != l) s_mixed = 1;
            }
        }
        __syncthreads();
        if (!s_mixed) {
            const int8_t o = (s_lab == -1) ? 0 : 2;  // vacuum stays 0 (refinement.py:342-343), else "known"
            if ((g.nz & 15) == 0 && z0 + ET_Z <= g.nz) {   // 16-byte stores: 32 rows of 64 flags, 4 threads per row
                if (tid < 128) {
                    const int row = tid >> 2, seg = tid & 3;
                    const int xr = tx0 + (row >> 3), y = y0 + (row & 7);
                    if (xr < nplanes && y < g.ny) {
                        int x = xa + xr;
                        if (x >= g.nx) x -= g.nx;
                        const unsigned w = o ? 0x02020202u : 0u;
                        *reinterpret_cast<uint4 *>(known + ((size_t)(x * g.ny + y) * g.nz + z0 + seg * 16)) = make_uint4(w, w, w, w);
                    }
                }
                return;
            }
            const int tz = tid & 63, tyb = tid >> 6;
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
    // the brick bytes (records exist / holds a maximum) of the bricks under the tile's voxels, per x-plane of the tile and
    // 8-voxel run of z: pass 2 asks them per edge candidate (round 3: a dependent global byte load in front of every candidate)
    __shared__ unsigned char s_binfo[ET_X][ET_Z / 8];
    {
        // Row-wise staging, every load of a wave in flight before the first wait.  Round 4 (as in k_brick_masks): a row's address
        // is one scalar addition -- wave w stages rows y0-1+ey of all six x-planes for its own two or three ey (3, 3, 2, 2),
        // plane bases and row offsets are scalars computed once, the z wrap once per lane.  Round 3 derived (x, y) of each of a
        // wave's 15 rows on the scalar unit: a division, three wraps, two products and the `small` branch per row.
        const int wv = __builtin_amdgcn_readfirstlane(tid / XB_WAVE), lane = tid % XB_WAVE;
        constexpr int NROW = (ET_X + 2) * (ET_Y + 2);
        static_assert(ET_Z == XB_WAVE && 2 * NROW <= TPB && TPB / XB_WAVE == 4 && ET_Y + 2 == 10, "tile shape");
        int Zc = z0 + lane;
        if (small) Zc %= g.nz;
        else Zc = wrap_u(Zc, g.nz);
        const int ey0 = wv < 2 ? 3 * wv : 2 * wv + 2, ney = wv < 2 ? 3 : 2;   // rows 0-2, 3-5, 6-7, 8-9
        int yoff[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            int Y = y0 + ey0 + j - 1;
            if (small) Y = ((Y % g.ny) + g.ny) % g.ny;
            else Y = wrap_u(Y, g.ny);
            yoff[j] = Y * g.nz;
        }
        int val[ET_X + 2][3];
#pragma unroll
        for (int ex = 0; ex < ET_X + 2; ex++) {
            int X = xa + tx0 + ex - 1;
            if (small) X = ((X % g.nx) + g.nx) % g.nx;
            else { X = wrap_u(X, g.nx); X = wrap_u(X, g.nx); }
            const int *plane = labels + (size_t)X * g.nyz;
#pragma unroll
            for (int j = 0; j < 3; j++) val[ex][j] = (j < ney) ? plane[yoff[j] + Zc] : 0;
        }
        int hv = 0;
        const int hr = tid >> 1, hside = tid & 1;
        const int hex = hr / (ET_Y + 2), hey = hr - hex * (ET_Y + 2);
        if (tid < 2 * NROW) {
            int X = xa + tx0 + hex - 1, Y = y0 + hey - 1, Z = hside ? z0 + ET_Z : z0 - 1;
            if (small) {
                X = ((X % g.nx) + g.nx) % g.nx; Y = ((Y % g.ny) + g.ny) % g.ny; Z = ((Z % g.nz) + g.nz) % g.nz;
            } else {
                X = wrap_u(X, g.nx); X = wrap_u(X, g.nx); Y = wrap_u(Y, g.ny); Z = wrap_u(Z, g.nz);
            }
            hv = labels[(X * g.ny + Y) * g.nz + Z];
        }
        if (tid >= TPB - ET_X * (ET_Z / 8)) {   // the last 32 threads: one brick byte each
            const int i = tid - (TPB - ET_X * (ET_Z / 8)), tx = i / (ET_Z / 8), zb = i % (ET_Z / 8);
            int x = xa + tx0 + tx;
            if (x >= g.nx) x -= g.nx;
            const int nb1 = (g.ny + 7) >> 3, nb2 = (g.nz + 7) >> 3, bz = (z0 >> 3) + zb;
            s_binfo[tx][zb] = (brick_rec && bz < nb2 && y0 < g.ny) ? brick_rec[((x >> 3) * nb1 + (y0 >> 3)) * nb2 + bz] : (unsigned char)1;
        }
#pragma unroll
        for (int ex = 0; ex < ET_X + 2; ex++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (j < ney) tile[ex][ey0 + j][lane + 1] = val[ex][j];
        if (tid < 2 * NROW) tile[hex][hey][hside ? ET_Z + 1 : 0] = hv;
    }
    __syncthreads();
    const int tz = tid & 63, tyb = tid >> 6;
    // "some non-vacuum neighbour carries another label" (refinement.py:357-372) is separable: with vacuum (-1) read as
    // the largest unsigned value, it holds iff the unsigned minimum of the 27 labels is below the voxel's own label or
    // their signed maximum above it.  Minima / maxima along z per (x, y) row, then over y for the thread's two y
    // positions, then over x per voxel: 108 LDS reads and ~40 operations per voxel instead of 216 and ~110.
    unsigned ymin[ET_X + 2][2];
    int ymax[ET_X + 2][2];
#pragma unroll
    for (int ex = 0; ex < ET_X + 2; ex++) {
        unsigned rmin[6];
        int rmax[6];
#pragma unroll
        for (int r = 0; r < 6; r++) {
            const int ey = tyb + r + (r >= 3 ? 1 : 0);   // rows tyb..tyb+2 (y = tyb) and tyb+4..tyb+6 (y = tyb + 4)
            const int a = tile[ex][ey][tz], b = tile[ex][ey][tz + 1], c2 = tile[ex][ey][tz + 2];
            rmin[r] = min(min((unsigned)a, (unsigned)b), (unsigned)c2);
            rmax[r] = max(max(a, b), c2);
        }
        ymin[ex][0] = min(min(rmin[0], rmin[1]), rmin[2]); ymax[ex][0] = max(max(rmax[0], rmax[1]), rmax[2]);
        ymin[ex][1] = min(min(rmin[3], rmin[4]), rmin[5]); ymax[ex][1] = max(max(rmax[3], rmax[4]), rmax[5]);
        __builtin_amdgcn_sched_barrier(0);   // keep the 18 LDS reads of one x-plane together: hoisted, all 108 cost a register each
    }
    // pass 1 (unrolled, registers only): which of the thread's 8 voxels have a foreign neighbour
    unsigned cand = 0, inside = 0, vac = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int tx = k >> 1, ty = tyb + ((k & 1) << 2);
        if (tx0 + tx < nplanes && y0 + ty < g.ny && z0 + tz < g.nz) {
            inside |= 1u << k;
            const int lab = tile[tx + 1][ty + 1][tz + 1];
            if (lab == -1) vac |= 1u << k;   // vacuum voxels are not classified (refinement.py:342-343)
            else {
                const unsigned bmin = min(min(ymin[tx][k & 1], ymin[tx + 1][k & 1]), ymin[tx + 2][k & 1]);
                const int bmax = max(max(ymax[tx][k & 1], ymax[tx + 1][k & 1]), ymax[tx + 2][k & 1]);
                if (bmin < (unsigned)lab || bmax > lab) cand |= 1u << k;
            }
        }
    }
    // pass 2: refinement.py:374-383, an edge unless it is a 26-neighbour maximum.  Round 4: the brick byte comes from LDS, and
    // the table keys the candidates need are gathered TOGETHER (one 8-byte load per candidate, all in flight) -- round 3 walked
    // the candidates one by one, a brick byte and then a record per candidate: up to eight times two dependent loads per
    // thread, the reason the sweep of a listed tile took ~25 us.
    unsigned edges = 0, full = 0;
#pragma unroll 1
    for (int h = 0; h < 2; h++) {   // the thread's voxels four at a time (the keys of eight at once cost the kernel its occupancy)
        if (!((cand >> (4 * h)) & 15u)) continue;
        unsigned pend = 0;
        double key[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int k = 4 * h + q;
            key[q] = 0.;
            if ((cand >> k) & 1u) {
                const int tx = 2 * h + (q >> 1), ty = tyb + ((q & 1) << 2);
                const int binfo = s_binfo[tx][tz >> 3];
                if (brick_rec && no_vacuum && !(binfo & 2)) {
                    // no voxel of this brick is a 26-neighbour maximum: each has a strictly denser neighbour
                    // (weighted > rho(v) implies rho(n) > rho(v)), and without vacuum no neighbour is skipped
                    edges |= 1u << k;
                } else {
                    int x = xa + tx0 + tx;
                    if (x >= g.nx) x -= g.nx;
                    if (G && plane_in_window(g, x) && (binfo & 1)) {  // (slabs: a window of planes; sparse table: flagged bricks)
                        key[q] = G[rec_slot(g, (x * g.ny + y0 + ty) * g.nz + z0 + tz)].key;
                        pend |= 1u << q;
                    } else
                        full |= 1u << k;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if ((pend >> q) & 1u) {
                // the table knows the best distance-weighted neighbour of v; if there is one (and it is not vacuum) that
                // neighbour is denser than v: not a maximum.  (weighted > rho(v) implies rho(n) > rho(v); the converse can
                // fail by rounding, so "no such neighbour" still takes the full test)
                const int k = 4 * h + q, tx = 2 * h + (q >> 1), ty = tyb + ((q & 1) << 2);
                const int og = key_og(key[q]);
                if (og != XB_OG_SELF && tile[tx + og / 9][ty + (og / 3) % 3][tz + og % 3] != -1) edges |= 1u << k;
                else full |= 1u << k;
            }
    }
#pragma unroll 1
    for (unsigned m = full; m; m &= m - 1) {   // rare: the full 27-point density test
        const int k = __ffs(m) - 1;
        const int tx = k >> 1, ty = tyb + ((k & 1) << 2);
        int x = xa + tx0 + tx;
        if (x >= g.nx) x -= g.nx;
        const int y = y0 + ty, z = z0 + tz;
        const int v = (x * g.ny + y) * g.nz + z;
        bool is_max = true;
        const double c = rho[v];
        for (int dx = -1; dx < 2; dx++) {
            const int X = wrapi(x + dx, g.nx);
            for (int dy = -1; dy < 2; dy++) {
                const int Y = wrapi(y + dy, g.ny);
                for (int dz = -1; dz < 2; dz++) {
                    const int Z = wrapi(z + dz, g.nz);
                    if (tile[tx + 1 + dx][ty + 1 + dy][tz + 1 + dz] != -1 && rho[(X * g.ny + Y) * g.nz + Z] > c) is_max = false;
                }
            }
        }
        if (!is_max) edges |= 1u << k;
    }
    // pass 3: the flags, and the owned edge voxels for the list
    int vidx[8];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        vidx[k] = -1;
        if ((inside >> k) & 1u) {
            const int tx = k >> 1, ty = tyb + ((k & 1) << 2);
            int x = xa + tx0 + tx;
            if (x >= g.nx) x -= g.nx;
            const int v = (x * g.ny + y0 + ty) * g.nz + z0 + tz;
            const bool e = (edges >> k) & 1u;
            known[v] = ((vac >> k) & 1u) ? 0 : (e ? -2 : 2);
            if (e && x >= g.x0 && x < g.x1) { vidx[k] = v; cnt++; }
        }
    }
    int total;
    const int off = block_scan_excl(cnt, total);
    __shared__ int base_s;
    if (tid == 0) base_s = total ? atomicAdd(list_count, total) : 0;
    __syncthreads();
    int w = base_s + off;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (vidx[k] >= 0) list[w++] = vidx[k];
}

// The dilation (refinement.py:385-404) tile by tile, for the sweep over listed tiles: a workgroup stages the flags of its
// tile + a one-voxel periodic halo in LDS (4 KB) and turns every known >= 0 voxel of the tile with an edge voxel (-2) in its
// 27-box into -1 -- the same flags as k_edge_dilate_list leaves (all -2 are final before this kernel starts; it only turns
// 0 / 2 into -1, so reading a neighbour tile's flags while that tile is being written is harmless).  Only listed tiles
// can hold such a voxel: an unlisted tile is of one non-vacuum label with all the 26 bricks around its own, so no voxel
// within two voxels of it has a foreign neighbour.  4 KB in + 2 KB out per tile, rows of 64 bytes, instead of 27 scattered
// byte loads per edge voxel (0.14 -> 0.04 ms at 512^3).
__device__ __forceinline__ unsigned long long bytes_equal_fe(unsigned long long w) {   // 0x80 in every byte of w that is 0xFE (-2), exactly
    const unsigned long long x = w ^ 0xFEFEFEFEFEFEFEFEull, m = 0x7F7F7F7F7F7F7F7Full;
    return ~(((x & m) + m) | x | m);
}
// (nz a multiple of ET_Z: a thread owns 8 z-consecutive voxels of one row -- 16-byte global loads, 8-byte LDS reads and one
// 8-byte store)
__global__ __launch_bounds__(TPB) void k_edge_dilate_tiles(GridL g, int8_t *known, const int *__restrict__ tiles, const int *n_tiles) {
    constexpr int ROW = ET_Z + 8;   // a row of flags in LDS: the tile's 64 bytes at offset 4, the halo bytes at 3 and 68
    __shared__ __attribute__((aligned(16))) int8_t s[(ET_X + 2) * (ET_Y + 2) * ROW];
    static_assert(ET_X * ET_Y * (ET_Z / 8) == TPB && (ET_X + 2) * (ET_Y + 2) * 4 <= TPB, "one 8-voxel chunk per thread");
    const int ntz = g.nz / ET_Z, nty = (g.ny + ET_Y - 1) / ET_Y, n = *n_tiles;
    const int n_xcd = (gridDim.x % 8 == 0) ? 8 : 1;   // (an eighth of the list per XCD, as in k_edge_flag_listed)
    const int part = (n + n_xcd - 1) / n_xcd, part0 = (int)(blockIdx.x % n_xcd) * part, part1 = min(n, part0 + part);
  for (int item = part0 + blockIdx.x / n_xcd; item < part1; item += gridDim.x / n_xcd) {   // (uniform per block)
    const int t = (int)((unsigned)tiles[item] & 0x7fffffffu);
    const int tx0 = (t / (ntz * nty)) * ET_X, y0 = ((t / ntz) % nty) * ET_Y, z0 = (t % ntz) * ET_Z;
    __syncthreads();   // the previous tile's readers are done
    if (threadIdx.x < (ET_X + 2) * (ET_Y + 2) * 4) {   // 60 rows x 4 x 16 bytes
        const int row = threadIdx.x >> 2, q = threadIdx.x & 3;
        const size_t base = (size_t)(wrapi(tx0 + row / (ET_Y + 2) - 1, g.nx) * g.ny + wrapi(y0 + row % (ET_Y + 2) - 1, g.ny)) * g.nz;
        const uint4 w = *reinterpret_cast<const uint4 *>(known + base + z0 + 16 * q);
        unsigned *d = reinterpret_cast<unsigned *>(s + row * ROW + 4 + 16 * q);   // (4-byte aligned: the reads below want offset 4 mod 8)
        d[0] = w.x; d[1] = w.y; d[2] = w.z; d[3] = w.w;
        if (q < 2) s[row * ROW + (q ? 4 + ET_Z : 3)] = known[base + wrapi(q ? z0 + ET_Z : z0 - 1, g.nz)];
    }
    __syncthreads();
    const int ex = threadIdx.x / (ET_Y * (ET_Z / 8)), ey = (threadIdx.x / (ET_Z / 8)) % ET_Y, ch = threadIdx.x % (ET_Z / 8);
    // bytes [8 ch, 8 ch + 16) of the nine rows around: the chunk's voxels sit at bytes 4 .. 11 of that window
    unsigned long long e0 = 0, e1 = 0, c0 = 0, c1 = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const unsigned long long *p = reinterpret_cast<const unsigned long long *>(s + ((ex + j / 3) * (ET_Y + 2) + ey + j % 3) * ROW + 8 * ch);
        const unsigned long long w0 = p[0], w1 = p[1];
        e0 |= bytes_equal_fe(w0); e1 |= bytes_equal_fe(w1);
        if (j == 4) { c0 = w0; c1 = w1; }
    }
    // near[p] = an edge voxel in column p - 1, p or p + 1 (bit 7 of byte p of the 16-byte window e1:e0)
    const unsigned long long n0 = e0 | (e0 << 8) | (e0 >> 8) | (e1 << 56), n1 = e1 | (e1 << 8) | (e1 >> 8) | (e0 >> 56);
    // the chunk: bytes 4..7 of word 0, 0..3 of word 1; a byte with a clear sign bit (known >= 0) and `near` becomes 0xFF
    unsigned long long old = (c0 >> 32) | (c1 << 32), near = (n0 >> 32) | (n1 << 32);
    const unsigned long long hit = near & ~old & 0x8080808080808080ull;
    if (hit && tx0 + ex < g.nx && y0 + ey < g.ny) {   // (a tile the grid cuts in x or y: its rows inside the grid)
        const unsigned long long fill = (hit >> 7) * 0xFFull;
        *reinterpret_cast<unsigned long long *>(known + ((size_t)(tx0 + ex) * g.ny + y0 + ey) * g.nz + z0 + 8 * ch) = old | fill;
    }
  }
}

__global__ __launch_bounds__(TPB) void k_edge_flag_tiled(GridL g, const double *__restrict__ rho,
                                                         const int *__restrict__ labels,
                                                         int8_t *__restrict__ known, int xa, int nplanes,
                                                         int *__restrict__ list, int *list_count, int small,
                                                         const int *__restrict__ buni,
                                                         const GradRec *__restrict__ G,
                                                         const unsigned char *__restrict__ brick_rec, int no_vacuum) {
    edge_tile(g, rho, labels, known, xa, nplanes, list, list_count, small, buni, G, brick_rec, no_vacuum, blockIdx.z * ET_X,
              blockIdx.y * ET_Y, blockIdx.x * ET_Z);
}
// One GPU, whole bricks: `known` is preset to 2 and only the tiles that are not of one non-vacuum label with their
// surroundings are swept -- a sixth of the 65 536 tiles at 512^3, whose early-exit workgroups cost a third of the sweep.
// k_edge_tile_list: one thread per tile, the buni3 test of edge_tile; entry = tile index, bit 31 set for a tile of
// uniform VACUUM (its flags are 0, refinement.py:342-343).
// (tx_lo, tx_n: the tile planes tx_lo, tx_lo + 1, ... modulo nx / ET_X -- a slab lists the tiles of its own planes and of
// its halo planes separately)
// (gate: the state block of an assignment whose outcome the host has not seen yet, xb_assign_refine -- no tile is listed unless
// it ended the usual way: numbered on the device, no tie voxel; the sweep and the retraces then find nothing to do)
__global__ __launch_bounds__(TPB) void k_edge_tile_list(GridL g, const int *__restrict__ buni3, int *tiles, int *n_tiles, int tx_lo,
                                                        int tx_n, const int *gate = nullptr) {
    if (gate && (gate[FS_SORT_OK] == 0 || gate[FS_TIES] != 0)) return;
    const int ntz = (g.nz + ET_Z - 1) / ET_Z, nty = (g.ny + ET_Y - 1) / ET_Y, ntx = (g.nx + ET_X - 1) / ET_X;
    const int i = blockIdx.x * TPB + threadIdx.x;
    bool hit = false;
    unsigned entry = 0;
    if (i < tx_n * nty * ntz) {
        int tx = tx_lo + i / (ntz * nty);
        if (tx >= ntx) tx -= ntx;
        const int tz = i % ntz, ty = (i / ntz) % nty;
        const int t = (tx * nty + ty) * ntz + tz;
        const int nb1 = (g.ny + 7) >> 3, nb2 = (g.nz + 7) >> 3;
        const int bxa = (tx * ET_X) >> 3, bxb = (tx * ET_X + ET_X - 1) >> 3, nbz = min(ET_Z / 8, nb2 - ((tz * ET_Z) >> 3));
        int lab = XB_MIXED;
        bool mixed = false;
        for (int k = 0; k < 2 * nbz; k++) {
            const int l = buni3[(((k & 1) ? bxb : bxa) * nb1 + ((ty * ET_Y) >> 3)) * nb2 + ((tz * ET_Z) >> 3) + (k >> 1)];
            if (l == XB_MIXED) mixed = true;
            else if (lab == XB_MIXED) lab = l;
            else if (lab != l) mixed = true;
        }
        hit = mixed || lab == -1;
        entry = (unsigned)t | ((!mixed && lab == -1) ? 0x80000000u : 0u);
    }
    int total;
    const int off = block_scan_excl(hit ? 1 : 0, total);
    __shared__ int base_s;
    if (threadIdx.x == 0) base_s = total ? atomicAdd(n_tiles, total) : 0;
    __syncthreads();
    if (hit) tiles[base_s + off] = (int)entry;
}
__global__ __launch_bounds__(TPB) void k_edge_flag_listed(GridL g, const double *__restrict__ rho, const int *__restrict__ labels,
                                                          int8_t *__restrict__ known, int *__restrict__ list, int *list_count,
                                                          int small, const GradRec *__restrict__ G,
                                                          const unsigned char *__restrict__ brick_rec, int no_vacuum,
                                                          const int *__restrict__ tiles, const int *n_tiles) {
    // Round 4: a fixed grid strides over the listed tiles.  (Round 3 launched one workgroup per tile of the GRID and let the
    // unlisted ones return: five sixths of 65 536 workgroups at 512^3, each holding a slot of a compute unit for the ~2 us
    // it takes to start, read the list length and leave -- half of the kernel's 0.27 ms.  A loop used to cost the kernel its
    // occupancy, the compiler kept edge_tile's ~60 LDS row addresses alive across iterations; the staging by planes has none.)
    const int n = *n_tiles;
    const int ntz = (g.nz + ET_Z - 1) / ET_Z, nty = (g.ny + ET_Y - 1) / ET_Y;
    // Round 5: workgroups go to the eight XCDs in turn, each XCD has an L2 of its own, and the tile list is in tile order (z fastest):
    // XCD k sweeps the k-th contiguous eighth of the list, so that the tiles whose halos overlap -- neighbours in z, y and x -- are
    // staged through ONE L2 at about the same time (round 4 dealt consecutive tiles to different XCDs: 83 % of the sweep's L2
    // requests missed, every halo line came from HBM once per XCD that touched it).
    const int n_xcd = (gridDim.x % 8 == 0) ? 8 : 1;
    const int part = (n + n_xcd - 1) / n_xcd, part0 = (int)(blockIdx.x % n_xcd) * part, part1 = min(n, part0 + part);
#pragma unroll 1
    for (int item = part0 + blockIdx.x / n_xcd; item < part1; item += gridDim.x / n_xcd) {
        const unsigned entry = (unsigned)tiles[item];
        const int t = (int)(entry & 0x7fffffffu);
        const int tx0 = (t / (ntz * nty)) * ET_X, y0 = ((t / ntz) % nty) * ET_Y, z0 = (t % ntz) * ET_Z;
        if (entry & 0x80000000u) {   // uniform vacuum: flags 0
            for (int i = threadIdx.x; i < ET_X * ET_Y * ET_Z; i += TPB) {
                const int z = z0 + (i % ET_Z), y = y0 + (i / ET_Z) % ET_Y, x = tx0 + i / (ET_Z * ET_Y);
                if (z < g.nz && y < g.ny && x < g.nx) known[(x * g.ny + y) * g.nz + z] = 0;
            }
            continue;
        }
        __syncthreads();   // the previous tile's readers are done with the staged labels
        int salt = 0;
        asm volatile("" : "+v"(salt));
        edge_tile(g, rho, labels, known, 0, g.nx, list, list_count, small, nullptr, G, brick_rec, no_vacuum, tx0, y0, z0, salt);
    }
}

// The retraces leave known == -2 on exactly the start voxels they relabelled, all of them entries of the edge list: the list of
// the next edge_check is a pass over that list (round 4) instead of a sweep of the whole grid for the flag and a host wait.
// (n_changed: the retrace pass's count of relabelled voxels -- none, the usual case after a neargrid assignment: nothing to list)
__global__ __launch_bounds__(TPB) void k_list_changed(const int *__restrict__ list, const int *n_dev, const int8_t *__restrict__ known,
                                                      int *__restrict__ out, int *out_count, int out_cap, const int *n_changed) {
    __shared__ int s_buf[BlockAppender<1>::CAP], s_n[2];
    if (*n_changed == 0) return;
    BlockAppender<1> app;
    app.init(s_buf, s_n, out, out_count, out_cap);
    const int n = *n_dev;
    for (int base = blockIdx.x * TPB; base < n; base += gridDim.x * TPB) {   // (uniform per block)
        const int t = base + threadIdx.x;
        const int v = t < n ? list[t] : 0;
        const bool hit = t < n && known[v] == -2;
        app.add(hit ? 1 : 0, [&](int) { return v; });
    }
    app.finish();
}
// compaction of owned voxels with known == value, 16 voxels per thread, one atomic per block
#define CK_CHUNKS 4   // 16-voxel chunks per thread: all loads of a thread in flight, one scan + one atomic per 16 K voxels
__global__ __launch_bounds__(TPB) void k_compact_known16(GridL g, const int8_t *__restrict__ known, int value,
                                                         int *__restrict__ list, int *count) {
    const long long vbeg = (long long)g.x0 * g.nyz, vend = (long long)g.x1 * g.nyz;
    long long base[CK_CHUNKS];
    int8_t b[CK_CHUNKS][16];
#pragma unroll
    for (int c = 0; c < CK_CHUNKS; c++) {
        base[c] = vbeg + (((long long)blockIdx.x * CK_CHUNKS + c) * TPB + threadIdx.x) * 16;
        if (base[c] + 16 <= vend && ((vbeg & 15) == 0)) {
            *reinterpret_cast<uint4 *>(b[c]) = *reinterpret_cast<const uint4 *>(known + base[c]);
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) b[c][k] = (base[c] + k < vend) ? known[base[c] + k] : (int8_t)(value + 1);
        }
    }
    int cnt = 0;
#pragma unroll
    for (int c = 0; c < CK_CHUNKS; c++)
#pragma unroll
        for (int k = 0; k < 16; k++) cnt += (b[c][k] == value);
    int total;
    const int off = block_scan_excl(cnt, total);
    __shared__ int base_s;
    if (threadIdx.x == 0) base_s = total ? atomicAdd(count, total) : 0;
    __syncthreads();
    int w = base_s + off;
#pragma unroll
    for (int c = 0; c < CK_CHUNKS; c++)
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (b[c][k] == value) list[w++] = (int)(base[c] + k);
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
// RHO: the kernel carries the from-rho fallback (make_rec_rho, register hungry) for voxels whose record is not in the
// table -- outside the table window of a slab, or (sparse table, `brick_rec`) in a brick without records.  The
// lean instantiation (single GPU) never derives one:
//  * regions_ok (the labels are the last neargrid assignment's, no vacuum): a brick without records is a brick of a
//    trapping region; the region is closed under every possible move of both tie rules (k_brick_masks), so the
//    retrace can only end inside it -- on a known == 2 voxel or on the maximum -- and every voxel of the region
//    carries the region's label: the retrace stops at q and takes labels[q] (refinement.py:283-303);
//  * otherwise the retrace goes to `defer_list` (redone by the RHO instantiation), but only if its walk has to go ON
//    through the missing record: the path-membership test only needs a value <= the key q would have been pushed
//    with (key_floor).
//
// Slabs (WalkerIO): a retrace that leaves the valid planes before it ends is parked (known == -6) and its state at
// the moment of arrival on the first invalid voxel is written to `wio.out` (Walker).  The scheduler hands the walkers
// to the rank that owns that plane, which carries them on with the RESUME instantiation (its labels / known are the
// authoritative ones there), writes (start voxel, final label) pairs for those that end and exports the others
// again; the owner of the start voxel applies the pair (k_walkers_apply).  A retrace only reads labels at
// known == 2 voxels / maxima, which no retrace rewrites, so where it is finished does not matter.
struct Walker {          // 80 bytes, no padding; travels as 10 int64
    int v, vol_num;      // start voxel and its label when the retrace started
    int lp, lq;          // last voxel of the path, voxel just arrived at (not yet tested / pushed)
    double dr0, dr1, dr2;
    int widx[2];         // PathWindow<2>
    double wval[2];
    double m_old;
    int steps, og_move;
};
struct WalkerIO {
    const Walker *in;    // RESUME: the walkers of every rank; the ones arriving in [own0, own1) are carried on
    Walker *out;         // exported walkers (null: escaped retraces are only parked)
    int *out_count;
    int out_cap;
    int *res;            // RESUME: pairs (start voxel, final label; XB_WALKER_STUCK: needs the exact slow path)
    int *res_count;
    int own0, own1;
};
#define XB_WALKER_STUCK (-2147483647 - 1)

// Tried in round 4 and dropped: the retraces of one GPU in PHASES -- a first launch gives every retrace a few steps and hands the
// ones still moving over as walkers (80 B each, segmented buffers, wave-aggregated slot counters), a second launch packs the
// survivors into full waves, a third finishes the rest.  Lane use rises from 29 of 64 as intended, but the kernel is not bound
// by its wave-steps: 0.52 ms in one launch against 0.76 (budget 8) ... 1.05 ms (budgets 2 + 4) in phases at 512^3 -- every
// hand-over is a record the next launch has to gather again, and idle lanes cost a latency-bound kernel nothing.
// EXPORT (slabs, device-driven step): the lean instantiation hands a retrace that leaves the valid planes over as a walker
// itself (it holds the whole state) instead of deferring it to the from-rho instantiation, which walked it again from its
// start only to export it at the same voxel.
template <int K, bool RHO, bool RESUME = false, bool EXPORT = false>
__global__ __launch_bounds__(TPB) void k_refine_trace(GridL g, const GradRec *__restrict__ G, int *labels,
                                                      int8_t *known, const int *__restrict__ list, int n_host,
                                                      const int *n_dev, int *changed, int *escaped, int *ovf_list,
                                                      int *ovf_count, int ovf_cap, int maxsteps,
                                                      const double *__restrict__ rho, const double *__restrict__ gc,
                                                      const unsigned char *__restrict__ brick_rec, int *defer_list,
                                                      int *defer_count, int regions_ok, const int *__restrict__ region_blab,
                                                      WalkerIO wio) {
    static_assert(!RESUME || (RHO && K == 2), "walkers are carried on by the from-rho kernel");
    const int n = n_dev ? *n_dev : n_host;   // the list length may live on the device: the grid strides over it
    int n_ch = 0, n_es = 0;
  // (XCD k retraces the k-th contiguous eighth of the list, which is in tile order: xcd_range)
  const XcdRange xr = xcd_range((n + (int)blockDim.x - 1) / (int)blockDim.x);
  for (int chunk = xr.begin; chunk < xr.end; chunk += xr.step) {   // uniform per block (any block size up to TPB)
    const int t = chunk * (int)blockDim.x + threadIdx.x;
    bool valid = t < n;
    int v = (valid && !RESUME) ? list[t] : 0;
    bool moving = false;
    int result = -3;  // terminal voxel index; -2 overflow; -4 escaped; -5 deferred to the from-rho kernel
    int px = 0, py = 0, pz = 0, lp = 0, steps = 0, vol_num = 0;
    double dr0 = 0., dr1 = 0., dr2 = 0.;
    GradRec rec = {0., 0., 0., 0.};
    PathWindow<K> w;
    w.init(0, 0.);
    bool arrive = false, og_in = false;   // RESUME: the first pass of the loop only tests the voxel arrived at
    int lq_in = 0;
    if (RESUME) {
        if (valid) {
            const Walker wk = wio.in[t];
            const int qx = wk.lq / g.nyz;
            valid = qx >= wio.own0 && qx < wio.own1;
            if (valid) {
                v = wk.v; vol_num = wk.vol_num; lp = wk.lp; lq_in = wk.lq;
                px = lp / g.nyz;
                const int r = lp - px * g.nyz;
                py = r / g.nz;
                pz = r - py * g.nz;
                dr0 = wk.dr0; dr1 = wk.dr1; dr2 = wk.dr2;
                w.idx[0] = wk.widx[0]; w.idx[K - 1] = wk.widx[1];
                w.val[0] = wk.wval[0]; w.val[K - 1] = wk.wval[1];
                w.m_old = wk.m_old;
                steps = wk.steps; og_in = wk.og_move != 0;
                arrive = true; moving = true;
            }
        }
    } else
    if (valid) {
        px = v / g.nyz;
        const int r = v - px * g.nyz;
        py = r / g.nz;
        pz = r - py * g.nz;
        lp = v;
        rec = fetch_rec_w(g, G, v);
        vol_num = labels[v];
        moving = true;
        if (region_blab && region_blab[((px >> 3) * ((g.ny + 7) >> 3) + (py >> 3)) * ((g.nz + 7) >> 3) + (pz >> 3)] > 0) {
            result = v; moving = false;   // slabs: an edge voxel inside a trapping region keeps its label (see below)
        } else
        if (!(plane_in_window(g, px) && rec_exists(brick_rec, g, px, py, pz))) {   // an edge voxel without a record
            if (RHO) rec = make_rec_rho(g, rho, gc, px, py, pz);
            else { result = regions_ok ? v : -5; moving = false; }   // inside a trapping region: it ends there, label unchanged
        }
        w.init(v, rec.key);
    }
#ifdef XB_DEBUG_COUNT
    int dbg_it = 0;
#endif
    while (__any(moving)) {
        if (moving) {
#ifdef XB_DEBUG_COUNT
            dbg_it++;
#endif
            int qx = 0, qy = 0, qz = 0, lq = 0;
            bool og_move = false;
            if (RESUME && arrive) {   // the move was made by the rank the walker comes from
                arrive = false;
                lq = lq_in; og_move = og_in;
                qx = lq / g.nyz;
                const int r = lq - qx * g.nyz;
                qy = r / g.nz;
                qz = r - qy * g.nz;
            } else {
            const int bits = key_bits(rec.key);
            const int code = bits & 63;
            og_move = (code == XB_STAY_CODE);
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
            }
            if (moving) {
                const bool in_win = plane_in_window(g, qx);
                const bool ok_plane = plane_valid(g, qx);
                // the record is gathered speculatively, together with the flag and the brick byte: asking the brick byte
                // first saved the gathers of never-written records (most of this kernel's HBM traffic) but cost more in
                // dependent latency than it saved (0.49 -> 0.55 ms)
                GradRec nr = fetch_rec(G, in_win ? rec_slot(g, lq) : 0);   // (outside the window: any valid slot, the value is not used)
                const int8_t kq = known[ok_plane ? lq : lp];
                const bool missing = !(in_win && rec_exists(brick_rec, g, qx, qy, qz));
                // slabs (region_blab: the brick labels of the trapping regions, the same on every rank): a retrace that
                // enters a region ends in it with the region's label -- it no longer glides along a dividing surface for
                // tens of planes, so a narrow label halo is enough and the remote path queries become rare
                const bool in_region = region_blab && ok_plane && region_blab[((qx >> 3) * ((g.ny + 7) >> 3) + (qy >> 3)) * ((g.nz + 7) >> 3) + (qz >> 3)] > 0;
                if (in_region || (!RHO && regions_ok && in_win && missing && ok_plane)) {
                    // q lies in a trapping region (closed, one label): the retrace ends in it whatever happens next --
                    // no record, no density, no membership test needed (q cannot be an old path voxel: the path would
                    // not have left the region)
                    result = lq;
                    moving = false;
                } else {
                    if (missing && ok_plane) {
                        if (RHO) nr = make_rec_rho(g, rho, gc, qx, qy, qz);
                        else nr.key = key_floor(rho[lq]);
                    }
                    if (!ok_plane) {
                        result = (RHO || EXPORT || !defer_list) ? -4 : -5;   // (lean kernel on a slab without EXPORT: the from-rho pass redoes and exports it)
                        moving = false;
                        if ((RHO || EXPORT) && K == 2 && wio.out) {   // hand the walker over as it arrives at q
                            const int k = atomicAdd(wio.out_count, 1);
                            if (k < wio.out_cap) {
                                Walker wk;
                                wk.v = v; wk.vol_num = vol_num; wk.lp = lp; wk.lq = lq;
                                wk.dr0 = dr0; wk.dr1 = dr1; wk.dr2 = dr2;
                                wk.widx[0] = w.idx[0]; wk.widx[1] = w.idx[K - 1];
                                wk.wval[0] = w.val[0]; wk.wval[1] = w.val[K - 1];
                                wk.m_old = w.m_old; wk.steps = steps; wk.og_move = og_move ? 1 : 0;
                                wio.out[k] = wk;
                            }
                        }
                    }
                    else if ((!og_move && nr.key <= w.m_old) || ++steps > maxsteps) { result = -2; moving = false; }
                    else if (kq == 2) { result = lq; moving = false; }  // refinement.py:294-303
                    else if (!RHO && missing) { result = -5; moving = false; }
                    else {
                        w.push(lq, nr.key);
                        px = qx; py = qy; pz = qz; lp = lq; rec = nr;
                    }
                }
            }
        }
    }
#ifdef XB_DEBUG_COUNT
    if (xb_dbg_steps && valid && !RESUME) xb_dbg_steps[v] = (signed char)min(dbg_it + 1, 127);   // (+1: an edge voxel whose retrace ends before its first step still counts)
#endif
    int ch = 0, es = 0;
    if (RESUME) {
        if (valid) {
            if (result == -4) es = 1;    // exported again: it left this rank's valid planes too
            else {
                const int k = atomicAdd(wio.res_count, 1);
                wio.res[2 * k] = v;
                wio.res[2 * k + 1] = result >= 0 ? labels[result] : XB_WALKER_STUCK;
            }
        }
    } else
    if (valid) {
        if (result >= 0) {
            const int nv = labels[result];
            if (nv != vol_num) { labels[v] = nv; known[v] = -2; ch = 1; }  // refinement.py:288-289
            else known[v] = -1;                                             // refinement.py:291 (+5 +1 -5)
        } else if (result == -2) {
            const int k = atomicAdd(ovf_count, 1);
            if (k < ovf_cap) ovf_list[k] = v;
        } else if (result == -4) { known[v] = -6; es = 1; }  // left the valid slab: parked for the fallback
        else if (result == -5) defer_list[atomicAdd(defer_count, 1)] = v;   // the list takes every voxel: no overflow
    }
    n_ch += ch; n_es += es;
  }
    for (int o = 32; o > 0; o >>= 1) { n_ch += __shfl_down(n_ch, o); n_es += __shfl_down(n_es, o); }
    if (threadIdx.x % XB_WAVE == 0) {
        if (n_ch) atomicAdd(changed, n_ch);
        if (n_es) atomicAdd(escaped, n_es);
    }
}

// the (start voxel, final label) pairs of carried-on walkers, applied by the owner of the start voxel exactly as the
// retrace itself would have (refinement.py:288-291); a stuck walker stays parked (known == -6)
__global__ void k_walkers_apply(GridL g, const int *__restrict__ res, int n, int own0, int own1, int *labels, int8_t *known,
                                int *changed, int *stuck) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int v = res[2 * t], nv = res[2 * t + 1];
    const int x = v / g.nyz;
    if (x < own0 || x >= own1) return;
    if (nv == XB_WALKER_STUCK) { atomicAdd(stuck, 1); return; }
    if (nv != labels[v]) { labels[v] = nv; known[v] = -2; atomicAdd(changed, 1); }
    else known[v] = -1;
}

// ---------------------------------------------------------------------------------------------
// refinement.edge_check (refinement.py:409-508).  The sequential scan re-classifies the 27-box of
// every voxel that is still -2 when the scan reaches it; an earlier processed neighbour j < i
// rewrites i to -1 / -3 unless i is an (edge & maximum) voxel, in which case i is processed too.
// So the processed set P is the lexicographically-first greedy choice:
//     i in P  <=>  class(i) == edge&max  or  no j in P with j < i, j in box(i).
// P is resolved in dependency order (a voxel decides once all earlier changed neighbours have decided,
// k_ec_first + k_ec_chase below); the final `known` is then a pure function of P and the static classes:
//   edge&max voxel            -> processed at once (earlier boxes leave it -2, refinement.py:480)
//   an earlier neighbour is P -> skipped
//   no earlier neighbour left undecided -> processed
// temp codes in `known`: -2 (0xFE) undecided, -4 (0xFC) processed, -10 (0xF6) skipped -- both decisions
// clear one bit of 0xFE, so a lane claims and publishes a decision with a single atomicAnd on the
// aligned word holding the status byte (the returned word tells whether it was first).
#define EC_PROC (-4)
#define EC_SKIP (-10)
// ---------------------------------------------------------------------------------------------
// Resolution by dependency counters (a topological order of the greedy choice).  Per listed voxel 16 bits of
// `pend` (indexed by voxel):
//     bits 0-7   number of C-order EARLIER listed neighbours that have not decided yet
//     bits 8-14  number of earlier neighbours that were processed
//     bit  15    the voxel is an edge&maximum voxel (processed whatever its neighbours do)
// A voxel is decidable as soon as bit 15 is set (processed), an earlier neighbour was processed (skipped) or
// the counter reaches zero (processed); the lane that decides it (claim by atomicAnd on the status word: one
// winner in the grid) tells every LATER listed neighbour with ONE atomicAdd on its 16 bits (-1, or +0x100-1
// when it was processed itself) and queues exactly the neighbours that became decidable through it (first
// processed neighbour, or the decrement that reached zero with none processed).  Every voxel is evaluated once,
// from its 16 bits alone (when the counter reaches zero every earlier neighbour has already contributed), so a
// chain step costs a couple of device-scope round trips instead of re-reading 27 statuses for every wake-up;
// the result is the sequential one because each rule is the reference's (refinement.py:428-470) applied to
// final inputs.  `known` of a listed voxel is -2 / -4 / -10 at any time, which is how the later listed
// neighbours are found (membership does not change).
// The code below is written branch-free on purpose: with a branch per neighbour the compiler waits for every
// load / atomic before it issues the next one (measured: ~20 serial round trips, 10 us per chain step).
// ---------------------------------------------------------------------------------------------
// Round 2: the word is 64 bits wide and also holds, in bits 32-58, WHICH box positions of the voxel are later listed
// neighbours.  The atomicAdd that wakes a voxel returns that mask with the counters, the queue entry carries it, and
// the consumer needs no load of its own before it notifies its neighbours: a chain step inside the chase is ONE
// device-scope round trip (the notifications) instead of two (status claim + box loads, then the notifications).
// A queue entry is unique by construction (exactly one notifier sees the first processed neighbour arrive / the
// counter reach zero), so its status is published without waiting for the claim; only the blanket scan of round 1 and
// the seeds it leaves (a voxel may be decided by its scan thread AND queued by a notifier) still claim first.
#define EC_CNT 0x00FFu
#define EC_NPROC 0x7F00u
#define EC_CLS1 0x8000u
typedef unsigned long long ec_word;
__device__ __forceinline__ bool ec_listed(int8_t k) { return k == -2 || k == EC_PROC || k == EC_SKIP; }
// status bytes of the 27-box of (x,y,z) as 9 words: byte (iz+1) of word (ix+1)*3+(iy+1); all loads in flight
__device__ __forceinline__ void ec_box(const Grid &g, const int8_t *__restrict__ known, int x, int y, int z, int rows[9],
                                       unsigned int w[9]) {
    const bool z_inner = z >= 1 && z + 2 < g.nz;
    const int zm = wrapi(z - 1, g.nz), zp = wrapi(z + 1, g.nz);
#pragma unroll
    for (int ix = -1; ix < 2; ix++) {
        const int tx = wrapi(x + ix, g.nx);
#pragma unroll
        for (int iy = -1; iy < 2; iy++) {
            const int row = (tx * g.ny + wrapi(y + iy, g.ny)) * g.nz;
            rows[(ix + 1) * 3 + iy + 1] = row;
        }
    }
    if (z_inner) {  // z-1..z+1 contiguous: one (unaligned) 32-bit load per row
#pragma unroll
        for (int k = 0; k < 9; k++) w[k] = *reinterpret_cast<const unsigned int *>(known + rows[k] + z - 1);
    } else {
#pragma unroll
        for (int k = 0; k < 9; k++)
            w[k] = (unsigned int)(uint8_t)known[rows[k] + zm] | ((unsigned int)(uint8_t)known[rows[k] + z] << 8) |
                   ((unsigned int)(uint8_t)known[rows[k] + zp] << 16);
    }
}
__device__ __forceinline__ int ec_box_voxel(const Grid &g, const int rows[9], int z, int j) {
    return rows[j / 3] + wrapi(z + (j % 3) - 1, g.nz);
}
// pend[v] := (listed earlier neighbours) | (edge&max ? bit 15); every listed voxel once, nothing decided yet
__global__ __launch_bounds__(TPB) void k_ec_init(Grid g, const double *__restrict__ rho, const int *__restrict__ labels,
                                                 const int8_t *__restrict__ known, const int *__restrict__ list, int n,
                                                 ec_word *pend) {
    const XcdRange xr = xcd_range((n + TPB - 1) / TPB);   // (a grid of one workgroup per chunk: each takes one)
    const int t = xr.begin * TPB + threadIdx.x;
    if (xr.begin >= xr.end || t >= n) return;
    const int v = list[t];
    const int x = v / g.nyz;
    const int r = v - x * g.nyz;
    const int y = r / g.nz, z = r - y * g.nz;
    int rows[9];
    unsigned int w[9];
    ec_box(g, known, x, y, z, rows, w);
    int cnt = 0;
    unsigned int later = 0;
#pragma unroll
    for (int j = 0; j < 27; j++) {
        const int8_t k = (int8_t)((w[j / 3] >> (8 * (j % 3))) & 0xff);
        const int u = ec_box_voxel(g, rows, z, j);
        cnt += (u < v) & (k == -2);
        later |= (unsigned int)((u > v) & (k == -2)) << j;
    }
    // class: edge & maximum.  A denser non-vacuum FACE neighbour already rules the maximum out (13 loads
    // instead of the 54 of the full classification)
    const double c0 = rho[v];
    bool denser = false;
#pragma unroll
    for (int f = 0; f < 6; f++) {
        const int d = (f & 1) ? 1 : -1;
        const int l = f < 2 ? lin3(g, wrapi(x + d, g.nx), y, z) : (f < 4 ? lin3(g, x, wrapi(y + d, g.ny), z) : lin3(g, x, y, wrapi(z + d, g.nz)));
        denser |= (labels[l] != -1) & (rho[l] > c0);
    }
    bool cls1 = false;
    if (!denser) {
        bool is_edge, is_max;
        classify27(g, rho, labels, x, y, z, v, is_edge, is_max);
        cls1 = is_edge && is_max;
    }
    pend[v] = (ec_word)(cnt | (cls1 ? EC_CLS1 : 0u)) | ((ec_word)later << 32);
}
// Slabs: the class bit alone, for the owned listed voxels (their 27-boxes lie in label-valid planes) ...
__global__ __launch_bounds__(TPB) void k_ec_class(Grid g, const double *__restrict__ rho, const int *__restrict__ labels,
                                                  const int *__restrict__ list, int n, int8_t *cls) {
    const int t = blockIdx.x * TPB + threadIdx.x;
    if (t >= n) return;
    const int v = list[t];
    const int x = v / g.nyz;
    const int r = v - x * g.nyz;
    bool is_edge, is_max;
    classify27(g, rho, labels, x, r / g.nz, r % g.nz, v, is_edge, is_max);
    cls[t] = (is_edge && is_max) ? 1 : 0;
}
// ... and the 16 bits of every voxel of the GLOBAL list from the classes their owners computed
__global__ __launch_bounds__(TPB) void k_ec_init_cls(Grid g, const int8_t *__restrict__ known, const int *__restrict__ list, int n,
                                                     const int8_t *__restrict__ cls, ec_word *pend) {
    const int t = blockIdx.x * TPB + threadIdx.x;
    if (t >= n) return;
    const int v = list[t];
    const int x = v / g.nyz;
    const int r = v - x * g.nyz;
    const int y = r / g.nz, z = r - y * g.nz;
    int rows[9];
    unsigned int w[9];
    ec_box(g, known, x, y, z, rows, w);
    int cnt = 0;
    unsigned int later = 0;
#pragma unroll
    for (int j = 0; j < 27; j++) {
        const int8_t k = (int8_t)((w[j / 3] >> (8 * (j % 3))) & 0xff);
        const int u = ec_box_voxel(g, rows, z, j);
        cnt += (u < v) & (k == -2);
        later |= (unsigned int)((u > v) & (k == -2)) << j;
    }
    pend[v] = (ec_word)(cnt | (cls[t] ? EC_CLS1 : 0u)) | ((ec_word)later << 32);
}
__global__ void k_scatter_byte(int8_t *a, const int *__restrict__ idx, int n, int8_t value) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) a[idx[t]] = value;
}
// st[t] = 0 for the entries whose 27-box cannot touch the planes [xa, xa + np) (mod nx): their boxes are not applied
__global__ void k_ec_keep_near(Grid g, const int *__restrict__ list, int n, int8_t *st, int xa, int np) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    int d = list[t] / g.nyz - xa;
    if (d < 0) d += g.nx;
    if (d >= np) st[t] = 0;
}
// Decide v if its word allows it and tell the later listed neighbours; push(e) receives, for every voxel that became
// decidable through this decision, its queue entry.
// `entry` (64 bits): bits 0-30 the voxel; bit 63 set when the notifier already knows the decision is "skipped", bit 62
// when it knows "processed" (the notifier has the neighbour's word from its atomic); bits 32-58 the voxel's later
// listed neighbours (from the same word).  Neither flag: read the word (blanket scan of round 1, where a voxel may not be
// decidable yet; and every entry that went through a 32-bit list -- seeds, queue overflows: the word says the same thing once the
// notification that made the entry has landed, and it has: the notifier had its atomic's result).  Round 6: the flags used to sit
// in bits 30 / 31 of the voxel's own word, which capped 'changed' refinement at 2^30 voxels -- exactly 1024^3.  CLAIM: wait for the status claim and give way to whoever decided the voxel first (blanket scan and
// its seeds); without it the entry is known to be the only one for its voxel and the status is published on the side.
#define EC_E_SKIP 0x8000000000000000ull
#define EC_E_PROC 0x4000000000000000ull
#define EC_E_VOXEL 0x7FFFFFFFull
#define EC_E_LATER 0x07FFFFFF00000000ull
template <bool CLAIM, typename Push>
__device__ __forceinline__ void ec_resolve(const Grid &g, int8_t *known, ec_word *pend, ec_word entry, Push push) {
    const int v = (int)(entry & EC_E_VOXEL);
    int d;
    unsigned int later;
    if ((entry & (EC_E_SKIP | EC_E_PROC)) && !CLAIM) {
        d = (entry & EC_E_SKIP) ? 2 : 1;
        later = (unsigned int)((entry & EC_E_LATER) >> 32);
    } else {
        const ec_word b = __hip_atomic_load(pend + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        later = (unsigned int)(b >> 32);
        if (entry & EC_E_SKIP) d = 2;
        else if (entry & EC_E_PROC) d = 1;
        else if (b & EC_CLS1) d = 1;
        else if (b & EC_NPROC) d = 2;
        else if ((b & EC_CNT) == 0) d = 1;
        else return;  // still waiting for an earlier neighbour (blanket scan only)
    }
    {   // publish (and, CLAIM, claim): one atomicAnd clears the decision's bit of 0xFE; the returned word names the winner
        const int sh = (v & 3) * 8;
        unsigned int *word = reinterpret_cast<unsigned int *>(known + (v & ~3));
        const unsigned int clr = ~((d == 1 ? 0x02u : 0x08u) << sh);
        if (CLAIM) {
            const unsigned int old = atomicAnd(word, clr);
            if (((old >> sh) & 0xffu) != 0xFEu) return;
        } else
            __hip_atomic_fetch_and(word, clr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // result unused: no wait
    }
    const int x = v / g.nyz;
    const int r = v - x * g.nyz;
    const int y = r / g.nz, z = r - y * g.nz;
    int rows[9];
#pragma unroll
    for (int ix = -1; ix < 2; ix++) {
        const int tx = wrapi(x + ix, g.nx);
#pragma unroll
        for (int iy = -1; iy < 2; iy++) rows[(ix + 1) * 3 + iy + 1] = (tx * g.ny + wrapi(y + iy, g.ny)) * g.nz;
    }
    // One atomicAdd per later listed neighbour, issued together in 10 slots (a voxel rarely has more; the rest is handled
    // one by one below).  The atomics of a workgroup share one CU's
    // address unit, so 10 instead of one per box position matters.
    const ec_word delta = d == 1 ? 0xFFull : ~0ull;  // +0x100 - 1  |  -1  (the count never borrows: it counts this very neighbour)
    ec_word o[10];
    int uu[10];
#pragma unroll
    for (int s = 0; s < 10; s++) {
        const bool hit = later != 0;
        const int j = hit ? __ffs(later) - 1 : 13;
        later &= later - 1;
        const int u = hit ? ec_box_voxel(g, rows, z, j) : v;
        uu[s] = hit ? u : -1;
        o[s] = 0;
        if (hit) o[s] = atomicAdd(pend + u, delta);   // (predicated: an empty slot issues nothing; -0.15 ms of 3.5 at 512^3)
    }
#pragma unroll
    for (int s = 0; s < 10; s++) {
        const unsigned int ob = (unsigned int)o[s] & 0xffffu;
        const bool wake = (uu[s] >= 0) & !(ob & (EC_NPROC | EC_CLS1)) & ((d == 1) | ((ob & EC_CNT) == 1));
        // first processed earlier neighbour: u gets skipped / the last one u waited for, none processed: processed
        if (wake) push((ec_word)(unsigned int)uu[s] | (d == 1 ? EC_E_SKIP : EC_E_PROC) | (o[s] & EC_E_LATER));
    }
    while (later) {  // the later neighbours beyond the slots
        const int u = ec_box_voxel(g, rows, z, __ffs(later) - 1);
        later &= later - 1;
        const ec_word ow = atomicAdd(pend + u, delta);
        const unsigned int ob = (unsigned int)ow & 0xffffu;
        if (!(ob & (EC_NPROC | EC_CLS1)) && (d == 1 || (ob & EC_CNT) == 1))
            push((ec_word)(unsigned int)u | (d == 1 ? EC_E_SKIP : EC_E_PROC) | (ow & EC_E_LATER));
    }
}
// The same for a QUEUE entry of the chase (the decision and the later-neighbour mask ride in the entry), spread over 16 lanes:
// lane `sub` notifies the sub-th later neighbour.  Round 4: a chain hop is one round of a workgroup in which a single wave has
// work, so the hop lasts as long as that wave's longest dependent instruction chain -- and walked by one lane, the ten
// notifications of a voxel (decode with two integer divisions, nine row bases, ten address computations, ten wake tests, the
// pushes) were ~700 instructions, 1.3 of the 3.0 us per hop; a device-scope atomic with its result takes 0.4-0.7 us on this
// card and a pair of barriers 0.08 (tools/ubench_rtt.hip).  Sixteen lanes cut the chain to ~100 instructions.
#define EC_LANES 8
#define EC_LONG_N 128   // a queue longer than this takes the lane-per-entry form (measured: 64 / 96 / 128 / 256 / 512 entries)
template <typename Push>
__device__ __forceinline__ void ec_resolve_lanes(const Grid &g, double inv_nyz, double inv_nz, int8_t *known, ec_word *pend, ec_word entry,
                                                 int sub, Push push) {
    const int v = (int)(entry & EC_E_VOXEL);
    const int d = (entry & EC_E_SKIP) ? 2 : 1;
    const unsigned int later = (unsigned int)((entry & EC_E_LATER) >> 32);
    if (sub == 0) {   // publish the decision (a queue entry is the only one for its voxel: nobody waits for the result)
        const int sh = (v & 3) * 8;
        __hip_atomic_fetch_and(reinterpret_cast<unsigned int *>(known + (v & ~3)), ~((d == 1 ? 0x02u : 0x08u) << sh), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    // Round 5: lane `sub` takes the box POSITIONS sub, sub + 8, sub + 16, sub + 24 (one mask operation; round 4 gave it the sub-th and
    // the (sub + 8)-th SET bit, found by clearing bits one at a time: two loops of up to 7 and 8 dependent steps in front of
    // the atomics of every hop).  The later neighbours are the 13 positions behind the centre -- 14 .. 26, at most two to a lane --
    // unless the box wraps around the grid (then any position can be later): a lane issues up to four atomics, all in flight
    // together.
    static_assert(EC_LANES == 8, "a lane's positions are sub + 8 k");
    unsigned int mine = later & (0x01010101u << sub);
    if (!mine) return;
    // v -> (x, y, z) with two multiplications by reciprocals and a correction step (exact for every int32 index; the product in
    // unsigned arithmetic: a quotient one too large may pass 2^31 - 1 on the largest grids)
    int x = (int)((double)v * inv_nyz);
    int r = (int)((unsigned)v - (unsigned)x * (unsigned)g.nyz);
    if (r < 0) { x--; r += g.nyz; } else if (r >= g.nyz) { x++; r -= g.nyz; }
    int y = (int)((double)r * inv_nz);
    int z = r - y * g.nz;
    if (z < 0) { y--; z += g.nz; } else if (z >= g.nz) { y++; z -= g.nz; }
    const ec_word delta = d == 1 ? 0xFFull : ~0ull;  // +0x100 - 1  |  -1
    ec_word o[4];
    int u[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = sub + 8 * k;
        u[k] = -1;
        o[k] = 0;
        if (j < 27 && ((mine >> j) & 1u)) {
            u[k] = lin3(g, wrapi(x + j / 9 - 1, g.nx), wrapi(y + (j / 3) % 3 - 1, g.ny), wrapi(z + j % 3 - 1, g.nz));
            o[k] = atomicAdd(pend + u[k], delta);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned int b = (unsigned int)o[k] & 0xffffu;
        if (u[k] >= 0 && !(b & (EC_NPROC | EC_CLS1)) && (d == 1 || (b & EC_CNT) == 1))
            push((ec_word)(unsigned int)u[k] | (d == 1 ? EC_E_SKIP : EC_E_PROC) | (o[k] & EC_E_LATER));
    }
}
// Round 1: every listed voxel once; what is decidable at once (no earlier listed neighbour, or edge&max) is
// decided, the voxels that become decidable through these decisions seed the chase (32-bit entries: the seeds read
// their word again).
__global__ __launch_bounds__(TPB) void k_ec_first(Grid g, int8_t *known, ec_word *pend, const int *__restrict__ in,
                                                  int n, int *out, int *n_out, int out_cap) {
    const XcdRange xr = xcd_range((n + TPB - 1) / TPB);
    for (int chunk = xr.begin; chunk < xr.end; chunk += xr.step)
      if (const int e = chunk * TPB + (int)threadIdx.x; e < n)
        ec_resolve<true>(g, known, pend, (ec_word)(unsigned int)in[e], [&](ec_word u) {
            const int at = atomicAdd(n_out, 1);
            if (at < out_cap) out[at] = (int)(u & EC_E_VOXEL);
        });
}
// The rest: the dependency chains are ~1000 voxels long while only a few thousand voxels are decidable at any
// time, so global rounds (a launch or a grid barrier each) would cost 20-40 us per chain step.  Every workgroup
// chases its own share asynchronously instead -- no barrier across workgroups, nothing waits on another
// workgroup: a round resolves the workgroup's queue (LDS) and queues what became decidable for its next
// round.  All cross-workgroup traffic is device-scope atomics on the status and counter words (they resolve at
// the device coherence point; the workgroups sit on different XCDs, one L2 each) -- no __threadfence(): on the
// 8-XCD gfx950 an agent-scope fence writes back and invalidates the XCD's L2 (~2 us each).
// Queue overflows go to `ovf` and seed the next launch.
#define EC_CHASE_THREADS 1024
#define EC_Q 6000   // queue entries per buffer (2 buffers of 64-bit entries, 94 KB of LDS)
// Round 5: SHARING A LONG FRONT.  A workgroup keeps what it wakes, and the fronts are few: without sharing the workgroup that
// carries the longest one works through 1 082 rounds (the dynamic critical path) of which 550 hold ~70 entries and 180 more than
// 128 -- 1.9 and 3.4 us each, against 1.3 us for a round of a few entries (the XB_EC_PROBE build) -- because ~7 scattered
// device-scope atomics per entry go through ONE compute unit's address path, which issues ~600 of them per microsecond
// (tools/ubench_rtt.hip: one workgroup, 64 / 256 / 640 / 1024 lanes with one atomic each: 0.51 / 0.45 / 1.09 / 1.72 us per round),
// while most compute units sit idle.  Now the last wave of a workgroup SHEDS up to 64 entries per round of what the queue holds
// beyond EC_SHARE_KEEP into another workgroup's mailbox: one atomicAdd on the target's tail reserves the slots (each used once per
// launch), plain device-scope stores fill them; the receiver's last wave looks at a window of its mailbox every round, resolves
// up to eight arrivals on the spot and queues the rest, and the workgroup keeps what THAT wakes -- a front spreads over as many
// compute units as it needs within a few rounds.  The result does not depend on who resolves what: the decisions are a function
// of the words alone.
// Leaving without a barrier, and without assuming that all workgroups of the launch are resident (several contexts may share
// the card; a workgroup that waits for one that cannot start until it leaves would wait for ever): a workgroup with nothing to do
// CLOSES its mailbox -- atomicExch of the tail with EC_MB_CLOSED: the old value says how many slots were ever reserved; it stays
// until those are filled and consumed, and leaves when it has nothing to do again.  A sender whose reservation falls beyond the
// capacity -- a full mailbox or a closed one -- simply keeps the entries.  So every entry is resolved by somebody, whatever the
// order of events.  WHEN to close is only a matter of speed: `active` (a hint: the workgroups that have started and have work)
// is zero, or EC_LINGER idle rounds have passed -- helpers should be around while fronts are being carried.
#define EC_SHARE_KEEP 32
#ifndef EC_MB_CAP   // (EC_MB_CAP and EC_LINGER: the only two build-time values -- a TEST builds the library with a tiny mailbox, see below)
#define EC_MB_CAP 8192            // slots of a workgroup's mailbox (64-bit entries; single use per launch)
#endif
#define EC_MB_CLOSED (1 << 30)    // a tail at or beyond this: the mailbox takes no more entries
#ifndef EC_LINGER
#define EC_LINGER (1 << 15)       // idle rounds (~0.5 us each) after which a workgroup closes whatever the hint says
#endif
// (tests/test_gpu_parity.py builds the library with 64 slots and 40 rounds: nine of ten sheds then meet a full or a closed mailbox)
// the sharing block: int 0 the hint, int 32 an error flag, ints 33.. statistics, the tails (one int per workgroup) from int 64 on,
// the slots from byte EC_SLOTS_AT on; all zero at launch
#define EC_SLOTS_AT 4096
__host__ __device__ __forceinline__ size_t ec_share_bytes(int groups) { return EC_SLOTS_AT + (size_t)groups * EC_MB_CAP * sizeof(ec_word); }
__global__ __launch_bounds__(EC_CHASE_THREADS) void k_ec_chase(Grid g, int8_t *known, ec_word *pend,
                                                               const int *__restrict__ seeds, const int *n_seeds_dev, int *ovf,
                                                               int *n_ovf, int ovf_cap, int qcap, int *share) {
    __shared__ ec_word q[2][EC_Q];
    __shared__ int s_n[3];   // (three queue lengths in rotation: this round's, the next one's, and the one being reset)
    // What lane 0 of the last wave tells the workgroup -- leave / the hint it read / this mailbox is closed / the slots reserved
    // before it closed -- in TWO copies by round parity: a round reads copy [round & 1] at its top and writes copy [(round + 1) & 1]
    // in (c).  With one copy and one barrier per round a slow wave's read at the top of a round raced with the last wave's write of
    // the same round (the waves could then disagree on `shed` -- an entry resolved twice or by nobody -- or leave in different
    // rounds); the copy a round writes was last read a round ago, before that round's barrier.
    __shared__ int s_bc[2][4];
    enum { BC_STOP = 0, BC_ACT = 1, BC_CLOSED = 2, BC_RESERVED = 3 };
    const int n_seeds = min(*n_seeds_dev, ovf_cap);   // (the count stays on the device: the host does not wait for it)
    const int per = (n_seeds + gridDim.x - 1) / gridDim.x;
    int seed_cur = blockIdx.x * per;
    const int seed_end = min(seed_cur + per, n_seeds);
    if (threadIdx.x < 3) s_n[threadIdx.x] = 0;
    if (threadIdx.x < 2) { s_bc[threadIdx.x][BC_STOP] = 0; s_bc[threadIdx.x][BC_ACT] = gridDim.x; s_bc[threadIdx.x][BC_CLOSED] = 0; s_bc[threadIdx.x][BC_RESERVED] = 0; }
    __syncthreads();
    const double inv_nyz = 1.0 / (double)g.nyz, inv_nz = 1.0 / (double)g.nz;
    const int G = gridDim.x;
    int *tails = share ? share + 64 : nullptr;
    ec_word *slots = share ? reinterpret_cast<ec_word *>(reinterpret_cast<char *>(share) + EC_SLOTS_AT) : nullptr;
    const int wave = threadIdx.x / XB_WAVE, lane = threadIdx.x % XB_WAVE;
    const bool last_wave = share && wave == EC_CHASE_THREADS / XB_WAVE - 1;
    ec_word *my_box = share ? slots + (size_t)blockIdx.x * EC_MB_CAP : nullptr;
    int head = 0;              // (last wave, uniform over its lanes) the next slot of this workgroup's mailbox
    bool counted = false;      // (last wave, lane 0) this workgroup is part of the hint
    unsigned idle_rounds = 0;  // (last wave, lane 0)
    if (last_wave && lane == 0 && seed_cur < seed_end) { atomicAdd(share, 1); counted = true; }
    int dbg_shed = 0, dbg_rounds = 0, dbg_got = 0;
#ifdef XB_EC_PROBE
    unsigned long long pr_t[5] = {0, 0, 0, 0, 0}, pr_c[5] = {0, 0, 0, 0, 0}, pr_e[5] = {0, 0, 0, 0, 0};   // rounds by class: <= 8, <= 32, more entries, shedding, the long form
#endif
    // ONE barrier per round (round 5; two before): the length of the queue a round reads, of the one it fills and of the one after
    // that are three words in rotation -- the third is reset while nobody looks at it (it was read a round ago, before that round's
    // barrier, and is written from the next round on), so no barrier has to separate the reads of a round from the reset
    for (int cur = 0, round = 0, ci = 0;; cur ^= 1, round++, ci = ci == 2 ? 0 : ci + 1) {
        const int n = min(s_n[ci], qcap);  // qcap <= EC_Q (smaller only to exercise the overflow path in tests)
        const int take = n <= qcap / 2 ? min(EC_CHASE_THREADS, seed_end - seed_cur) : 0;  // uniform
        if (n + take == 0 && !share) break;
        const int *bc = s_bc[round & 1];
        if (share && bc[BC_STOP]) break;   // (uniform: written in the previous round, before its barrier; nobody writes this copy in this round)
        const int act_prev = bc[BC_ACT], closed = bc[BC_CLOSED], reserved = bc[BC_RESERVED];
        if (threadIdx.x == 0) s_n[ci == 0 ? 2 : ci - 1] = 0;   // the length the round after next will fill
        ec_word *nq = q[cur ^ 1];
        int *n_next = &s_n[ci == 2 ? 0 : ci + 1];
        auto push = [&](ec_word u) {
            const int at = atomicAdd(n_next, 1);
            if (at < qcap) nq[at] = u;
            else {  // queue full: hand over to the next launch (as a seed: it reads its word again)
                const int o = atomicAdd(n_ovf, 1);
                if (o < ovf_cap) ovf[o] = (int)(u & EC_E_VOXEL);
            }
        };
        // what this round gives away: up to 64 entries from the end of the queue -- in the chains' tail whatever it holds beyond
        // EC_SHARE_KEEP, of a long queue a wave's worth once a quarter of the workgroups are idle
        const int shed = !share ? 0 : (n <= EC_LONG_N ? min(max(n - EC_SHARE_KEEP, 0), XB_WAVE) : (4 * act_prev < 3 * G ? XB_WAVE : 0));
        const int n_own = n - shed;
        const bool work = n + take > 0;
        dbg_rounds += work ? 1 : 0;
#ifdef XB_EC_PROBE
        const unsigned long long pr_t0 = wall_clock64();
#endif
        if (last_wave) {
            // (a) the mailbox: a window from the head on (a closed one: the slots reserved before it closed)
            const int limit = closed ? reserved : EC_MB_CAP;
            ec_word got = 0;
            if (head + lane < limit) got = __hip_atomic_load(my_box + head + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int act = 1;
            if (lane == 0) act = __hip_atomic_load(share, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (b) this round's surplus
            if (shed) {
                unsigned h = (unsigned)(blockIdx.x * 977u + (unsigned)round * 131u);
                h ^= h >> 7;
                int target = (int)(h % (unsigned)(G - 1));
                if (target >= (int)blockIdx.x) target++;
                int slot = 0;
                if (lane == 0) slot = atomicAdd(tails + target, shed);
                slot = __shfl(slot, 0);
                if (lane < shed) {
                    const ec_word e = q[cur][n_own + lane];
                    if ((unsigned)(slot + lane) < (unsigned)EC_MB_CAP)
                        __hip_atomic_store(slots + (size_t)target * EC_MB_CAP + slot + lane, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        push(e);   // that mailbox is full or closed: the entry stays here
                }
                dbg_shed += shed;
            }
            // (a, continued) what arrived: the filled slots in front of the first gap; the first eight are resolved here and now
            // (eight lanes each, as in the tail form below), the others join the next round's queue
            const unsigned long long full = __ballot(got != 0);
            const int k = full == ~0ull ? XB_WAVE : __ffsll((unsigned long long)~full) - 1;
            if (k) {   // (uniform over the wave)
                const int now = min(k, XB_WAVE / EC_LANES);
                if (k > now) {
                    int at = 0;
                    if (lane == 0) at = atomicAdd(n_next, k - now);
                    at = __shfl(at, 0);
                    if (lane >= now && lane < k) {
                        if (at + lane - now < qcap) nq[at + lane - now] = got;
                        else {
                            const int o = atomicAdd(n_ovf, 1);
                            if (o < ovf_cap) ovf[o] = (int)(got & EC_E_VOXEL);
                        }
                    }
                }
                const ec_word mine = __shfl(got, lane / EC_LANES);
                if (lane / EC_LANES < now) ec_resolve_lanes(g, inv_nyz, inv_nz, known, pend, mine, lane % EC_LANES, push);
                head += k;
                dbg_got += k;
            }
            // (c) stay, close or leave (one lane decides; the others read the verdict at the top of the next round, after this round's barrier)
            if (lane == 0) {
                int n_stop = 0, n_closed = closed, n_reserved = reserved;
                if (work || k) {
                    idle_rounds = 0;
                    if (!counted && !closed) { atomicAdd(share, 1); counted = true; }
                } else {
                    if (counted) { atomicSub(share, 1); counted = false; }
                    idle_rounds++;
                    if (!closed) {
                        if (act <= 0 || idle_rounds > EC_LINGER) {
                            const int old = atomicExch(tails + blockIdx.x, EC_MB_CLOSED);
                            n_reserved = min(old, EC_MB_CAP);
                            n_closed = 1;
                        }
                    } else if (head >= reserved) n_stop = 1;   // closed, everything ever reserved is consumed, nothing to do: leave
                    else if (idle_rounds > (1u << 24)) { share[32] = 1; n_stop = 1; }   // (a reserved slot that never fills: loud, not endless)
                }
                int *nb = s_bc[(round + 1) & 1];
                nb[BC_STOP] = n_stop; nb[BC_ACT] = act; nb[BC_CLOSED] = n_closed; nb[BC_RESERVED] = n_reserved;
            }
        }
        if (n_own > EC_LONG_N) {   // a long queue (the first rounds): throughput counts, a lane per entry
            for (int e = threadIdx.x; e < n_own; e += EC_CHASE_THREADS) ec_resolve<false>(g, known, pend, q[cur][e], push);
        } else {                                       // the chains' tail: latency counts, EC_LANES lanes per entry
            for (int e0 = 0; e0 < n_own; e0 += EC_CHASE_THREADS / EC_LANES) {
                const int e = e0 + (int)(threadIdx.x / EC_LANES);
                if (e < n_own) ec_resolve_lanes(g, inv_nyz, inv_nz, known, pend, q[cur][e], threadIdx.x % EC_LANES, push);
            }
        }
        for (int e = threadIdx.x; e < take; e += EC_CHASE_THREADS)   // seeds: a lane each (they read their word and claim first)
            ec_resolve<true>(g, known, pend, (ec_word)(unsigned int)seeds[seed_cur + e], push);
        seed_cur += take;
        if (share && !work) __builtin_amdgcn_s_sleep(8);   // nothing to do: wait a little before looking again
        __syncthreads();  // the next round's queue is complete (every atomic's result was used: they have returned)
#ifdef XB_EC_PROBE
        if (work) {
            const unsigned long long dt = wall_clock64() - pr_t0;
            const int cls = n > EC_LONG_N ? 4 : (shed ? 3 : (n_own <= 8 ? 0 : (n_own <= 32 ? 1 : 2)));
            pr_t[cls] += dt; pr_c[cls]++; pr_e[cls] += n_own;
        }
#endif
    }
#ifdef XB_EC_PROBE
    if (threadIdx.x == 0)
        for (int k = 0; k < 5; k++) { xb_dbg[1024 + blockIdx.x * 15 + k] = pr_t[k]; xb_dbg[1024 + blockIdx.x * 15 + 5 + k] = pr_c[k]; xb_dbg[1024 + blockIdx.x * 15 + 10 + k] = pr_e[k]; }
#endif
    if (last_wave && lane == 0) {   // statistics (debug switch 4 prints them)
        atomicAdd(share + 33, dbg_shed); atomicAdd(share + 35, dbg_got); atomicMax(share + 34, dbg_rounds); atomicAdd(share + 36, dbg_rounds);
    }
}
// the voxels the resolution left undecided (must be none) are counted for a loud failure
// st[t] = 1 for the processed entries (the known codes get overwritten during the apply pass)
__global__ void k_ec_collect(const int8_t *__restrict__ known, const int *__restrict__ list, int n, int8_t *st,
                             int *undecided) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int8_t k = known[list[t]];
    st[t] = (k == EC_PROC) ? 1 : 2;
    if (k == -2) atomicAdd(undecided, 1);
}
// apply: every processed voxel re-classifies its 27-box (refinement.py:428-504).
// Round 4: every box voxel is classified ONCE.  The boxes of neighbouring processed voxels overlap ninefold on a dividing
// surface, and round 3 classified every (processed voxel, box position) pair: 729 label loads per processed voxel, 0.79 ms
// and 2.4 GB of fetches at 512^3.  Now the processed voxels carry a flag byte in an array of their own (`pflag`, indexed by
// voxel, zero everywhere else: k_ec_mark sets and clears it); for a box voxel u a thread reads the 27 flags around u (nine
// 3-byte rows), and only the processed voxel with the SMALLEST index among them classifies u -- no atomics, no duplicates.
// The same 27 flags give `checked` the multiplicity the sequential scan gives it: a non-edge voxel is "checked" once per
// processed voxel whose box holds it (refinement.py:477-479).  (Tried first: claiming box voxels with atomicOr on a bitmap --
// 27 scattered atomics per processed voxel cost more than the loads they saved, 0.86 ms.)
// (plist / n_proc: the processed voxels are also listed -- a quarter of the entries at 512^3 --, so that k_ec_apply spends its
// threads on their boxes only)
__global__ __launch_bounds__(TPB) void k_ec_mark(const int *__restrict__ list, int n, const int8_t *__restrict__ st, int8_t *__restrict__ pflag,
                                                 int8_t value, int *plist, int *n_proc, int pcap) {
    __shared__ int s_buf[BlockAppender<1>::CAP], s_n[2];
    BlockAppender<1> app;
    if (plist) app.init(s_buf, s_n, plist, n_proc, pcap);
    const int t = blockIdx.x * TPB + threadIdx.x;
    const bool proc = t < n && st[t] == 1;
    const int v = proc ? list[t] : 0;
    if (proc) pflag[v] = value;
    if (plist) {
        app.add(proc ? 1 : 0, [&](int) { return v; });
        app.finish();
    }
}
// (a thread per (listed voxel, box position) pair, a fixed grid striding over the pairs: the 27 positions of a box are
// independent, and walked one after the other by one thread they were 27 serial memory latencies)
__global__ __launch_bounds__(TPB) void k_ec_apply(Grid g, const double *__restrict__ rho,
                                                  const int *__restrict__ labels, int8_t *known,
                                                  const int *__restrict__ plist, const int *__restrict__ n_proc,
                                                  unsigned long long *checked, int *new_edges, int *n_new, int new_cap,
                                                  const int8_t *__restrict__ pflag) {
    __shared__ int s_buf[BlockAppender<1>::CAP], s_n[2];
    BlockAppender<1> app;
    app.init(s_buf, s_n, new_edges, n_new, new_cap);
    unsigned int nchk = 0;
    const long long pairs = 27LL * *n_proc;
    const XcdRange xr = xcd_range((int)((pairs + TPB - 1) / TPB));   // (the list is in C order: neighbours' boxes overlap)
    for (int chunk = xr.begin; chunk < xr.end; chunk += xr.step) {   // (uniform per block)
        const long long p = (long long)chunk * TPB + threadIdx.x;
        const int t = (int)(p / 27), j = (int)(p - 27LL * t);
        const bool act = p < pairs;
        int new_edge = -1;
        if (act) {
            const int v = plist[t];
            const int x = v / g.nyz;
            const int r = v - x * g.nyz;
            const int y = r / g.nz, z = r - y * g.nz;
            const int tx = wrapi(x + j / 9 - 1, g.nx), ty = wrapi(y + (j / 3) % 3 - 1, g.ny), tz = wrapi(z + j % 3 - 1, g.nz);
            const int l = lin3(g, tx, ty, tz);
            // the processed voxels in l's own 27-box: how many, and is v the first of them (in C order)?
            int rows[9];
            unsigned int w[9];
            ec_box(g, pflag, tx, ty, tz, rows, w);
            unsigned cnt = 0;
            bool first = true;
#pragma unroll
            for (int k = 0; k < 27; k++) {
                const bool pk = ((w[k / 3] >> (8 * (k % 3))) & 0xffu) != 0;
                cnt += pk;
                first &= !(pk && ec_box_voxel(g, rows, tz, k) < v);
            }
            if (first) {
                // NB no vacuum test on the box voxel (bug-compatible)
                bool is_edge, is_max;
                classify27(g, rho, labels, tx, ty, tz, l, is_edge, is_max);
                if (!is_edge) { known[l] = -1; nchk += cnt; }
                else if (!is_max) { known[l] = -3; new_edge = l; }
            }
        }
        app.add(new_edge >= 0 ? 1 : 0, [&](int) { return new_edge; });   // the new edges, for the -1 ring around them
    }
    app.finish();
    for (int o = 32; o > 0; o >>= 1) nchk += __shfl_down(nchk, o);  // one atomic per wave of the (fixed) grid
    if (threadIdx.x % XB_WAVE == 0 && nchk) atomicAdd(checked, (unsigned long long)nchk);
}
// restore processed edge&max voxels (untouched by their own box) to -2
__global__ void k_ec_restore(int8_t *known, const int *list, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int8_t k = known[list[t]];
    if (k == EC_PROC || k == EC_SKIP) known[list[t]] = -2;
}
// count -3 and turn them into -2 (refinement.py:505-507); 16 voxels per thread and step, one atomic per block
// (count_lo, count_hi): only the -3 voxels with a linear index in that range are counted (a slab counts its own planes)
// Round 4: the sweep reads every flag anyway, so it also LISTS the voxels that are -2 afterwards (new edges and the restored
// edge&maximum voxels) in that range -- the retrace list of the next refinement pass, which round 3 compacted out of the flags
// with another sweep and another host wait.  (list_out null: no list; more entries than list_cap: the count says so.)
__global__ __launch_bounds__(TPB) void k_ec_finish(int8_t *known, long long N, unsigned long long *edges, long long count_lo,
                                                   long long count_hi, int *list_out, int *list_count, int list_cap) {
    __shared__ unsigned int s_cnt;
    __shared__ int s_buf[BlockAppender<16>::CAP], s_n[2];
    BlockAppender<16> app;
    app.init(s_buf, s_n, list_out, list_count, list_cap);
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    unsigned int cnt = 0;
    const long long stride = (long long)gridDim.x * TPB * 16;
    // (the next chunk travels while this one is counted and listed: the list append is a block scan with two barriers)
    uint4 nxt = make_uint4(0, 0, 0, 0);
    {
        const long long b0 = (long long)blockIdx.x * TPB * 16 + (long long)threadIdx.x * 16;
        if (b0 + 16 <= N) nxt = *reinterpret_cast<const uint4 *>(known + b0);
    }
    for (long long base0 = (long long)blockIdx.x * TPB * 16; base0 < N; base0 += stride) {   // (uniform per block)
        const long long base = base0 + (long long)threadIdx.x * 16;
        unsigned hits = 0;   // bit k: voxel base + k is -2 now and inside the range
        uint4 w = nxt;
        if (base + stride + 16 <= N) nxt = *reinterpret_cast<const uint4 *>(known + base + stride);
        if (base + 16 <= N) {
            int8_t *b = reinterpret_cast<int8_t *>(&w);
            unsigned int c = 0;
            if ((w.x | w.y | w.z | w.w) & 0x80808080u)   // (no negative flag among the 16: nothing to count or list -- most of the grid)
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const bool in = (base + k >= count_lo) & (base + k < count_hi);
                if (b[k] == -3) { b[k] = -2; c++; cnt += in; }
                hits |= (unsigned)((b[k] == -2) & in) << k;
            }
            if (c) *reinterpret_cast<uint4 *>(known + base) = w;
        } else if (base < N) {
            for (long long k = base; k < N; k++) {
                const bool in = k >= count_lo && k < count_hi;
                if (known[k] == -3) { known[k] = -2; cnt += in; }
                hits |= (unsigned)((known[k] == -2) & in) << (int)(k - base);
            }
        }
        if (list_out) {
            unsigned m = hits;
            app.add(__popc(hits), [&](int) { const int k = __ffs(m) - 1; m &= m - 1; return (int)(base + k); });
        }
    }
    if (list_out) app.finish();
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if (threadIdx.x % XB_WAVE == 0 && cnt) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd(edges, (unsigned long long)s_cnt);
}
