// k_ongrid.h -- device kernels of libbader_hip.so: ongrid assignment (pointer + jumping), numbering and relabel helpers.
// Included by bader_hip.hip (one translation unit); see bader_kernels.h for the common device code.
#pragma once

// ---------------------------------------------------------------------------------------------
// ongrid assignment (methods.py:15-219).  The ascent is memoryless, so the sequential path
// compression of the reference equals: best-neighbour pointer per voxel, then pointer jumping.
// ---------------------------------------------------------------------------------------------
// A chain that steps onto a vacuum voxel inherits -1 (methods.py:166-168).  In-place and
// asynchronous: any value read is an ancestor of the root, so progress is monotone.
#define OG_HOPS 3
__global__ __launch_bounds__(TPB) void k_og_jump(Grid g, int *labels, int *not_done) {
    const long long N = (long long)g.nx * g.nyz;
    const long long vv = (long long)blockIdx.x * TPB + threadIdx.x;
    if (vv >= N) return;
    const int v = (int)vv;
    int p = labels[v];
    if (p < 0 || p == v) return;
    int q = labels[p];
    if (q == p) return;  // parent is a root
#pragma unroll
    for (int hop = 0; hop < OG_HOPS && q >= 0; hop++) {  // a few more hops per sweep
        const int q2 = labels[q];
        if (q2 == q) break;
        q = q2;
    }
    labels[v] = q;
    if (q >= 0) *not_done = 1;
}
// ---------------------------------------------------------------------------------------------
// Round 4: the ongrid pass in the shape of k_brick_masks.  One sweep over the density gives every voxel its best-neighbour
// pointer (methods.py:84-117: the first neighbour in (ix, iy, iz) order with the largest distance-weighted value, strict '>')
// and every 8^3 brick what the region growth of the neargrid path wants (k_fused.h): the neighbour bricks its pointers enter,
// its number of maxima, its single maximum and its potential.  The trapping regions of the pointer field are then grown,
// walked and numbered by the SAME device-driven machinery as the neargrid ones -- one host wait per assignment, no seed
// cubes, no cap on the number of maxima (round 1-3: closed cubes around at most 1023 maxima, grown brick by brick with a host
// wait every second launch).  Staging as in k_brick_masks (bm_stage: scalar row addresses, z wrap once per lane), the tile in
// the same padded rows; a thread walks one (y, z) column along x with a rolling 3 x 3 x 3 window: 9 LDS reads per voxel
// instead of 27.  PART: a brick the grid cuts counts its voxels inside the grid; a pointer is one voxel long, so a brick of
// width one needs no special care here.
template <typename GT, bool PART>
__global__ __launch_bounds__(TPB) void k_og_masks(GT g, const double *__restrict__ rho, int *__restrict__ labels, int small,
                                                  int has_vacuum, int *__restrict__ bmask, int *__restrict__ bmaxv,
                                                  int *__restrict__ bpot) {
    __shared__ double tile[GT_X + 2][GT_Y + 2][BM_ROW];
    __shared__ int s_mask[GT_Z / 8], s_cnt[GT_Z / 8], s_mv[GT_Z / 8], s_pot[GT_Z / 8];
    __shared__ unsigned s_bmax;   // (bm_stage's mirror bound: unused here)
    const int x0 = blockIdx.z * GT_X, y0 = blockIdx.y * GT_Y, z0 = blockIdx.x * GT_Z;
    if (threadIdx.x < GT_Z / 8) { s_mask[threadIdx.x] = 0; s_cnt[threadIdx.x] = 0; s_mv[threadIdx.x] = -1; s_pot[threadIdx.x] = -2147483647 - 1; }
    __syncthreads();
    if (small & 1) bm_stage<true, PART>(g, rho, tile, &s_bmax, x0, y0, z0, 0);
    else bm_stage<false, PART>(g, rho, tile, &s_bmax, x0, y0, z0, 0);
    __syncthreads();
    const int tz = threadIdx.x & (GT_Z - 1), ty = threadIdx.x / GT_Z, zz = tz & 7;
    const int y = y0 + ty, z = z0 + tz;
    const bool col_in = y < g.ny && z < g.nz;
    // the valid voxels of this lane's brick along each axis (PART)
    const int wx = PART ? min(GT_X, g.nx - x0) : GT_X, wy = PART ? min(8, g.ny - y0) : 8, wz = PART ? min(8, g.nz - (z0 + (tz & ~7))) : 8;
    double a[3][3][3];
#pragma unroll
    for (int iy = 0; iy < 3; iy++)
#pragma unroll
        for (int iz = 0; iz < 3; iz++) {
            a[1][iy][iz] = tile[0][ty + iy][tz + iz];
            a[2][iy][iz] = tile[1][ty + iy][tz + iz];
        }
    int mine = 0;
    double cmax = -1.7976931348623157e308;
#pragma unroll
    for (int K = 0; K < GT_X; K++) {
#pragma unroll
        for (int iy = 0; iy < 3; iy++)
#pragma unroll
            for (int iz = 0; iz < 3; iz++) {
                a[0][iy][iz] = a[1][iy][iz];
                a[1][iy][iz] = a[2][iy][iz];
                a[2][iy][iz] = tile[K + 2][ty + iy][tz + iz];
            }
        const int x = x0 + K;
        const bool in = col_in && x < g.nx;
        const int v = (x * g.ny + y) * g.nz + z;
        const bool vac = in && has_vacuum && labels[v] == -1;
        const double c = a[1][1][1];
        double max_val = c;
        int og = XB_OG_SELF;
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
        if (in && !vac) {
            const int ox = og / 9 - 1, oy = (og / 3) % 3 - 1, oz = og % 3 - 1;
            const int qx = wrapi(x + ox, g.nx), qy = wrapi(y + oy, g.ny), qz = wrapi(z + oz, g.nz);
            labels[v] = (qx * g.ny + qy) * g.nz + qz;
            cmax = max_raw(cmax, c);
            if (og == XB_OG_SELF) {   // a maximum of the pointer field
                atomicAdd(&s_cnt[tz >> 3], 1);
                s_mv[tz >> 3] = v;
            }
            // the neighbour brick the pointer enters (position + offset against the brick's valid width)
            const int pa = K + ox, pb = ty + oy, pc = zz + oz;
            const int k0 = pa < 0 ? 0 : (pa >= wx ? 2 : 1), k1 = pb < 0 ? 0 : (pb >= wy ? 2 : 1), k2 = pc < 0 ? 0 : (pc >= wz ? 2 : 1);
            mine |= 1 << (k0 * 9 + k1 * 3 + k2);
        }
    }
    atomicOr(&s_mask[tz >> 3], mine);
    if (col_in) {
        const int fi = __float_as_int((float)cmax);
        atomicMax(&s_pot[tz >> 3], fi >= 0 ? fi : fi ^ 0x7fffffff);
    }
    __syncthreads();
    if (threadIdx.x < GT_Z / 8 && z0 + threadIdx.x * 8 < g.nz) {
        const int nb1 = (g.ny + 7) >> 3, nb2 = (g.nz + 7) >> 3;
        const int b = ((x0 >> 3) * nb1 + (y0 >> 3)) * nb2 + (z0 >> 3) + threadIdx.x, n = s_cnt[threadIdx.x];
        bmask[b] = (s_mask[threadIdx.x] & 0x7ffdfff) | (n >= 1 ? 1 << 27 : 0) | (n >= 2 ? 1 << 28 : 0);
        bmaxv[b] = n == 1 ? s_mv[threadIdx.x] : -1;
        bpot[b] = s_pot[threadIdx.x];
    }
}
// With trapping regions only the voxels of the listed (uncertain) bricks chase their pointers, and only until the chain enters
// a certain brick or reaches a root.  labels[] holds successor indices on entry; a walker overwrites its own entry with the
// root it found -- also an ancestor, so a chain that reads it mid-way still ends at the same root.  The list length lives on
// the device: a fixed grid of one-wave workgroups strides over the eighths of the listed bricks (lanes beyond the grid,
// in a brick the grid cuts, stay idle).
__global__ __launch_bounds__(XB_WAVE) void k_og_walk_dev(GridL g, const int *__restrict__ box_max, const int *__restrict__ blab,
                                                         int nb1, int nb2, const int *__restrict__ walk, int *fs, int *labels,
                                                         int *first, int *max_list, int max_cap, int maxsteps) {
    if (fs[FS_GROW_RETRY]) return;
    const int n_items = fs[FS_N_WALK] * 8, lane = threadIdx.x;
    const double inv_nyz = 1.0 / (double)g.nyz, inv_nz = 1.0 / (double)g.nz;
    const XcdRange xr = xcd_range(n_items, 8);   // (whole bricks per XCD: the eighths of a brick and its neighbours read the same pointers)
    for (int item = xr.begin; item < xr.end; item += xr.step) {
        // (the eighths are 4 x 2 x 8 voxels as in the neargrid trace, brick_sub_voxel: whole rows of 8 pointers)
        const int b = walk[item >> 3], sub = item & 7;
        const int x = (b / (nb1 * nb2)) * 8 + ((sub >> 2) << 2) + (lane >> 4);
        const int y = ((b / nb2) % nb1) * 8 + ((sub & 3) << 1) + ((lane >> 3) & 1);
        const int z = (b % nb2) * 8 + (lane & 7);
        const bool valid = x < g.nx && y < g.ny && z < g.nz;
        const int v = valid ? (x * g.ny + y) * g.nz + z : 0;
        int cur = v, p = valid ? labels[v] : -1, result = -1;
        bool done = !valid;
        for (int s = 0; s <= maxsteps && !done; s++) {
            if (p < 0) { done = true; break; }                    // vacuum: the chain inherits -1 (methods.py:166-168)
            if (p == cur) { result = p; done = true; break; }     // a root
            const int pn = labels[p];   // the next pointer, in flight together with the brick label of p
            // p -> (px, py, pz): two multiplications by reciprocals and a correction step (exact for every int32 index)
            int px = (int)((double)p * inv_nyz);
            int r = (int)((unsigned)p - (unsigned)px * (unsigned)g.nyz);
            if (r < 0) { px--; r += g.nyz; } else if (r >= g.nyz) { px++; r -= g.nyz; }
            int py = (int)((double)r * inv_nz);
            int pz = r - py * g.nz;
            if (pz < 0) { py--; pz += g.nz; } else if (pz >= g.nz) { py++; pz -= g.nz; }
            const int bl = blab[((px >> 3) * nb1 + (py >> 3)) * nb2 + (pz >> 3)];
            if (bl > 0) { result = box_max[bl - 1]; done = true; break; }
            cur = p;
            p = pn;
        }
        if (!done) atomicOr(&fs[FS_ERR], 1);  // the pointer field is acyclic: cannot happen, reported loudly if it does
        if (valid) labels[v] = result;
        note_maximum_wave(valid && result >= 0, result, v, first, max_list, &fs[FS_N_MAX], max_cap);
    }
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
// first[] back to "nobody arrived" for the listed maxima: n of them, or *n_dev (and then only when `gate` is set: the
// device-side numbering succeeded)
__global__ void k_reset_first(int *first, const int *max_list, int n, const int *n_dev, const int *gate) {
    if (gate && !*gate) return;
    if (n_dev) n = *n_dev;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) first[max_list[i]] = XB_INT_MAX;
}
// labels[v] (maximum index) -> rank stored in first[maximum]
__global__ __launch_bounds__(TPB) void k_relabel(Grid g, int *labels, const int *__restrict__ rank, const int *gate) {
    const long long vbeg = (long long)g.x0 * g.nyz, vend = (long long)g.x1 * g.nyz;
    const long long v = vbeg + (long long)blockIdx.x * TPB + threadIdx.x;
    if (v >= vend || (gate && !*gate)) return;
    const int m = labels[v];
    if (m >= 0) labels[v] = rank[m];
}
