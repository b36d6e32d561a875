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
// With trapping regions (see k_table.h) only the voxels of the listed (uncertain) bricks chase their
// pointers, and only until the chain enters a certain brick or reaches a root.  labels[] holds successor
// indices on entry; a walker overwrites its own entry with the root it found -- also an ancestor, so a chain
// that reads it mid-way still ends at the same root.  One wave per 4x4x4 eighth of a brick.
__global__ __launch_bounds__(XB_WAVE) void k_og_walk(GridL g, const int *__restrict__ box_max, const int *__restrict__ blab,
                                                     int nb1, int nb2, const int *__restrict__ walk, int n_walk, int *labels,
                                                     int *first, int *max_list, int *max_count, int max_cap, int maxsteps,
                                                     int *err) {
    const int e = blockIdx.x >> 3, sub = blockIdx.x & 7, lane = threadIdx.x;
    if (e >= n_walk) return;
    const int b = walk[e];
    const int x = (b / (nb1 * nb2)) * 8 + ((sub >> 2) << 2) + (lane >> 4);
    const int y = ((b / nb2) % nb1) * 8 + (((sub >> 1) & 1) << 2) + ((lane >> 2) & 3);
    const int z = (b % nb2) * 8 + ((sub & 1) << 2) + (lane & 3);
    const int v = (x * g.ny + y) * g.nz + z;
    int cur = v, p = labels[v], result = -1;
    bool done = false;
    for (int s = 0; s <= maxsteps && !done; s++) {
        if (p < 0) { done = true; break; }       // vacuum (not reached here: regions are used without vacuum)
        if (p == cur) { result = p; done = true; break; }     // a root
        const int px = p / g.nyz, r = p - px * g.nyz;
        const int bl = blab[((px >> 3) * nb1 + ((r / g.nz) >> 3)) * nb2 + ((r % g.nz) >> 3)];
        if (bl > 0) { result = box_max[bl - 1]; done = true; break; }
        cur = p;
        p = labels[cur];
    }
    if (!done) atomicExch(err, 1);  // the pointer field is acyclic: cannot happen, reported loudly if it does
    labels[v] = result;
    note_maximum_wave(result >= 0, result, v, first, max_list, max_count, max_cap);
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
// the same for a device-side count (and only when `gate` is set: the device-side numbering succeeded)
__global__ void k_reset_first_dev(int *first, const int *max_list, const int *n_dev, const int *gate) {
    if (!*gate) return;
    const int n = *n_dev;
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
