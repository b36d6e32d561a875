// k_common.h -- device kernels of libbader_hip.so: shared helpers, workload generator, vacuum sweep, numbering notes.
// Included by bader_hip.hip (one translation unit); see bader_kernels.h for the common device code.
#pragma once

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
// whole grid; charge/volume partial sums over the owned slab only (block reduce + one atomic pair per workgroup).
// Round 6: a fixed grid of workgroups strides over the voxels -- with one workgroup per 256 voxels a density WITH vacuum
// issued 2 x 524 288 atomics on two addresses at 512^3, 6.3 ms of an 11.8 ms step (device-scope atomics on one word serialise
// at ~88 per microsecond); a step without vacuum never noticed (no vacuum voxel, no atomic).
__global__ __launch_bounds__(TPB) void k_vacuum_assign(Grid g, const double *__restrict__ rho,
                                                       int *__restrict__ labels, double tol, double *sum_rho,
                                                       unsigned long long *count) {
    const long long N = (long long)g.nx * g.nyz;
    const long long lo = (long long)g.x0 * g.nyz, hi = (long long)g.x1 * g.nyz;   // the owned slab's voxels
    double s = 0.;
    unsigned long long n = 0;
    for (long long v = (long long)blockIdx.x * TPB + threadIdx.x; v < N; v += (long long)gridDim.x * TPB) {
        const double r = rho[v];
        const bool vac = r <= tol;  // NaN tol (vacuum_tol=None, interface.py:459) => never
        labels[v] = vac ? -1 : 0;
        if (vac && v >= lo && v < hi) { s += r; n++; }
    }
    __shared__ double sh[TPB / XB_WAVE];
    __shared__ unsigned long long shn[TPB / XB_WAVE];
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o); n += __shfl_down(n, o); }
    const int w = threadIdx.x / XB_WAVE, l = threadIdx.x % XB_WAVE;
    if (l == 0) { sh[w] = s; shn[w] = n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.;
        unsigned long long m = 0;
        for (int q = 0; q < TPB / XB_WAVE; q++) { t += sh[q]; m += shn[q]; }
        if (m) { atomicAdd(sum_rho, t); atomicAdd(count, m); }
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

// Appending to ONE list from a whole grid of workgroups.  Device-scope atomics on one counter serialise at ~88 per microsecond
// (k_fused.h), so a reservation per wave is too many once a list has 10^5 waves with something to add (measured: 1.3 ms for
// 2.4 M entries).  A workgroup stages its entries in LDS instead and reserves space for a few thousand at a time.  Every thread
// of the workgroup calls add() in the same iteration (block-uniform control flow) and finish() once at the end; the order of
// the entries in the list is arbitrary.  MAXPER: the most entries one thread adds per call.
template <int MAXPER>
struct BlockAppender {
    static constexpr int CAP = 4096 + TPB * MAXPER;
    int *buf;      // LDS, CAP ints
    int *s_n;      // LDS: entries staged; [1]: scratch for the broadcast of a reservation
    int *out, *out_count, out_cap;
    __device__ __forceinline__ void init(int *lds_buf, int *lds_n, int *o, int *oc, int ocap) {
        buf = lds_buf; s_n = lds_n; out = o; out_count = oc; out_cap = ocap;
        if (threadIdx.x == 0) s_n[0] = 0;
        __syncthreads();
    }
    __device__ __forceinline__ void flush() {   // (block uniform)
        const int n = s_n[0];
        __syncthreads();
        if (threadIdx.x == 0) { s_n[1] = n ? atomicAdd(out_count, n) : 0; s_n[0] = 0; }
        __syncthreads();
        const int base = s_n[1];
        for (int i = threadIdx.x; i < n; i += TPB)
            if (base + i < out_cap) out[base + i] = buf[i];
        __syncthreads();
    }
    // `cnt` entries of this thread, handed over by get(k), k < cnt
    template <typename Get>
    __device__ __forceinline__ void add(int cnt, Get get) {
        int total;
        const int off = block_scan_excl(cnt, total);
        if (total == 0) return;                       // (uniform)
        if (s_n[0] + total > CAP) flush();
        const int at = s_n[0] + off;
        for (int k = 0; k < cnt; k++) buf[at + k] = get(k);
        __syncthreads();
        if (threadIdx.x == 0) s_n[0] += total;
        __syncthreads();
    }
    __device__ __forceinline__ void finish() { flush(); }
};
