// k_sums.h -- device kernels of libbader_hip.so: charge sums, atom map, surface distance, masks, dtype widening.
// Included by bader_hip.hip (one translation unit); see bader_kernels.h for the common device code.
#pragma once

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
// does any label equal `value`?  (xb_upload_labels: "no vacuum voxel" unlocks the region fast paths)
__global__ __launch_bounds__(TPB) void k_any_equal(const int *__restrict__ a, long long n, int value, int *flag) {
    bool hit = false;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) hit |= (a[i] == value);
    if (__any(hit) && threadIdx.x % XB_WAVE == 0) *flag = 1;
}
// 16 bytes of narrowed labels per thread: what writes a page-locked host array over the bus (one 16-byte store per lane, a
// kilobyte per wave store); n16 = the number of whole 16-byte groups, the caller's plain k_narrow does the few labels behind them
template <typename T>
__global__ __launch_bounds__(TPB) void k_narrow_vec(const int *__restrict__ in, T *__restrict__ out, long long n16) {
    constexpr int PER = 16 / (int)sizeof(T);
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n16) return;
    const int *src = in + i * PER;
    T v[PER];
#pragma unroll
    for (int k = 0; k < PER; k += 4) {
        const int4 q = *reinterpret_cast<const int4 *>(src + k);
        v[k] = (T)q.x; v[k + 1] = (T)q.y; v[k + 2] = (T)q.z; v[k + 3] = (T)q.w;
    }
    *reinterpret_cast<uint4 *>(out + i * PER) = *reinterpret_cast<const uint4 *>(v);
}
template <typename T>
__global__ void k_narrow(const int *__restrict__ in, T *__restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (T)in[i];
}

// utils.atom_assign (utils.py:185-232): nearest atom of every Bader maximum over the 27 periodic images,
// squared distances compared with strict '<' in the reference's loop order (atoms, then x, y, z images), one
// thread per maximum.  The reference's `pbc` vector persists across maxima (utils.py:199, 206-208): the
// starting distance of maximum i > 0 uses the image vector the previous maximum's loops ended on, (+1,+1,+1).
__global__ void k_atom_assign(const double *__restrict__ b_max, int n_max, const double *__restrict__ atoms, int n_atoms,
                              const double *__restrict__ lattice, long long *atom_out, double *dist_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_max) return;
    double pbc[3] = {0., 0., 0.};
    if (i > 0)
        for (int k = 0; k < 3; k++) pbc[k] = (lattice[k] * 1 + lattice[3 + k] * 1) + lattice[6 + k] * 1;
    const double b0 = b_max[3 * i], b1 = b_max[3 * i + 1], b2 = b_max[3 * i + 2];
    double e0 = b0 - (atoms[0] + pbc[0]), e1 = b1 - (atoms[1] + pbc[1]), e2 = b2 - (atoms[2] + pbc[2]);
    double best = (e0 * e0 + e1 * e1) + e2 * e2;
    long long who = 0;
    for (int j = 0; j < n_atoms; j++) {
        const double a0 = atoms[3 * j], a1 = atoms[3 * j + 1], a2 = atoms[3 * j + 2];
        for (int x = -1; x < 2; x++)
            for (int y = -1; y < 2; y++)
                for (int z = -1; z < 2; z++) {
                    for (int k = 0; k < 3; k++) pbc[k] = (lattice[k] * x + lattice[3 + k] * y) + lattice[6 + k] * z;
                    e0 = b0 - (a0 + pbc[0]); e1 = b1 - (a1 + pbc[1]); e2 = b2 - (a2 + pbc[2]);
                    const double d = (e0 * e0 + e1 * e1) + e2 * e2;
                    if (d < best) { best = d; who = j; }
                }
    }
    atom_out[i] = who;
    dist_out[i] = sqrt(best);
}
