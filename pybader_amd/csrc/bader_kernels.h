// bader_kernels.h -- device code of libbader_hip.so (gfx950 / CDNA4, wave64).
//
// All float64 arithmetic restates the reference's expression trees exactly (separate multiply and
// add -- the library is built with -ffp-contract=off --, true division, truncating casts); see
// the citations on each function.  No MFMA: this path is a memory/latency-bound stencil + gather.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct Grid {
    int nx, ny, nz;   // shape, C-order [x][y][z]
    int nyz;          // ny*nz (elements per x-plane)
    int x0, x1;       // owned slab: planes [x0, x1)
    int vx0, vlen;    // planes whose labels/known are valid: [vx0, vx0+vlen) modulo nx
    double T[9];      // T_grad, row-major (interface.py:285-290)
    double dist[27];  // dist_mat [3][3][3], index 2 == -1 (interface.py:242-259)
};

#define XB_INT_MAX 0x7fffffff
#define XB_WAVE 64

__device__ __forceinline__ int wrapi(int v, int n) { return v < 0 ? v + n : (v >= n ? v - n : v); }
__device__ __forceinline__ int lin3(const Grid &g, int x, int y, int z) { return (x * g.ny + y) * g.nz + z; }
__device__ __forceinline__ bool plane_valid(const Grid &g, int x) {
    int d = x - g.vx0;
    if (d < 0) d += g.nx;
    return d < g.vlen;
}
// round-half-away-from-zero by a truncating cast: methods.py:347-350 / refinement.py:138-141
__device__ __forceinline__ int rha(double x) { return x > 0 ? (int)(x + .5) : (int)(x - .5); }

// One neargrid step (refinement.py:89-154, the strict tie test `<  >` of line 111) from voxel
// (px,py,pz) whose density is c, carrying the remainder dr.  Returns true when
// max_grad < 1E-14 (refinement.py:132-134: no move, dr untouched); else q/dr hold the new state.
__device__ __forceinline__ bool ng_step(const double *__restrict__ rho, const Grid &g, int px, int py, int pz,
                                        int lp, double c, double &dr0, double &dr1, double &dr2, int &qx,
                                        int &qy, int &qz) {
    const int xp = wrapi(px + 1, g.nx), xm = wrapi(px - 1, g.nx);
    const int yp = wrapi(py + 1, g.ny), ym = wrapi(py - 1, g.ny);
    const int zp = wrapi(pz + 1, g.nz), zm = wrapi(pz - 1, g.nz);
    const double hx = rho[lp + (xp - px) * g.nyz], lx = rho[lp + (xm - px) * g.nyz];
    const double hy = rho[lp + (yp - py) * g.nz], ly = rho[lp + (ym - py) * g.nz];
    const double hz = rho[lp + (zp - pz)], lz = rho[lp + (zm - pz)];
    // refinement.py:111-114: zero when p is a strict maximum along the axis, else central difference
    const double g0 = (hx < c && c > lx) ? 0. : (hx - lx) / 2.;
    const double g1 = (hy < c && c > ly) ? 0. : (hy - ly) / 2.;
    const double g2 = (hz < c && c > lz) ? 0. : (hz - lz) / 2.;
    // refinement.py:123-130: grad_dir = T_grad . grad with (a+b)+c association
    double d0 = ((g.T[0] * g0) + (g.T[1] * g1)) + (g.T[2] * g2);
    double d1 = ((g.T[3] * g0) + (g.T[4] * g1)) + (g.T[5] * g2);
    double d2 = ((g.T[6] * g0) + (g.T[7] * g1)) + (g.T[8] * g2);
    double mg = 0.;
    if (d0 > mg) mg = d0; else if (-d0 > mg) mg = -d0;
    if (d1 > mg) mg = d1; else if (-d1 > mg) mg = -d1;
    if (d2 > mg) mg = d2; else if (-d2 > mg) mg = -d2;
    if (mg < 1E-14) { qx = px; qy = py; qz = pz; return true; }
    // refinement.py:136-154
    d0 /= mg; d1 /= mg; d2 /= mg;
    int ig, id;
    ig = rha(d0); qx = px + ig; dr0 += d0 - (double)ig; id = rha(dr0); qx += id; dr0 -= (double)id;
    ig = rha(d1); qy = py + ig; dr1 += d1 - (double)ig; id = rha(dr1); qy += id; dr1 -= (double)id;
    ig = rha(d2); qz = pz + ig; dr2 += d2 - (double)ig; id = rha(dr2); qz += id; dr2 -= (double)id;
    if (qx >= g.nx) qx -= g.nx; else if (qx < 0) qx += g.nx;
    if (qy >= g.ny) qy -= g.ny; else if (qy < 0) qy += g.ny;
    if (qz >= g.nz) qz -= g.nz; else if (qz < 0) qz += g.nz;
    return false;
}

// One ongrid step (methods.py:84-117; refinement.py:204-235): the best of the 27 neighbours by
// (rho(n)-rho(p))*dist_mat + rho(p), strict '>', first wins in (ix,iy,iz) ascending order.
__device__ __forceinline__ void og_step(const double *__restrict__ rho, const Grid &g, int px, int py, int pz,
                                        double ctr, int &qx, int &qy, int &qz) {
    double max_val = ctr;
    qx = px; qy = py; qz = pz;
#pragma unroll
    for (int ix = -1; ix < 2; ix++) {
        const int tx = wrapi(px + ix, g.nx);
#pragma unroll
        for (int iy = -1; iy < 2; iy++) {
            const int ty = wrapi(py + iy, g.ny);
#pragma unroll
            for (int iz = -1; iz < 2; iz++) {
                const int tz = wrapi(pz + iz, g.nz);
                double v = rho[lin3(g, tx, ty, tz)];
                v = (v - ctr) * g.dist[((ix + 3) % 3) * 9 + ((iy + 3) % 3) * 3 + ((iz + 3) % 3)];
                v += ctr;
                if (v > max_val) { max_val = v; qx = tx; qy = ty; qz = tz; }
            }
        }
    }
}

// The last K voxels of a trajectory plus the largest density among the older ones.  A voxel q is
// on the path iff it is in the window, or (only possible when rho(q) <= m_old) among the older
// ones -- the second case is not decided here but handed to the exact slow kernel ("overflow").
// This keeps the reference's "already been here this path" test (refinement.py:200) exact without
// a per-thread path array: trajectories ascend, so rho(q) > m_old for all but pathological steps.
template <int K>
struct PathWindow {
    int idx[K];
    double val[K];
    double m_old;
    __device__ __forceinline__ void init(int l, double c) {
#pragma unroll
        for (int k = 0; k < K; k++) { idx[k] = -1; val[k] = -1.; }
        idx[0] = l; val[0] = c;
        m_old = -1.7976931348623157e308;
    }
    __device__ __forceinline__ bool contains(int l) const {
        bool f = false;
#pragma unroll
        for (int k = 0; k < K; k++) f |= (idx[k] == l);
        return f;
    }
    __device__ __forceinline__ void push(int l, double c) {
        if (idx[K - 1] >= 0 && val[K - 1] > m_old) m_old = val[K - 1];
#pragma unroll
        for (int k = K - 1; k > 0; k--) { idx[k] = idx[k - 1]; val[k] = val[k - 1]; }
        idx[0] = l; val[0] = c;
    }
};

enum { TR_STEP = 0, TR_NEED_OG = 1, TR_DONE = 2 };
