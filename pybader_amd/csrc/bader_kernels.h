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
    int wx0, wlen;    // planes whose gradient-field table records exist: [wx0, wx0+wlen) modulo nx
    int wbase, ntot;  // the table holds ONLY those planes: record of voxel l at slot (l - wbase) mod ntot (rec_slot); ntot = nx*ny*nz
    int main_ties;    // 1: the axis tie test of methods.py:324 (`<= >=`, the assignment), 0: refinement.py:111 (`< >`)
    double T[9];      // T_grad, row-major (interface.py:285-290)
    double dist[27];  // dist_mat [3][3][3], index 2 == -1 (interface.py:242-259)
};

// Grid for the tiled field kernels when dist_mat is symmetric (dist(-o) == dist(o), which it is for every
// lattice: the entries are 1/|o . voxel_lattice|): 14 instead of 27 distances, so that T + dist fit the
// SGPR file without spilling (the 27-entry version costs ~30 v_readlane per voxel in k_grad_field).
// dsym[k], k = min(idx, 26 - idx) with idx = ix*9 + iy*3 + iz over offsets ix,iy,iz in 0..2 (== -1..1).
struct GridS {
    int nx, ny, nz, nyz;
    int x0, x1, vx0, vlen;
    int wx0, wlen;
    int wbase, ntot;
    int main_ties;
    double T[9];
    double dsym[14];
};
__device__ __forceinline__ double dist_at(const Grid &g, int ix, int iy, int iz) {
    return g.dist[((ix + 2) % 3) * 9 + ((iy + 2) % 3) * 3 + ((iz + 2) % 3)];
}
__device__ __forceinline__ double dist_at(const GridS &g, int ix, int iy, int iz) {
    const int idx = ix * 9 + iy * 3 + iz;
    return g.dsym[idx < 26 - idx ? idx : 26 - idx];
}

// What the trace kernels need of Grid (keeps their SGPR count low: more waves per SIMD).
struct GridL {
    int nx, ny, nz, nyz;
    int x0, x1, vx0, vlen;
    int wx0, wlen;    // table window (see Grid)
    int wbase, ntot;  // ... and where its records lie (rec_slot)
    int use24;        // nx*ny < 2^24 and nz < 2^24: linear indices via 24-bit multiplies
    int main_ties;    // see Grid
};

#define XB_INT_MAX 0x7fffffff
#define XB_WAVE 64

// A work list dealt out by XCD: workgroups go to the eight XCDs in turn (blockIdx.x % 8), each XCD has an L2 of its own, and the
// lists of this library are in spatial order (tiles, Morton bricks) -- so XCD k takes the k-th contiguous part of the list and
// neighbours, which read the same lines, meet in ONE L2.  The chunks [0, n_chunks) of a list: this workgroup takes
// begin, begin + step, ... below end (`whole`: parts are multiples of it).  A grid that is not a multiple of 8 deals in turn.
struct XcdRange { int begin, end, step; };
__device__ __forceinline__ XcdRange xcd_range(int n_chunks, int whole = 1) {
    const int n_xcd = (gridDim.x % 8 == 0) ? 8 : 1, k = (int)(blockIdx.x % n_xcd);
    int per = (n_chunks + n_xcd - 1) / n_xcd;
    per = (per + whole - 1) / whole * whole;
    XcdRange r;
    r.begin = k * per + (int)(blockIdx.x / n_xcd);
    r.end = min(n_chunks, (k + 1) * per);
    r.step = (int)(gridDim.x / n_xcd);
    return r;
}
__device__ __forceinline__ int wrapi(int v, int n) { return v < 0 ? v + n : (v >= n ? v - n : v); }
__device__ __forceinline__ int lin3(const Grid &g, int x, int y, int z) { return (x * g.ny + y) * g.nz + z; }
template <typename GT>
__device__ __forceinline__ bool plane_valid(const GT &g, int x) {
    int d = x - g.vx0;
    if (d < 0) d += g.nx;
    return d < g.vlen;
}
template <typename GT>
__device__ __forceinline__ bool plane_in_window(const GT &g, int x) {
    int d = x - g.wx0;
    if (d < 0) d += g.nx;
    return d < g.wlen;
}
// round-half-away-from-zero by a truncating cast: methods.py:347-350 / refinement.py:138-141
__device__ __forceinline__ int rha(double x) { return x > 0 ? (int)(x + .5) : (int)(x - .5); }

// Direction part of one neargrid step (refinement.py:89-137, the strict tie test `<  >` of line
// 111): the normalised gradient direction grad_dir / max_grad at voxel p.  It depends on rho only
// (not on the carried remainder dr), so it is precomputed once per voxel by k_grad_field.
// Returns true when max_grad < 1E-14 (refinement.py:132-134: the trajectory does not move).
// The two callers of the reference differ in ONE comparison: methods.neargrid (the assignment) zeroes an axis
// when `hi <= c >= lo` (methods.py:324), refinement.neargrid when `hi < c > lo` (refinement.py:111); they
// only disagree on exact ties (hi == c or lo == c with c >= both), which real CHG files (5 significant
// digits) are full of.  `tie` reports that this voxel has such an axis (its two records differ).
__device__ __forceinline__ bool axis_flat(int main_ties, double hi, double c, double lo) {
    return main_ties ? (hi <= c && c >= lo) : (hi < c && c > lo);
}
__device__ __forceinline__ bool axis_tie(double hi, double c, double lo) {
    return (hi <= c && c >= lo) && !(hi < c && c > lo);
}
template <typename GT>
__device__ __forceinline__ bool ng_dir_vals(const GT &g, double c, double hx, double lx, double hy, double ly,
                                            double hz, double lz, double &d0, double &d1, double &d2) {
    // methods.py:324-327 / refinement.py:111-114: zero when p is a maximum along the axis, else central difference
    const double g0 = axis_flat(g.main_ties, hx, c, lx) ? 0. : (hx - lx) / 2.;
    const double g1 = axis_flat(g.main_ties, hy, c, ly) ? 0. : (hy - ly) / 2.;
    const double g2 = axis_flat(g.main_ties, hz, c, lz) ? 0. : (hz - lz) / 2.;
    // refinement.py:123-130: grad_dir = T_grad . grad with (a+b)+c association
    d0 = ((g.T[0] * g0) + (g.T[1] * g1)) + (g.T[2] * g2);
    d1 = ((g.T[3] * g0) + (g.T[4] * g1)) + (g.T[5] * g2);
    d2 = ((g.T[6] * g0) + (g.T[7] * g1)) + (g.T[8] * g2);
    // refinement.py:131-134: max_grad = the largest |component| (the reference's if/elif chain from 0; no NaNs)
    const double mg = fmax(fmax(fabs(d0), fabs(d1)), fabs(d2));
    if (mg < 1E-14) return true;
    d0 /= mg; d1 /= mg; d2 /= mg;  // refinement.py:137, true division
    return false;
}
__device__ __forceinline__ bool ng_dir(const double *__restrict__ rho, const Grid &g, int px, int py, int pz,
                                       int lp, double c, double &d0, double &d1, double &d2) {
    const int xp = wrapi(px + 1, g.nx), xm = wrapi(px - 1, g.nx);
    const int yp = wrapi(py + 1, g.ny), ym = wrapi(py - 1, g.ny);
    const int zp = wrapi(pz + 1, g.nz), zm = wrapi(pz - 1, g.nz);
    const double hx = rho[lp + (xp - px) * g.nyz], lx = rho[lp + (xm - px) * g.nyz];
    const double hy = rho[lp + (yp - py) * g.nz], ly = rho[lp + (ym - py) * g.nz];
    const double hz = rho[lp + (zp - pz)], lz = rho[lp + (zm - pz)];
    return ng_dir_vals(g, c, hx, lx, hy, ly, hz, lz, d0, d1, d2);
}

// Move part (refinement.py:138-154): step = rha(dir), dr += dir - step, corr = rha(dr),
// q = p + step + corr (wrapped), dr -= corr.  rha(x) == (int)(x + copysign(.5, x)) for every x.
__device__ __forceinline__ int rha_cs(double x) { return (int)(x + __builtin_copysign(0.5, x)); }
__device__ __forceinline__ void ng_move(const Grid &g, int px, int py, int pz, double d0, double d1, double d2,
                                        double &dr0, double &dr1, double &dr2, int &qx, int &qy, int &qz) {
    int ig, id;
    ig = rha_cs(d0); dr0 += d0 - (double)ig; id = rha_cs(dr0); qx = px + ig + id; dr0 -= (double)id;
    ig = rha_cs(d1); dr1 += d1 - (double)ig; id = rha_cs(dr1); qy = py + ig + id; dr1 -= (double)id;
    ig = rha_cs(d2); dr2 += d2 - (double)ig; id = rha_cs(dr2); qz = pz + ig + id; dr2 -= (double)id;
    qx = wrapi(qx, g.nx); qy = wrapi(qy, g.ny); qz = wrapi(qz, g.nz);
}

// One whole neargrid step computed from rho directly (used by the exact slow kernel).
__device__ __forceinline__ bool ng_step(const double *__restrict__ rho, const Grid &g, int px, int py, int pz,
                                        int lp, double c, double &dr0, double &dr1, double &dr2, int &qx,
                                        int &qy, int &qz) {
    double d0, d1, d2;
    if (ng_dir(rho, g, px, py, pz, lp, c, d0, d1, d2)) { qx = px; qy = py; qz = pz; return true; }
    ng_move(g, px, py, pz, d0, d1, d2, dr0, dr1, dr2, qx, qy, qz);
    return false;
}

// Per-voxel record of the gradient-field table (32 B, one aligned gather per trajectory step):
//   r0,r1,r2  remainder part of the normalised direction, grad_dir - int_grad (refinement.py:143),
//             exactly as the reference forms it (one float64 subtraction)
//   key       the density with its 21 lowest mantissa bits replaced by
//               bits  0-5   packed integer step int_grad+1 (2 bits per axis); 63 = the trajectory does
//                           not leave this voxel by a gradient step (max_grad < 1E-14)
//               bits  6-10  the ongrid successor (methods.py:87-117) as (ix+1)*9+(iy+1)*3+(iz+1);
//                           13 = the voxel itself, i.e. a 26-neighbour maximum
//               bits 11-20  trapping-box id + 1 (0 = in no box), see k_box_scan
// `key` orders path voxels for the window test only (any fixed per-voxel function keeps that test
// sound, see PathWindow); exact densities are never taken from it.
struct __attribute__((aligned(32))) GradRec { double r0, r1, r2, key; };
#define XB_STAY_CODE 63
#define XB_OG_SELF 13
#define XB_MAX_BOXES 1023
__device__ __forceinline__ GradRec fetch_rec(const GradRec *__restrict__ G, int l) {
    return *reinterpret_cast<const GradRec *>(reinterpret_cast<const char *>(G) + ((unsigned long long)(unsigned)l << 5));
}

// The table of a slab holds the records of its window planes only (round 3: a rank no longer allocates 32 B for every voxel
// of the grid): voxel l of a window plane lies at slot (l - wbase) mod ntot, the window's planes in order even when the
// window wraps round the periodic boundary.  With the whole grid as window (one GPU) wbase = 0 and the slot is l.
template <typename GT>
__device__ __forceinline__ int rec_slot(const GT &g, int l) {
    const int s = l - g.wbase;
    return s + ((s >> 31) & g.ntot);
}
template <typename GT>
__device__ __forceinline__ GradRec fetch_rec_w(const GT &g, const GradRec *__restrict__ G, int l) { return fetch_rec(G, rec_slot(g, l)); }

__device__ __forceinline__ double pack_key(double rho, int code, int og) {
    return __longlong_as_double((__double_as_longlong(rho) & ~0x1FFFFFLL) | (long long)(code | (og << 6)));
}
// the largest value with all-equal record bits that is <= every key pack_key can make of `rho` (round down on those bits)
__device__ __forceinline__ double key_floor(double rho) {
    const long long b = __double_as_longlong(rho);
    return __longlong_as_double(b >= 0 ? (b & ~0x1FFFFFLL) : (b | 0x1FFFFFLL));
}
__device__ __forceinline__ int key_bits(double key) { return (int)(__double_as_longlong(key) & 0x1FFFFFLL); }
__device__ __forceinline__ int key_code(double key) { return key_bits(key) & 63; }
__device__ __forceinline__ int key_og(double key) { return (key_bits(key) >> 6) & 31; }
__device__ __forceinline__ int key_box(double key) { return key_bits(key) >> 11; }

// wrap q in [-n, 2n) into [0, n) with two unsigned minima
__device__ __forceinline__ int wrap_u(int q, int n) {
    unsigned u = min((unsigned)q, (unsigned)(q + n));
    return (int)min(u, u - (unsigned)n);
}
__device__ __forceinline__ int lin3f(const GridL &g, int x, int y, int z) {
    if (g.use24) return (int)__umul24(__umul24(x, g.ny) + y, g.nz) + z;
    return (x * g.ny + y) * g.nz + z;
}
// Table-driven move (refinement.py:142-154): dr += r; corr = rha(dr); q = p + step + corr; dr -= corr.
__device__ __forceinline__ void ng_move_t(const GridL &g, int px, int py, int pz, const GradRec &rec, int code,
                                          double &dr0, double &dr1, double &dr2, int &qx, int &qy, int &qz) {
    int id;
    dr0 += rec.r0; id = rha_cs(dr0); qx = px + ((code & 3) - 1) + id; dr0 -= (double)id;
    dr1 += rec.r1; id = rha_cs(dr1); qy = py + (((code >> 2) & 3) - 1) + id; dr1 -= (double)id;
    dr2 += rec.r2; id = rha_cs(dr2); qz = pz + ((code >> 4) - 1) + id; dr2 -= (double)id;
    qx = wrap_u(qx, g.nx); qy = wrap_u(qy, g.ny); qz = wrap_u(qz, g.nz);
}

// One ongrid step (methods.py:84-117; refinement.py:204-235): the best of the 27 neighbours by
// (rho(n)-rho(p))*dist_mat + rho(p), strict '>', first wins in (ix,iy,iz) ascending order.
template <typename GT>
__device__ __forceinline__ void og_step(const double *__restrict__ rho, const GT &g, const double *__restrict__ dist,
                                        int px, int py, int pz, double ctr, int &qx, int &qy, int &qz) {
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
                double v = rho[(tx * g.ny + ty) * g.nz + tz];
                v = (v - ctr) * dist[((ix + 3) % 3) * 9 + ((iy + 3) % 3) * 3 + ((iz + 3) % 3)];
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
        for (int k = 0; k < K; k++) { idx[k] = -1; val[k] = -1.7976931348623157e308; }
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
        m_old = (val[K - 1] > m_old) ? val[K - 1] : m_old;
#pragma unroll
        for (int k = K - 1; k > 0; k--) { idx[k] = idx[k - 1]; val[k] = val[k - 1]; }
        idx[0] = l; val[0] = c;
    }
};

// The table record of voxel (x,y,z) derived from rho on the spot -- the arithmetic of k_grad_field, operation
// for operation -- for the few trajectories that leave the table window of a slab (multi-GPU): they carry
// on with ~33 loads per step instead of one gather, and never need the exact slow kernel for that.
// gc: dist_mat (27 doubles) followed by T_grad (9 doubles) in device memory.
struct TGradView { double T[9]; int main_ties; };
__device__ __noinline__ GradRec make_rec_rho(const GridL &g, const double *__restrict__ rho,
                                             const double *__restrict__ gc, int x, int y, int z) {
    const int v = (x * g.ny + y) * g.nz + z;
    const double c = rho[v];
    double max_val = c;
    int og = XB_OG_SELF;
    double nb[27];  // all 27 loads in flight before the scan
#pragma unroll
    for (int j = 0; j < 27; j++)
        nb[j] = rho[(wrapi(x + j / 9 - 1, g.nx) * g.ny + wrapi(y + (j / 3) % 3 - 1, g.ny)) * g.nz + wrapi(z + j % 3 - 1, g.nz)];
#pragma unroll
    for (int j = 0; j < 27; j++) {
        const int ix = j / 9, iy = (j / 3) % 3, iz = j % 3;
        double w = (nb[j] - c) * gc[((ix + 2) % 3) * 9 + ((iy + 2) % 3) * 3 + ((iz + 2) % 3)];
        w += c;
        if (w > max_val) { max_val = w; og = j; }
    }
    TGradView t;
    for (int k = 0; k < 9; k++) t.T[k] = gc[27 + k];
    t.main_ties = g.main_ties;
    const int xp = wrapi(x + 1, g.nx), xm = wrapi(x - 1, g.nx);
    const int yp = wrapi(y + 1, g.ny), ym = wrapi(y - 1, g.ny);
    const int zp = wrapi(z + 1, g.nz), zm = wrapi(z - 1, g.nz);
    GradRec o;
    double d0, d1, d2;
    int code;
    if (ng_dir_vals(t, c, rho[(xp * g.ny + y) * g.nz + z], rho[(xm * g.ny + y) * g.nz + z], rho[(x * g.ny + yp) * g.nz + z],
                    rho[(x * g.ny + ym) * g.nz + z], rho[(x * g.ny + y) * g.nz + zp], rho[(x * g.ny + y) * g.nz + zm], d0, d1, d2)) {
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
    return o;
}

enum { TR_STEP = 0, TR_NEED_OG = 1, TR_DONE = 2 };
