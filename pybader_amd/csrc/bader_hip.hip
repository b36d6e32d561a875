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
#include <thread>
#include <vector>

// every wait of the host for the card made inside the library is counted (per calling thread: xb_host_waits)
static thread_local long long xb_waits = 0;
static inline hipError_t xb_counted_sync(hipStream_t s) { xb_waits++; return (hipStreamSynchronize)(s); }
#define hipStreamSynchronize(s) xb_counted_sync(s)

// =============================================================================================
// kernels
// =============================================================================================
#include "k_common.h"
#include "k_table.h"
#include "k_masks.h"
#include "k_fused.h"
#include "k_trace.h"
#include "k_ongrid.h"
#include "k_edges.h"
#include "k_sums.h"
#include "k_text.h"

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
#define XB_BIG_CHUNK ((size_t)16 << 20)
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

namespace xbcomm { struct State; }
struct xb_ctx {
    int device = 0;
    xbcomm::State *comm = nullptr;   // RCCL transport (comm.h), one process per GPU
    hipStream_t stream = nullptr;
    Grid g{};
    bool has_grid = false;
    long long N = 0;
    int halo = 0;
    double *rho = nullptr;
    GradRec *grad = nullptr;   // gradient-field table, 32 B per voxel of the table window (the whole grid on one GPU)
    long long grad_cap = 0;    // records allocated
    long long list_cap = 0;    // ints allocated for `list` (N on one GPU; the slab's planes + scratch on a slab)
    int *ec_buf = nullptr;     // edge_check's two seed / overflow lists when `stage` is too small for them (slabs)
    long long ec_buf_cap = 0;
    double *dist_dev = nullptr; // dist_mat on the device
    int *boxbuf = nullptr;      // seeds / box tables of the table build (BB_* layout)
    int *box_max_tab = nullptr; // region id - 1 -> voxel of the region's maximum (inside boxbuf: BB_BOXMAX or BB_REGMAX)
    int n_boxes = 0;
    long long box_voxels = 0;
    int opt_boxes = 1;
    int opt_bricks = 1;
    int opt_dbg = 0;
    int opt_ec_groups = 256;    // workgroups of k_ec_chase (at most one per CU)
    int opt_ec_qcap = EC_Q;     // LDS queue entries used per buffer (smaller only in tests)
    std::vector<int64_t> esc_starts, esc_offsets, esc_vox;  // xb_escaped_paths -> xb_escaped_paths_fetch
    unsigned long long *ec_pend = nullptr;  // edge_check's counter word per voxel (8 N bytes, allocated on first use)
    std::vector<int8_t> esc_complete;
    bool has_vacuum = true;    // false only when volumes_init proved there is no -1 label
    bool regions_pending = false;  // labels of certain bricks are written by the relabel pass
    bool buni_valid = false;       // per-brick label uniformity (in `st`) matches the resident labels
    bool buni_halo_safe = false;   // ... and marks every brick outside the owned planes that is not of a trapping region as mixed:
                                   // it stays right when the peers' halo planes arrive (slabs, xb_assign_finish)
    bool regions_labels = false;   // the resident labels are the last neargrid assignment's (+ refinement): every voxel of a
                                   // trapping-region brick (blab > 0) still carries the region's label
    int n_walk = 0;                // bricks on the walk list of the last assignment
    int *walk = nullptr;           // ... and where that list lives (inside `list`)
    int table_margin = -1;         // planes of table each side of the slab (slabs); -1: whole grid
    bool table_prebuilt = false;   // xb_table_finish done: the next xb_assign_trace must not rebuild
    int table_stage = 0;           // windowed build: 1 = records + masks done, 2 = trapping regions done
    std::vector<int> window_seeds; // maxima found in the owned planes (windowed build)
    bool window_ties = true;       // the window holds a voxel whose record depends on the tie rule (windowed build)
    bool slab_sparse = false;      // windowed build by passes A / B (k_masks.h): masks for the own bricks, records for the
                                   // uncertain bricks of the window; false: round 1's full record per window voxel
    int ec_local_n = 0;            // xb_edge_check_local -> xb_edge_check_local_fetch
    int walk_n_out = 0;            // walkers exported by the last xb_refine_trace / xb_walkers_continue ...
    void *walk_out_dev = nullptr;  // ... and where they lie (device)
    std::vector<int64_t> walk_host, res_host;   // ... fetched: walkers (10 int64 each), result pairs (voxel | label << 32)
    int walk_n_res = 0;            // (start voxel, label) pairs of the last xb_walkers_continue
    void *walk_in = nullptr, *walk_out2 = nullptr, *walk_res = nullptr;   // xb_walkers_continue: incoming walkers, re-exported ones, results
    long long walk_cap = 0;
    long long stat_deferred = 0;   // retraces redone by the from-rho kernel (sparse table)
    long long stat_ovf_assign = 0, stat_ovf_refine = 0;   // trajectories handed to the exact slow kernel
    int *blab = nullptr;        // brick labels of the trapping regions (inside `list`), or null
    int nbk[3] = {0, 0, 0};
    int opt_trace_tpb = 64;   // one wave per block: a finished wave frees its slot at once
    bool grad_valid = false;
    int grad_cover = 0;        // 0: the table holds a record for every voxel (of the window); 1: only for the bricks flagged in brick_rec
    unsigned char *brick_rec = nullptr;   // per 8^3 brick: its records exist (k_brick_records), nbr bytes inside blab_buf's allocation
    void *xbuf = nullptr;      // the device-driven slab step's exchange blocks 3-5 (slab_step.h): tie flags, counters, maxima tables
    int slab_rank = 0, slab_nranks = 0, slab_stage = 0;
    void *wbuf[2] = {nullptr, nullptr};   // ... blocks 6 / 7: the walkers of a refinement pass and their results, one part per rank
    void *wk_in = nullptr;                // the walkers this rank carries on in a round (+ their count)
    int wbuf_ranks = 0, walk_last = -1, walk_round = 0, wcap = 0;
    int opt_async_comm = 0;    // collectives return without waiting (they are ordered on the context's stream); the device-driven slab step sets it
    int opt_self_exchange = 0; // tests only: xb_comm_exchange_planes accepts this rank as its own peer (one GPU exercises pack / send / recv / unpack)
    int opt_lean_mem = 1;      // slabs: table, `list` and `stage` sized by the slab instead of the grid (0: everything full size)
    int opt_trace_cache = 1;   // group trace: the own brick's records in LDS (k_ng_trace_g, LEAN 3 / 4)
    int opt_mask_diag = 1;     // pass A: the three-product form of T_grad . grad on orthogonal lattices (tests compare)
    int opt_narrow_halo = 1;   // label halos travel as dtype_calc(-n_maxima) (int8 / int16) instead of int32 (comm.h)
    int opt_chase = 1;         // region growth: provisional labels by k_grow_parent / k_grow_chase (0: propagation launches)
    int grow_kill_launches = 6;   // kill launches scheduled after a chase (raised to the worst case by the first assignment that needs more)
    long long stat_grow_retries = 0;
    int opt_trace_group = 8;   // persistent trace: waves per workgroup (the eighths of a brick on one compute unit, k_ng_trace_g)
    int opt_lean = 1;          // persistent trace: the lean walker (k_trace.h, ng_walk_lean); 0: ng_walk_wave (tests compare)
    int opt_mirror = 1;        // pass A: mirror prefilter of the ongrid face test (k_masks.h, bm_mirror)
    int opt_sparse = 1;        // single-GPU neargrid assignment: brick masks for every voxel + records for the walk-list bricks only
    int grad_rule = 0;         // which tie rule the resident table obeys: 0 refinement.py:111, 1 methods.py:324, 2 both
                               // (the density has no voxel where they differ)
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
    int *fs = nullptr;         // state block of the device-side control flow (k_fused.h), inside `counters`
    int *blab_buf = nullptr;   // brick labels of the trapping regions (fused path), nbr ints
    long long blab_alloc = 0;
    bool labels_zero_pending = false;   // volumes_init without vacuum: labels := 0 is owed (see xb_vacuum_assign)
    int zero_outside[3] = {-1, -1, -1}; // slab (x0, x1, halo) for which every label outside the planes [x0-halo, x1+halo) is known to be 0
    int opt_fused = 1;         // 0: the host-driven round-1 orchestration (kept for slabs and odd grids)
    int opt_trace_grid = 8192; // one-wave workgroups of the persistent trace
    int opt_trace_chunk = 1;   // items (4x4x4 eighths of a brick) per pull: 1 keeps the waves of an XCD on ~128 neighbouring bricks (2 MB of table, L2 resident); 32 per pull ran 1.8x slower
    int opt_trace_xcd = 1;     // 1: ranges by the real XCC id, 0: by blockIdx % 8
    int opt_morton = 1;        // walk list in Morton order of the bricks
    unsigned long long *counters64 = nullptr;
    double *dsum = nullptr;
    int *host_ints = nullptr;  // pinned
    char *big_pin[2] = {nullptr, nullptr};   // two pinned chunks for the large transfers (staged_h2d / staged_d2h)
    hipEvent_t big_ev[2] = {nullptr, nullptr};
    char *pin = nullptr;       // pinned staging for the small host arrays a step uploads (pageable copies pin pages on the fly)
    size_t pin_bytes = 0;
    std::vector<int> maxima_sorted;  // global, label order
    std::vector<int> local_max, local_first;
    bool first_clean = false;
    int list_n = 0;            // entries of `list` that hold the owned known == -2 voxels ...
    bool list_valid = false;   // ... when this is set (by xb_edge_find)
    bool timing = false;
    int opt_trace = 1;   // bit0: 4x4x4 brick per wave, bit1: XCD-aware block order
    TimedKernel tk[8];
    long long n_alloc = 0;
};

static inline unsigned nblocks(long long n) { return (unsigned)((n + TPB - 1) / TPB); }
static GridL light(const Grid &g) {
    GridL l;
    l.nx = g.nx; l.ny = g.ny; l.nz = g.nz; l.nyz = g.nyz;
    l.x0 = g.x0; l.x1 = g.x1; l.vx0 = g.vx0; l.vlen = g.vlen;
    l.wx0 = g.wx0; l.wlen = g.wlen; l.wbase = g.wbase; l.ntot = g.ntot;
    l.use24 = ((long long)g.nx * g.ny < (1 << 24)) && g.nz < (1 << 24);
    l.main_ties = g.main_ties;
    return l;
}

// temporary device memory that is released on every return path
template <typename T>
struct DevBuf {
    T *p = nullptr;
    hipError_t alloc(size_t n) { return hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)); }
    ~DevBuf() { hipFree(p); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

// the 14-distance form of the grid for the tiled field kernels; false when dist_mat is not symmetric
static bool sym_grid(const Grid &g, GridS &s) {
    s.nx = g.nx; s.ny = g.ny; s.nz = g.nz; s.nyz = g.nyz;
    s.x0 = g.x0; s.x1 = g.x1; s.vx0 = g.vx0; s.vlen = g.vlen; s.wx0 = g.wx0; s.wlen = g.wlen; s.wbase = g.wbase; s.ntot = g.ntot;
    s.main_ties = g.main_ties;
    for (int k = 0; k < 9; k++) s.T[k] = g.T[k];
    auto at = [&](int idx) {
        const int ix = idx / 9, iy = (idx / 3) % 3, iz = idx % 3;
        return g.dist[((ix + 2) % 3) * 9 + ((iy + 2) % 3) * 3 + ((iz + 2) % 3)];
    };
    for (int idx = 0; idx <= 13; idx++) {
        const double a = at(idx), b = at(26 - idx);
        if (std::memcmp(&a, &b, sizeof a) != 0) return false;
        s.dsym[idx] = a;
    }
    return true;
}

// Mirror prefilter of pass A (k_masks.h, bm_mirror): which axes of dist_mat are mirror symmetric (bit j: d(+1 on axis j) ==
// d(-1 on axis j) for all nine pairs -- that lattice vector is orthogonal to the other two) and the margin scale
// 2^-48 (1 + 1/d_min).  dist_mat is indexed 0, +1, -1 (index 2 == -1).
static void mirror_prefilter(const Grid &g, int &mirror, double &mu_scale) {
    mirror = 0;
    double dmin = 0.;
    bool first = true;
    for (int i = 0; i < 27; i++)
        if (i != 0) { dmin = first ? g.dist[i] : std::min(dmin, g.dist[i]); first = false; }
    if (!(dmin > 0.) || !std::isfinite(dmin)) { mu_scale = 0.; return; }
    for (int ax = 0; ax < 3; ax++) {
        bool ok = true;
        for (int u = 0; u < 3 && ok; u++)
            for (int v = 0; v < 3 && ok; v++) {
                int ip[3], im[3];
                ip[ax] = 1; im[ax] = 2;
                ip[(ax + 1) % 3] = im[(ax + 1) % 3] = u;
                ip[(ax + 2) % 3] = im[(ax + 2) % 3] = v;
                const double a = g.dist[ip[0] * 9 + ip[1] * 3 + ip[2]], b = g.dist[im[0] * 9 + im[1] * 3 + im[2]];
                ok = std::memcmp(&a, &b, sizeof a) == 0;
            }
        if (ok) mirror |= 1 << ax;
    }
    mu_scale = std::ldexp(1. + 1. / dmin, -48);
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
    HIPCHK(hipMalloc(&c->counters, 1024 * sizeof(int)));
    c->fs = c->counters + 128;
    HIPCHK(hipMalloc(&c->counters64, 16 * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&c->dsum, 16 * sizeof(double)));
    HIPCHK(hipMalloc(&c->dist_dev, 36 * sizeof(double)));  // dist_mat (27) then T_grad (9): make_rec_rho
    HIPCHK(hipMalloc(&c->boxbuf, (size_t)(1 << 20) * sizeof(int)));
    HIPCHK(hipHostMalloc(&c->host_ints, 4096 * sizeof(int)));
    *out = c;
    return XB_OK;
}

static void free_grid(xb_ctx *c) {
    hipFree(c->rho); hipFree(c->grad); hipFree(c->labels); hipFree(c->known); hipFree(c->first); hipFree(c->list);
    hipFree(c->st); hipFree(c->stage); hipFree(c->ec_pend); c->ec_pend = nullptr; hipFree(c->max_list); hipFree(c->max_aux); hipFree(c->ovf_list);
    hipFree(c->blab_buf); c->blab_buf = nullptr; c->blab_alloc = 0; c->labels_zero_pending = false;
    hipFree(c->ec_buf); c->ec_buf = nullptr; c->ec_buf_cap = 0; c->grad_cap = 0; c->list_cap = 0;
    c->brick_rec = nullptr; c->grad_cover = 0;
    c->rho = nullptr; c->grad = nullptr; c->grad_valid = false; c->labels = nullptr; c->known = nullptr; c->first = nullptr; c->list = nullptr;
    c->st = nullptr; c->stage = nullptr; c->max_list = nullptr; c->max_aux = nullptr; c->ovf_list = nullptr;
    c->n_alloc = 0; c->stage_bytes = 0;
}

int xb_comm_destroy(xb_ctx *c);
void xb_destroy(xb_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    xb_comm_destroy(c);
    for (auto &t : c->tk)
        for (auto &p : t.pending) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    free_grid(c);
    hipFree(c->counters); hipFree(c->counters64); hipFree(c->dsum); hipFree(c->dist_dev); hipFree(c->boxbuf);
    hipFree(c->walk_in); hipFree(c->walk_out2); hipFree(c->walk_res); hipFree(c->xbuf); hipFree(c->wbuf[0]); hipFree(c->wbuf[1]); hipFree(c->wk_in);
    hipHostFree(c->host_ints);
    hipHostFree(c->pin);
    for (int k = 0; k < 2; k++) { hipHostFree(c->big_pin[k]); if (c->big_ev[k]) hipEventDestroy(c->big_ev[k]); }
    hipStreamDestroy(c->stream);
    delete c;
}

int xb_sync(xb_ctx *c) {
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}
void *xb_stream(xb_ctx *c) { return (void *)c->stream; }

// `list` and `stage` hold lists over the planes a rank works on (edges, tiles, walkers, dtype staging) and a few
// per-brick arrays: the whole grid's worth on one GPU, the slab + halo (+ tile rounding) on a slab.  Grown on demand,
// never shrunk while the grid stays; contents are scratch between calls EXCEPT the walk list (set up after this).
static int need_scratch(xb_ctx *c) {
    const Grid &g = c->g;
    const long long N = c->N ? c->N : (long long)g.nx * g.nyz;
    const int own = g.x1 - g.x0;
    long long planes = own + 2LL * (std::max(c->halo, 16) + 16);
    if (own == g.nx || planes >= g.nx || !c->opt_lean_mem) planes = g.nx;
    const long long list_want = planes == g.nx ? N : std::max<long long>(planes * g.nyz, 8 * (N / 512) + 4096);
    const size_t stage_want = planes == g.nx ? (size_t)N * 8 : std::max<size_t>((size_t)planes * g.nyz * 8, (size_t)64 << 20);
    if (c->list_cap < list_want) {
        HIPCHK(hipStreamSynchronize(c->stream));
        hipFree(c->list); c->list = nullptr; c->list_cap = 0;
        HIPCHK(hipMalloc(&c->list, (size_t)list_want * sizeof(int)));
        c->list_cap = list_want;
        c->list_valid = false; c->walk = nullptr; c->n_walk = 0;
    }
    if (c->stage_bytes < stage_want) {
        HIPCHK(hipStreamSynchronize(c->stream));
        hipFree(c->stage); c->stage = nullptr; c->stage_bytes = 0;
        HIPCHK(hipMalloc(&c->stage, stage_want));
        c->stage_bytes = stage_want;
    }
    return XB_OK;
}
// the table: one record per voxel of the window planes (xb_set_table_window), allocated when a build first needs it
static int need_grad(xb_ctx *c) {
    Grid &g = c->g;
    const long long want = (long long)g.wlen * g.nyz;
    g.wbase = g.wlen < g.nx ? g.wx0 * g.nyz : 0;
    g.ntot = (int)c->N;
    if (c->grad_cap < want || c->grad_cap > 2 * want) {
        HIPCHK(hipStreamSynchronize(c->stream));
        hipFree(c->grad); c->grad = nullptr; c->grad_cap = 0; c->grad_valid = false;
        HIPCHK(hipMalloc(&c->grad, (size_t)want * sizeof(GradRec)));
        c->grad_cap = want;
    }
    return XB_OK;
}

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
        // (the table -- 32 B per voxel of its window -- and the two scratch arrays are sized by what this rank works on:
        // need_grad / need_scratch, below and on xb_set_halo / xb_set_table_window)
        HIPCHK(hipMalloc(&c->labels, N * sizeof(int)));
        HIPCHK(hipMalloc(&c->known, N + 16));  // slack: edge_check reads the 3 z-neighbours as one 32-bit word
        HIPCHK(hipMalloc(&c->first, N * sizeof(int)));
        HIPCHK(hipMalloc(&c->st, N));
        c->max_cap = (int)std::min<long long>(N, 1 << 22);
        HIPCHK(hipMalloc(&c->max_list, c->max_cap * sizeof(int)));
        HIPCHK(hipMalloc(&c->max_aux, c->max_cap * sizeof(int)));
        c->ovf_cap = (int)std::min<long long>(N, 1 << 22);
        HIPCHK(hipMalloc(&c->ovf_list, c->ovf_cap * sizeof(int)));
        c->n_alloc = N;
        c->first_clean = false;
    }
    c->zero_outside[0] = -1;
    Grid &g = c->g;
    if (g.nx != (int)shape[0] || g.ny != (int)shape[1] || g.nz != (int)shape[2]) c->grad_valid = false;
    if (dist_mat && !T_grad) return fail(XB_E_ARG, "xb_set_grid: dist_mat without T_grad");
    g.nx = (int)shape[0]; g.ny = (int)shape[1]; g.nz = (int)shape[2];
    g.nyz = g.ny * g.nz;
    g.x0 = (int)x0; g.x1 = (int)x1;
    if (dist_mat) {
        if (memcmp(g.dist, dist_mat, sizeof g.dist) != 0) c->grad_valid = false;   // the tabulated ongrid successors depend on it
        memcpy(g.dist, dist_mat, sizeof g.dist);
        HIPCHK(hipMemcpyAsync(c->dist_dev, dist_mat, sizeof g.dist, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->dist_dev + 27, T_grad, sizeof g.T, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (T_grad && memcmp(g.T, T_grad, sizeof g.T) != 0) { memcpy(g.T, T_grad, sizeof g.T); c->grad_valid = false; }
    c->N = N;
    c->halo = (x0 == 0 && x1 == shape[0]) ? g.nx : 0;
    set_valid_range(c);
    g.wx0 = 0; g.wlen = g.nx;      // table window: whole grid unless xb_set_table_window says otherwise
    g.wbase = 0; g.ntot = (int)N;
    c->table_margin = -1;
    if (int rc = need_scratch(c)) return rc;
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

static int settle_labels(xb_ctx *c);
int xb_set_halo(xb_ctx *c, int64_t halo) {
    if (!c || !c->has_grid) return fail(XB_E_STATE, "xb_set_halo: no grid");
    if (halo < 2) return fail(XB_E_ARG, "xb_set_halo: halo must be >= 2 planes");
    // a deferred `labels := 0` was sized with the old halo: pay it first, and forget what is known to be zero outside
    HIPCHK(hipSetDevice(c->device));
    if (int rc = settle_labels(c)) return rc;
    c->zero_outside[0] = -1;
    c->halo = (int)halo;
    set_valid_range(c);
    return need_scratch(c);
}

// volumes_init without vacuum owes `labels := 0` (xb_vacuum_assign defers the 4 B/voxel memset because the
// neargrid / ongrid assignment that normally follows overwrites every label without reading any); every other
// entry point pays the debt first, so the deferral is not observable.
static int zero_slab_labels(xb_ctx *c) {   // the owned + halo planes of a slab (the others are known to be zero)
    const Grid &g = c->g;
    const int len = std::min(g.nx, (g.x1 - g.x0) + 2 * c->halo), first = ((g.x0 - c->halo) % g.nx + g.nx) % g.nx;
    const int run1 = std::min(len, g.nx - first);
    HIPCHK(hipMemsetAsync(c->labels + (size_t)first * g.nyz, 0, (size_t)run1 * g.nyz * sizeof(int), c->stream));
    if (len > run1) HIPCHK(hipMemsetAsync(c->labels, 0, (size_t)(len - run1) * g.nyz * sizeof(int), c->stream));
    return XB_OK;
}
static int settle_labels(xb_ctx *c) {
    if (c->labels_zero_pending) {
        c->labels_zero_pending = false;
        if (c->g.x1 - c->g.x0 == c->g.nx) HIPCHK(hipMemsetAsync(c->labels, 0, c->N * sizeof(int), c->stream));
        else return zero_slab_labels(c);
    }
    return XB_OK;
}
#define NEED_GRID_RAW(name) \
    if (!c || !c->has_grid) return fail(XB_E_STATE, name ": call xb_set_grid first"); \
    HIPCHK(hipSetDevice(c->device))
#define NEED_GRID(name) \
    NEED_GRID_RAW(name); \
    if (int rc_ = settle_labels(c)) return rc_

// Large host <-> device transfers of PAGEABLE host memory (every numpy array at the boundary): the runtime stages them
// through its own pinned buffer with one copying thread (7 GB/s measured for a 128 MB density).  Here: two pinned
// buffers in turn, the host side of each chunk copied by a few threads while the previous chunk is on the bus.
static int big_buffers(xb_ctx *c) {
    if (c->big_pin[0]) return XB_OK;
    for (int k = 0; k < 2; k++) {
        HIPCHK(hipHostMalloc(&c->big_pin[k], XB_BIG_CHUNK));
        HIPCHK(hipEventCreateWithFlags(&c->big_ev[k], hipEventDisableTiming));
    }
    return XB_OK;
}
static void copy_threads(char *dst, const char *src, size_t n) {
    const int T = 4;
    const size_t per = ((n / T) + 4095) & ~(size_t)4095;
    std::thread th[T - 1];
    for (int t = 1; t < T; t++) {
        const size_t a = std::min(n, per * t), b = std::min(n, per * (t + 1));
        th[t - 1] = std::thread([=] { if (b > a) memcpy(dst + a, src + a, b - a); });
    }
    memcpy(dst, src, std::min(n, per));
    for (auto &x : th) x.join();
}
static int staged_h2d(xb_ctx *c, void *dst_dev, const void *src_host, size_t bytes) {
    if (bytes < (4u << 20)) { HIPCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream)); return XB_OK; }
    if (int rc = big_buffers(c)) return rc;
    size_t off = 0;
    for (int k = 0; off < bytes; k ^= 1) {
        const size_t n = std::min<size_t>(XB_BIG_CHUNK, bytes - off);
        HIPCHK(hipEventSynchronize(c->big_ev[k]));   // the transfer that last used this buffer is done
        copy_threads(c->big_pin[k], (const char *)src_host + off, n);
        HIPCHK(hipMemcpyAsync((char *)dst_dev + off, c->big_pin[k], n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipEventRecord(c->big_ev[k], c->stream));
        off += n;
    }
    return XB_OK;
}
static int staged_d2h(xb_ctx *c, void *dst_host, const void *src_dev, size_t bytes) {   // returns with the data on the host
    if (bytes < (4u << 20)) {
        HIPCHK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return XB_OK;
    }
    if (int rc = big_buffers(c)) return rc;
    size_t off = 0, prev_off = 0, prev_n = 0;
    int prev = -1;
    for (int k = 0; off < bytes; k ^= 1) {
        const size_t n = std::min<size_t>(XB_BIG_CHUNK, bytes - off);
        HIPCHK(hipMemcpyAsync(c->big_pin[k], (const char *)src_dev + off, n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipEventRecord(c->big_ev[k], c->stream));
        if (prev >= 0) {   // unpack the chunk before while this one is on the bus
            HIPCHK(hipEventSynchronize(c->big_ev[prev]));
            copy_threads((char *)dst_host + prev_off, c->big_pin[prev], prev_n);
        }
        prev = k; prev_off = off; prev_n = n;
        off += n;
    }
    HIPCHK(hipEventSynchronize(c->big_ev[prev]));
    copy_threads((char *)dst_host + prev_off, c->big_pin[prev], prev_n);
    return XB_OK;
}

int xb_upload_density(xb_ctx *c, const double *rho_host) {
    NEED_GRID("xb_upload_density");
    c->grad_valid = false;
    if (int rc = staged_h2d(c, c->rho, rho_host, c->N * sizeof(double))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}
int xb_download_density(xb_ctx *c, double *rho_host) {
    NEED_GRID("xb_download_density");
    HIPCHK(hipMemcpyAsync(rho_host, c->rho, c->N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}

// ---- density block of a CHGCAR / CHG file: text -> resident rho (k_text.h) ----------------------
static int read_counter(xb_ctx *c, int idx, int *out);
__global__ void k_patch_doubles(const long long *__restrict__ at, const double *__restrict__ val, int n, double *out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[at[t]] = val[t];
}
// in-place exclusive scan of n ints on the device (levels of 2048)
static int device_scan(xb_ctx *c, int *data, int n, int *scratch) {
    const int nb = (n + 2047) / 2048;
    k_scan_2048<<<nb, TPB, 0, c->stream>>>(data, n, data, scratch);
    HIPCHK(hipGetLastError());
    if (nb > 1) {
        if (int rc = device_scan(c, scratch, nb, scratch + nb)) return rc;
        k_scan_add<<<nb, TPB, 0, c->stream>>>(data, n, scratch);
        HIPCHK(hipGetLastError());
    }
    return XB_OK;
}
int xb_parse_density_text(xb_ctx *c, const char *text, int64_t nbytes, double divisor, int64_t *n_tokens,
                          int64_t *n_host) {
    NEED_GRID("xb_parse_density_text");
    const Grid &g = c->g;
    if (!text || nbytes <= 0) return fail(XB_E_ARG, "xb_parse_density_text: empty text");
    if (nbytes / (TPB * TXT_BYTES) >= (1LL << 31) - 2) return fail(XB_E_LIMIT, "xb_parse_density_text: text too large");
    if (!(divisor == divisor) || divisor == 0.) return fail(XB_E_ARG, "xb_parse_density_text: bad divisor");
    c->grad_valid = false;
    static const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11,
                                   1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const int nblk = (int)((nbytes + TPB * TXT_BYTES - 1) / (TPB * TXT_BYTES));
    const int todo_cap = 1 << 20;
    unsigned char *dtext = nullptr;
    int *counts = nullptr;
    double *dp10 = nullptr;
    long long *dtodo = nullptr;
    int rc = XB_OK;
    auto cleanup = [&]() { hipFree(dtext); hipFree(counts); hipFree(dp10); hipFree(dtodo); };
    hipError_t e = hipMalloc(&dtext, (size_t)nbytes + 32);
    if (e == hipSuccess) e = hipMalloc(&counts, ((size_t)nblk + nblk / 1024 + 4096) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&dp10, sizeof P10);
    if (e == hipSuccess) e = hipMalloc(&dtodo, (size_t)todo_cap * 2 * sizeof(long long));
    if (e == hipSuccess) e = hipMemcpyAsync(dtext, text, (size_t)nbytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dp10, P10, sizeof P10, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->counters + 6, 0, sizeof(int), c->stream);
    if (e != hipSuccess) { cleanup(); return fail(XB_E_HIP, "xb_parse_density_text: %s", hipGetErrorString(e)); }
    int last_count = 0, last_off = 0, n_todo = 0;
    k_text_count<<<nblk, TPB, 0, c->stream>>>(dtext, nbytes, counts);
    e = hipMemcpyAsync(&last_count, counts + nblk - 1, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) rc = device_scan(c, counts, nblk, counts + nblk);
    if (e == hipSuccess && rc == XB_OK) e = hipMemcpyAsync(&last_off, counts + nblk - 1, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && rc == XB_OK) e = hipStreamSynchronize(c->stream);
    const long long tokens = (long long)last_off + last_count;
    if (e == hipSuccess && rc == XB_OK && tokens < c->N)
        rc = fail(XB_E_SHORT, "xb_parse_density_text: %lld numbers in the text, the grid has %lld voxels", tokens, (long long)c->N);
    if (e == hipSuccess && rc == XB_OK) {
        k_text_parse<<<nblk, TPB, 0, c->stream>>>(dtext, nbytes, counts, dp10, divisor, g.nx, g.ny, g.nz, c->rho, dtodo,
                                                 c->counters + 6, todo_cap);
        e = hipGetLastError();
        if (e == hipSuccess) rc = read_counter(c, 6, &n_todo);
    }
    if (e == hipSuccess && rc == XB_OK && n_todo > todo_cap)
        rc = fail(XB_E_LIMIT, "xb_parse_density_text: %d tokens need the host parser (cap %d)", n_todo, todo_cap);
    if (e == hipSuccess && rc == XB_OK && n_todo) {  // the rare tokens outside the exact fast path: strtod on the host
        std::vector<long long> todo(2 * (size_t)n_todo), at(n_todo);
        std::vector<double> val(n_todo);
        e = hipMemcpyAsync(todo.data(), dtodo, todo.size() * sizeof(long long), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        for (int k = 0; k < n_todo && e == hipSuccess && rc == XB_OK; k++) {
            const long long off = todo[2 * k], idx = todo[2 * k + 1];
            long long end = off;
            while (end < nbytes && !(text[end] == ' ' || (text[end] >= 9 && text[end] <= 13))) end++;
            const std::string tok(text + off, text + end);
            char *stop = nullptr;
            const double v = std::strtod(tok.c_str(), &stop);
            if (stop == tok.c_str() || *stop != 0) rc = fail(XB_E_ARG, "xb_parse_density_text: could not convert '%s' to a number", tok.c_str());
            const long long x = idx % g.nx, r = idx / g.nx;
            at[k] = (x * g.ny + r % g.ny) * g.nz + r / g.ny;
            val[k] = v / divisor;
        }
        if (e == hipSuccess && rc == XB_OK) {
            long long *dat = dtodo;                                   // reuse: indices then values
            double *dval = reinterpret_cast<double *>(dtodo + n_todo);
            e = hipMemcpyAsync(dat, at.data(), n_todo * sizeof(long long), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(dval, val.data(), n_todo * sizeof(double), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) k_patch_doubles<<<(n_todo + 255) / 256, 256, 0, c->stream>>>(dat, dval, n_todo, c->rho);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) return fail(XB_E_HIP, "xb_parse_density_text: %s", hipGetErrorString(e));
    if (rc != XB_OK) return rc;
    if (n_tokens) *n_tokens = tokens;
    if (n_host) *n_host = n_todo;
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
    NEED_GRID_RAW("xb_upload_labels");
    c->labels_zero_pending = false;   // every label is overwritten
    c->zero_outside[0] = -1;
    c->list_valid = false;
    c->has_vacuum = true;
    c->buni_valid = false; c->regions_labels = false;
    const size_t sz = dtype_size(dtype);
    if (!sz) return fail(XB_E_ARG, "xb_upload_labels: bad dtype code %d", dtype);
    if (dtype == XB_I32) {
        if (int rc = staged_h2d(c, c->labels, labels_host, c->N * 4)) return rc;
    } else {
        // through `stage`, a chunk at a time when it is smaller than the grid (slabs)
        const long long per = std::max<long long>(1, (long long)(c->stage_bytes / sz));
        for (long long o = 0; o < c->N; o += per) {
            const long long n = std::min(per, c->N - o);
            if (int rc = staged_h2d(c, c->stage, (const char *)labels_host + (size_t)o * sz, (size_t)n * sz)) return rc;
            if (dtype == XB_I8) k_widen<int8_t><<<nblocks(n), TPB, 0, c->stream>>>((const int8_t *)c->stage, c->labels + o, n);
            else if (dtype == XB_I16) k_widen<int16_t><<<nblocks(n), TPB, 0, c->stream>>>((const int16_t *)c->stage, c->labels + o, n);
            else k_widen<long long><<<nblocks(n), TPB, 0, c->stream>>>((const long long *)c->stage, c->labels + o, n);
            HIPCHK(hipGetLastError());
        }
    }
    // vacuum voxels present?  (the reference's callers hand bader_calc the volumes_init map: -1 only with a vacuum_tol)
    HIPCHK(hipMemsetAsync(c->counters + 14, 0, sizeof(int), c->stream));
    k_any_equal<<<2048, TPB, 0, c->stream>>>(c->labels, c->N, -1, c->counters + 14);
    HIPCHK(hipGetLastError());
    int any = 0;
    if (int rc = read_counter(c, 14, &any)) return rc;
    c->has_vacuum = any != 0;
    return XB_OK;
}
int xb_download_labels(xb_ctx *c, void *labels_host, int dtype) {
    NEED_GRID("xb_download_labels");
    const size_t sz = dtype_size(dtype);
    if (!sz) return fail(XB_E_ARG, "xb_download_labels: bad dtype code %d", dtype);
    if (dtype == XB_I32) {
        if (int rc = staged_d2h(c, labels_host, c->labels, c->N * 4)) return rc;
    } else {
        const long long per = std::max<long long>(1, (long long)(c->stage_bytes / sz));
        for (long long o = 0; o < c->N; o += per) {
            const long long n = std::min(per, c->N - o);
            if (dtype == XB_I8) k_narrow<int8_t><<<nblocks(n), TPB, 0, c->stream>>>(c->labels + o, (int8_t *)c->stage, n);
            else if (dtype == XB_I16) k_narrow<int16_t><<<nblocks(n), TPB, 0, c->stream>>>(c->labels + o, (int16_t *)c->stage, n);
            else k_narrow<long long><<<nblocks(n), TPB, 0, c->stream>>>(c->labels + o, (long long *)c->stage, n);
            HIPCHK(hipGetLastError());
            if (int rc = staged_d2h(c, (char *)labels_host + (size_t)o * sz, c->stage, (size_t)n * sz)) return rc;   // (waits: `stage` is free again)
        }
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
    NEED_GRID_RAW("xb_vacuum_assign");
    c->buni_valid = false; c->regions_labels = false;
    c->labels_zero_pending = false;
    if (vac_tol != vac_tol) {
        // vacuum_tol=None reaches the reference's sweep as NaN (interface.py:459): `rho <= NaN` is never
        // true, so the result is all-zero labels and zero vacuum charge/volume -- no need to read rho.
        // On one slab the 4 B/voxel memset is deferred (settle_labels): the assignment that follows
        // overwrites every label without reading any.
        // A slab clears its own + halo planes only once the others are known to be zero (they were cleared by an earlier
        // call and nothing has written there since).
        const Grid &g = c->g;
        if (g.x1 - g.x0 == g.nx) c->labels_zero_pending = true;
        else if (c->halo < 2 || (g.x1 - g.x0) + 2 * c->halo >= g.nx || c->zero_outside[0] != g.x0 || c->zero_outside[1] != g.x1 ||
                 c->zero_outside[2] != c->halo) {
            HIPCHK(hipMemsetAsync(c->labels, 0, c->N * sizeof(int), c->stream));
            c->zero_outside[0] = g.x0; c->zero_outside[1] = g.x1; c->zero_outside[2] = c->halo;   // (nothing but plane uploads writes out there)
        } else
            c->labels_zero_pending = true;   // (owed for the owned + halo planes; a neargrid assignment on regions writes every owned label itself)
        c->has_vacuum = false;
        if (vac_charge) *vac_charge = 0.;
        if (vac_volume) *vac_volume = 0.;
        return XB_OK;
    }
    c->zero_outside[0] = -1;
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
// (up to XB_BOX_SEEDS_MAX seed cubes: a cell with hundreds of atoms keeps its trapping regions)
enum { BB_SEEDS = 0, BB_SEED_CAP = 4096, BB_MXYZ = 4096, BB_RCAP = 7168, BB_BOXMAX = 8192, BB_EXT = 9216, BB_BAD = 16384,
       BB_TOTAL = 1 << 20, XB_BOX_SEEDS_MAX = 1023 /* box ids fit the 10 key bits */,
       BB_REGMAX = 1 << 16, BB_REGFIRST = 1 << 17 /* maximum / first brick of up to XB_REGIONS_MAX regions (k_seed_bricks) */ };

// (re)build the gradient-field table from the resident density; with `boxes`, also find and stamp
// the trapping boxes around the 26-neighbour maxima (k_box_scan)
static bool table_windowed(const xb_ctx *c) { return c->g.wlen < c->g.nx; }
static int table_regions(xb_ctx *c, std::vector<int> seeds, bool bricks, bool ranges_from_rho = false);

// per-brick arrays that outlive an assignment: blab_buf (nbr ints: region label per brick) and brick_rec (nbr bytes)
static int ensure_brick_bytes(xb_ctx *c, int nbr) {
    if (c->blab_alloc < nbr) {
        hipFree(c->blab_buf); c->blab_buf = nullptr; c->blab_alloc = 0; c->brick_rec = nullptr;
        if (c->grad_cover) { c->grad_cover = 0; c->grad_valid = false; }
        HIPCHK(hipMalloc(&c->blab_buf, (size_t)nbr * sizeof(int) + (size_t)nbr + 16));
        c->blab_alloc = nbr;
        c->brick_rec = reinterpret_cast<unsigned char *>(c->blab_buf + nbr);
    }
    return XB_OK;
}

// main_rule: records under the assignment's tie test (methods.py:324) instead of the refinement's
// (refinement.py:111); a table built for one rule serves the other when no voxel of the density has such a tie.
static int ensure_grad(xb_ctx *c, bool force, bool boxes, bool main_rule) {
    if (int rc = need_grad(c)) return rc;
    c->g.main_ties = main_rule ? 1 : 0;   // the trace / slow kernels of this phase follow the same rule
    if (c->grad_valid && !force && (c->grad_rule == 2 || c->grad_rule == (main_rule ? 1 : 0))) return XB_OK;
    const Grid &g = c->g;
    if (c->grad_valid && !force && !boxes && c->grad_cover == 1 && c->brick_rec) {
        // records exist for the flagged bricks only, under the other tie rule: redo exactly those
        const int nbr = (g.nx / BRK) * (g.ny / BRK) * (g.nz / BRK);
        const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
        ScopedTimer t(c, 4);
        GridS gs;
        if (sym_grid(g, gs))
            k_brick_records<GridS><<<4096, TPB, 0, c->stream>>>(gs, c->rho, c->grad, nullptr, nullptr, nbr, g.ny / BRK, g.nz / BRK, c->brick_rec, small);
        else
            k_brick_records<Grid><<<4096, TPB, 0, c->stream>>>(g, c->rho, c->grad, nullptr, nullptr, nbr, g.ny / BRK, g.nz / BRK, c->brick_rec, small);
        HIPCHK(hipGetLastError());
        c->grad_rule = main_rule ? 1 : 0;
        return XB_OK;
    }
    if (!force && !boxes && c->opt_sparse && !table_windowed(c) && g.x0 == 0 && g.x1 == g.nx && g.nx % BRK == 0 && g.ny % BRK == 0 &&
        g.nz % BRK == 0 && g.nx >= 16 && g.ny >= 16 && g.nz >= 16) {
        // a refinement without a table from an assignment (ongrid, uploaded labels): retraces only run near label
        // boundaries, so only the bricks whose 27-brick surroundings are not of one label get records (k_masks.h);
        // a retrace that walks on through a brick without records is redone by the from-rho kernel
        const int nb0 = g.nx / BRK, nb1 = g.ny / BRK, nb2 = g.nz / BRK, nbr = nb0 * nb1 * nb2;
        if (int rc = ensure_brick_bytes(c, nbr)) return rc;
        const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
        ScopedTimer t(c, 4);
        int *buni = reinterpret_cast<int *>(c->st);
        if (!c->buni_valid) k_label_uniform<<<(unsigned)nbr, TPB, 0, c->stream>>>(light(g), c->labels, nb1, nb2, buni, 0, nbr);
        c->buni_valid = true;
        k_buni3<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nb0, nb1, nb2, buni, buni + nbr);
        k_flag_mixed_bricks<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, buni + nbr, c->brick_rec);
        GridS gs;
        if (sym_grid(g, gs))
            k_brick_records<GridS><<<4096, TPB, 0, c->stream>>>(gs, c->rho, c->grad, nullptr, nullptr, nbr, nb1, nb2, c->brick_rec, small);
        else
            k_brick_records<Grid><<<4096, TPB, 0, c->stream>>>(g, c->rho, c->grad, nullptr, nullptr, nbr, nb1, nb2, c->brick_rec, small);
        HIPCHK(hipGetLastError());
        c->grad_valid = true;
        c->grad_cover = 1;
        c->grad_rule = main_rule ? 1 : 0;
        c->regions_labels = false;
        c->blab = nullptr;
        c->table_stage = 0;
        return XB_OK;
    }
    c->grad_cover = 0;
    ScopedTimer t(c, 4);
    HIPCHK(hipMemsetAsync(c->counters + 9, 0, 2 * sizeof(int), c->stream));
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
        GridS gs;
        if (sym_grid(g, gs))
            k_grad_field<GridS><<<grid, TPB, 0, c->stream>>>(gs, c->rho, c->grad, c->boxbuf + BB_SEEDS, c->counters + 9,
                                                            BB_SEED_CAP, small, bricks ? c->list + nbr_all : nullptr,
                                                            c->counters + 10);
        else
            k_grad_field<Grid><<<grid, TPB, 0, c->stream>>>(g, c->rho, c->grad, c->boxbuf + BB_SEEDS, c->counters + 9,
                                                           BB_SEED_CAP, small, bricks ? c->list + nbr_all : nullptr,
                                                           c->counters + 10);
    }
    HIPCHK(hipGetLastError());
    c->grad_valid = true;
    c->grad_rule = main_rule ? 1 : 0;   // until the tie counter says the rules agree on this density
    c->n_boxes = 0;
    c->box_voxels = 0;
    c->blab = nullptr;
    c->table_stage = 1;
    if (!boxes || !c->opt_boxes) return XB_OK;
    int ns = 0;
    {
        HIPCHK(hipMemcpyAsync(c->host_ints, c->counters + 9, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        ns = c->host_ints[0];
        c->window_ties = c->host_ints[1] != 0;
        // (a windowed table only knows its own planes: xb_table_finish decides with every rank's answer)
        if (c->host_ints[1] == 0 && !table_windowed(c)) c->grad_rule = 2;
    }
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
static int table_regions(xb_ctx *c, std::vector<int> seeds, bool bricks, bool ranges_from_rho) {
    const Grid &g = c->g;
    c->box_max_tab = c->boxbuf + BB_BOXMAX;
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
        if (table_windowed(c) || ranges_from_rho)  // no table (ongrid) / a cube may lie outside the window: ranges from rho
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
        // k_brick_grow: labels travel up to BG bricks per launch; a launch that changes nothing is the fixpoint
        const int max_launches = 2 * (nb0 + nb1 + nb2) + 16;
        const dim3 ggrid((nb2 + BG - 1) / BG, (nb1 + BG - 1) / BG, (nb0 + BG - 1) / BG);
        bool kill_converged = false;
        for (int phase = 0; phase < 2; phase++) {  // 0: propagate provisional labels, 1: kill violators
            for (int launch = 1; launch <= max_launches; launch++) {
                if (launch & 1) HIPCHK(hipMemsetAsync(c->counters + 11, 0, sizeof(int), c->stream));
                k_brick_grow<<<ggrid, BG * BG * BG, 0, c->stream>>>(nb0, nb1, nb2, bmask, seed, buf[cur], buf[1 - cur],
                                                                 c->counters + 11, phase, BG);
                cur = 1 - cur;
                if (!(launch & 1)) {  // poll the change flag of the last two launches
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
        k_count_positive<<<64, TPB, 0, c->stream>>>(blab, nbr, c->counters + 11);
        int ncertain = 0;
        if (int rc = read_counter(c, 11, &ncertain)) return rc;
        c->box_voxels = (long long)ncertain * BRK * BRK * BRK;
        // the labels move out of `list` (the refinement's edge list overwrites it, and the slab retraces still read them)
        if (int rc = ensure_brick_bytes(c, nbr)) return rc;
        HIPCHK(hipMemcpyAsync(c->blab_buf, blab, (size_t)nbr * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
        c->blab = c->blab_buf;
        c->nbk[0] = nb0; c->nbk[1] = nb1; c->nbk[2] = nb2;
    }
    HIPCHK(hipStreamSynchronize(c->stream));  // host vectors must outlive the copies
    c->n_boxes = (int)box_max.size();
    return XB_OK;
}

// host -> device through the pinned staging buffer; `slot` bytes into it (several uploads of one call use disjoint slots).
// The caller synchronises the stream before the buffer is reused.
static int upload_pinned(xb_ctx *c, void *dst, const void *src, size_t bytes, size_t slot = 0) {
    if (!bytes) return XB_OK;
    if (slot + bytes > c->pin_bytes) {
        HIPCHK(hipStreamSynchronize(c->stream));
        const size_t want = std::max<size_t>(2 * (slot + bytes), 1 << 20);
        char *p = nullptr;
        HIPCHK(hipHostMalloc(&p, want));
        if (c->pin && slot) memcpy(p, c->pin, slot);
        hipHostFree(c->pin);
        c->pin = p; c->pin_bytes = want;
    }
    memcpy(c->pin + slot, src, bytes);
    HIPCHK(hipMemcpyAsync(dst, c->pin + slot, bytes, hipMemcpyHostToDevice, c->stream));
    return XB_OK;
}
// device -> host the same way (waits for the stream)
static int download_pinned(xb_ctx *c, void *dst, const void *src_dev, size_t bytes) {
    if (!bytes) return XB_OK;
    if (bytes > c->pin_bytes) {
        HIPCHK(hipStreamSynchronize(c->stream));
        hipHostFree(c->pin); c->pin = nullptr; c->pin_bytes = 0;
        const size_t want = std::max<size_t>(2 * bytes, 1 << 20);
        HIPCHK(hipHostMalloc(&c->pin, want));
        c->pin_bytes = want;
    }
    HIPCHK(hipMemcpyAsync(c->pin, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(dst, c->pin, bytes);
    return XB_OK;
}
static int read_counter(xb_ctx *c, int idx, int *out) {
    HIPCHK(hipMemcpyAsync(c->host_ints, c->counters + idx, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *out = c->host_ints[0];
    return XB_OK;
}

// run the exact slow kernel over ovf_list[0..n) in chunks
static int run_slow(xb_ctx *c, int n, int refine, int *max_count = nullptr, int *changed = nullptr, int *escaped = nullptr) {
    if (!max_count) max_count = c->counters + 0;
    if (!changed) changed = c->counters + 2;
    if (!escaped) escaped = c->counters + 3;
    const int lmax = 1 << 15, chunk = 2048;
    DevBuf<int> path;
    HIPCHK(path.alloc((size_t)chunk * lmax));
    HIPCHK(hipMemsetAsync(c->counters + 8, 0, sizeof(int), c->stream));  // err
    for (int o = 0; o < n; o += chunk) {
        const int m = std::min(chunk, n - o);
        k_trace_slow<<<(m + 63) / 64, 64, 0, c->stream>>>(c->g, c->rho, c->labels, c->known, c->known,
                                                         c->ovf_list + o, m, path.p, lmax, refine, c->first,
                                                         c->max_list, max_count, c->max_cap,
                                                         changed, escaped, c->counters + 8, nullptr);
    }
    hipError_t e = hipGetLastError();
    int err = 0;
    int rc = read_counter(c, 8, &err);   // synchronises the stream: the scratch may go afterwards
    if (e != hipSuccess) return fail(XB_E_HIP, "k_trace_slow: %s", hipGetErrorString(e));
    if (rc) return rc;
    if (err) return fail(XB_E_LIMIT, "trajectory longer than %d voxels", lmax);
    return XB_OK;
}

int xb_assign_trace(xb_ctx *c, int method, int64_t *n_local) {
    NEED_GRID_RAW("xb_assign_trace");
    if (int rc = need_grad(c)) return rc;
    const Grid &g = c->g;
    const long long own = (long long)(g.x1 - g.x0) * g.nyz;
    // a deferred labels := 0 is dropped when this call writes every owned label without reading any: a neargrid assignment
    // over trapping regions without vacuum (the halo planes are the peers' to fill before anything reads them)
    if (c->labels_zero_pending && method == XB_METHOD_NEARGRID && !c->has_vacuum && c->table_prebuilt && c->blab &&
        g.x0 % 8 == 0 && g.x1 % 8 == 0 && g.x1 - g.x0 < g.nx)
        c->labels_zero_pending = false;
    else if (int rc_ = settle_labels(c)) return rc_;
    const int *box_max = nullptr;   // region id - 1 -> its maximum (set once the regions of this call exist)
    int *max_count_dev = c->counters + 0;   // where the kernels of this call count the maxima they note
    bool fast_slab = false;                 // windowed slab on passes A/B: the persistent trace, counts on the device
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
            if (int rc = ensure_grad(c, true, true, true)) return rc;
        }
        c->g.main_ties = 1;   // methods.neargrid's stepping rule for everything the assignment traces
        box_max = c->box_max_tab ? c->box_max_tab : c->boxbuf + BB_BOXMAX;
        {
            ScopedTimer t(c, 0);
            const int opt = c->opt_trace;
            const int tpb = c->opt_trace_tpb;
            const bool slab_bricks = (g.x0 % 8 == 0) && (g.x1 % 8 == 0);
            if (c->blab && slab_bricks) {
                // trapping regions known per brick: fill them in one sweep, trace only the rest
                const int nbr = c->nbk[0] * c->nbk[1] * c->nbk[2];
                int *walk = c->list + 4 * nbr;  // a free slice of `list` (seed, masks and the two growth buffers come first)
                c->walk = walk;
                // (the persistent kernel pays off from ~10^5 list items on: 0.24 vs 0.29 ms with the plain launch for an eighth of
                // 512^3, 6.9 vs 7.7 ms for half of 1024^3)
                fast_slab = table_windowed(c) && c->slab_sparse && c->opt_fused &&
                            (long long)(g.x1 - g.x0) * g.ny * g.nz >= 65536LL * 512;
                int *walk_count = c->counters + 13;
                if (fast_slab) {   // the state block of the device-side control flow: list length, cursors, maxima and redo counts
                    HIPCHK(hipMemsetAsync(c->fs, 0, FS_TOTAL * sizeof(int), c->stream));
                    walk_count = c->fs + FS_N_WALK;
                    max_count_dev = c->fs + FS_N_MAX;
                } else
                    HIPCHK(hipMemsetAsync(c->counters + 13, 0, sizeof(int), c->stream));
                k_brick_walk_list<<<(nbr + 16 * TPB - 1) / (16 * TPB), TPB, 0, c->stream>>>(nbr, (g.x0 / 8) * c->nbk[1] * c->nbk[2],
                                                                         (g.x1 / 8) * c->nbk[1] * c->nbk[2], c->blab, walk, walk_count);
                if (c->has_vacuum) {
                    k_fill_certain<<<nblocks(own), TPB, 0, c->stream>>>(light(g), c->blab, c->nbk[1], c->nbk[2],
                                                                        box_max, c->labels, c->first, c->max_list,
                                                                        max_count_dev, c->max_cap);
                } else {
                    k_note_certain_bricks<<<(nbr + 255) / 256, 256, 0, c->stream>>>(
                        light(g), c->nbk[0], c->nbk[1], c->nbk[2], (g.x0 / 8) * c->nbk[1] * c->nbk[2],
                        (g.x1 / 8) * c->nbk[1] * c->nbk[2], c->blab, box_max, c->first, c->max_list,
                        max_count_dev, c->max_cap);
                    c->regions_pending = true;
                }
                int nwalk = 0;
                if (fast_slab) {
                    // the persistent trace of the one-GPU path (per-XCD cursors over the list, its length on the device): no
                    // host wait before it.  A trajectory that leaves the table window lands on a list (in `stage`) and is
                    // redone by the kernel that derives missing records from rho.
                    ScopedTimer tw(c, 6);
                    int *redo = (int *)c->stage;
                    const int redo_cap = (int)std::min<size_t>(c->stage_bytes / sizeof(int), 0x7fffffffu);
                    k_ng_trace_p<2, 0><<<c->opt_trace_grid, XB_WAVE, 0, c->stream>>>(light(g), c->grad, box_max, c->blab, c->nbk[1], c->nbk[2], walk,
                                                                                  c->fs, c->labels, c->first, c->max_list, c->max_cap, redo,
                                                                                  redo_cap, maxsteps, c->has_vacuum ? 1 : 0,
                                                                                  c->opt_trace_chunk, c->opt_trace_xcd);
                    k_ng_trace_list<2><<<512, TPB, 0, c->stream>>>(
                        light(g), c->grad, box_max, c->blab, c->nbk[1], c->nbk[2], redo, c->fs + FS_N_OVF, c->labels,
                        c->first, c->max_list, max_count_dev, c->max_cap, c->ovf_list, c->counters + 1, c->ovf_cap,
                        maxsteps, c->rho, c->dist_dev, c->has_vacuum ? 1 : 0);
                } else {
                if (int rc = read_counter(c, 13, &nwalk)) return rc;
                c->n_walk = nwalk;
                }
                if (nwalk) {
                    const long long waves = 8LL * nwalk;
                    ScopedTimer tw(c, 6);
                    const unsigned nblk = (unsigned)((waves + tpb / XB_WAVE - 1) / (tpb / XB_WAVE));
                    if (table_windowed(c)) {
                        // the lean kernel first: a trajectory that leaves the table window lands on a list (in `stage`, its
                        // length stays on the device) and is redone by the kernel that derives missing records from rho
                        int *redo = (int *)c->stage;
                        const int redo_cap = (int)std::min<size_t>(c->stage_bytes / sizeof(int), 0x7fffffffu);
                        HIPCHK(hipMemsetAsync(c->counters + 15, 0, sizeof(int), c->stream));
                        k_ng_trace<2, false><<<nblk, tpb, 0, c->stream>>>(
                            light(g), c->grad, box_max, c->blab, c->nbk[1], c->nbk[2], walk, nwalk, c->labels,
                            c->first, c->max_list, c->counters + 0, c->max_cap, redo, c->counters + 15, redo_cap,
                            maxsteps, opt, c->rho, c->dist_dev, c->has_vacuum ? 1 : 0);
                        k_ng_trace_list<2><<<512, TPB, 0, c->stream>>>(
                            light(g), c->grad, box_max, c->blab, c->nbk[1], c->nbk[2], redo, c->counters + 15, c->labels,
                            c->first, c->max_list, c->counters + 0, c->max_cap, c->ovf_list, c->counters + 1, c->ovf_cap,
                            maxsteps, c->rho, c->dist_dev, c->has_vacuum ? 1 : 0);
                    } else
                        k_ng_trace<2, false><<<nblk, tpb, 0, c->stream>>>(
                            light(g), c->grad, box_max, c->blab, c->nbk[1], c->nbk[2], walk, nwalk, c->labels,
                            c->first, c->max_list, c->counters + 0, c->max_cap, c->ovf_list, c->counters + 1, c->ovf_cap,
                            maxsteps, opt, c->rho, c->dist_dev, c->has_vacuum ? 1 : 0);
                }
            } else {
                const long long waves = (opt & 1)
                    ? (long long)((g.x1 - g.x0 + 3) / 4) * ((g.ny + 3) / 4) * ((g.nz + 3) / 4)
                    : (long long)(g.x1 - g.x0) * g.ny * ((g.nz + 63) / 64);
                (table_windowed(c) ? k_ng_trace<2, true> : k_ng_trace<2, false>)<<<(unsigned)((waves + tpb / XB_WAVE - 1) / (tpb / XB_WAVE)), tpb, 0, c->stream>>>(
                    light(g), c->grad, box_max, c->blab, c->nbk[1], c->nbk[2], nullptr, 0, c->labels,
                    c->first, c->max_list, c->counters + 0, c->max_cap, c->ovf_list, c->counters + 1, c->ovf_cap,
                    maxsteps, opt, c->rho, c->dist_dev, c->has_vacuum ? 1 : 0);
            }
        }
        HIPCHK(hipGetLastError());
        int novf = 0;
        if (fast_slab) {   // one wait: overflows, the list length (xb_assign_finish scans those bricks)
            HIPCHK(hipMemcpyAsync(c->host_ints, c->counters + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(c->host_ints + 1, c->fs + FS_N_WALK, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            novf = c->host_ints[0];
            c->n_walk = c->host_ints[1];
        } else if (int rc = read_counter(c, 1, &novf)) return rc;
        if (novf > c->ovf_cap) return fail(XB_E_LIMIT, "%d trajectories need the slow path (cap %d)", novf, c->ovf_cap);
        c->stat_ovf_assign += novf;
        if (novf > 0) {
            if (fast_slab) {   // the exact slow kernel counts its maxima in counters[0]: carry the count over
                HIPCHK(hipMemcpyAsync(c->counters + 0, max_count_dev, sizeof(int), hipMemcpyDeviceToDevice, c->stream));
                max_count_dev = c->counters + 0;
            }
            if (int rc = run_slow(c, novf, 0)) return rc;
        }
        c->g.main_ties = 0;
    } else if (method == XB_METHOD_ONGRID) {
        c->zero_outside[0] = -1;   // (the pointer pass writes the labels of every plane, on a slab too)
        const int nbr_all = (int)(c->N / (BRK * BRK * BRK));
        // trapping regions of the pointer field (whole 8^3 bricks, one slab, no vacuum), else plain pointer jumping
        const bool regions = c->opt_boxes && c->opt_bricks && !c->has_vacuum && g.x1 - g.x0 == g.nx && g.nx % BRK == 0 &&
                             g.ny % BRK == 0 && g.nz % BRK == 0 && g.nx >= 16 && g.ny >= 16 && g.nz >= 16 &&
                             40LL * nbr_all <= c->N;
        c->blab = nullptr;
        c->n_boxes = 0;
        c->box_voxels = 0;
        {
            ScopedTimer t(c, 1);
            const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
            dim3 grid((g.nz + GT_Z - 1) / GT_Z, (g.ny + GT_Y - 1) / GT_Y, (g.nx + GT_X - 1) / GT_X);
            HIPCHK(hipMemsetAsync(c->counters + 9, 0, sizeof(int), c->stream));
            int *bm = regions ? c->list + nbr_all : nullptr;
            GridS gs;
            if (sym_grid(g, gs))
                k_og_pointer_tiled<GridS><<<grid, TPB, 0, c->stream>>>(gs, c->rho, c->labels, small, c->has_vacuum ? 1 : 0,
                                                                       c->boxbuf + BB_SEEDS, c->counters + 9, BB_SEED_CAP, bm);
            else
                k_og_pointer_tiled<Grid><<<grid, TPB, 0, c->stream>>>(g, c->rho, c->labels, small, c->has_vacuum ? 1 : 0,
                                                                      c->boxbuf + BB_SEEDS, c->counters + 9, BB_SEED_CAP, bm);
        }
        HIPCHK(hipGetLastError());
        if (regions) {
            int ns = 0;
            if (int rc = read_counter(c, 9, &ns)) return rc;
            if (ns >= 1 && ns <= XB_BOX_SEEDS_MAX) {
                std::vector<int> seeds(ns);
                HIPCHK(hipMemcpyAsync(seeds.data(), c->boxbuf + BB_SEEDS, ns * sizeof(int), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                if (int rc = table_regions(c, seeds, true, true)) return rc;
            }
        }
        box_max = c->box_max_tab ? c->box_max_tab : c->boxbuf + BB_BOXMAX;
        if (c->blab) {
            const int nbr = c->nbk[0] * c->nbk[1] * c->nbk[2];
            int *walk = c->list + 4 * nbr;
            c->walk = walk;
            HIPCHK(hipMemsetAsync(c->counters + 13, 0, sizeof(int), c->stream));
            k_brick_walk_list<<<(nbr + 16 * TPB - 1) / (16 * TPB), TPB, 0, c->stream>>>(nbr, 0, nbr, c->blab, walk, c->counters + 13);
            k_note_certain_bricks<<<(nbr + 255) / 256, 256, 0, c->stream>>>(light(g), c->nbk[0], c->nbk[1], c->nbk[2], 0, nbr, c->blab,
                                                                          box_max, c->first, c->max_list,
                                                                          c->counters + 0, c->max_cap);
            int nwalk = 0;
            if (int rc = read_counter(c, 13, &nwalk)) return rc;
            c->n_walk = nwalk;
            if (nwalk) {
                HIPCHK(hipMemsetAsync(c->counters + 8, 0, sizeof(int), c->stream));
                k_og_walk<<<8 * nwalk, XB_WAVE, 0, c->stream>>>(light(g), box_max, c->blab, c->nbk[1], c->nbk[2], walk,
                                                               nwalk, c->labels, c->first, c->max_list, c->counters + 0, c->max_cap,
                                                               1 << 22, c->counters + 8);
                HIPCHK(hipGetLastError());
                int err = 0;
                if (int rc = read_counter(c, 8, &err)) return rc;
                if (err) return fail(XB_E_STATE, "ongrid pointer chase did not terminate");
            }
            c->regions_pending = true;
        } else {
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
        }
    } else
        return fail(XB_E_ARG, "xb_assign: unknown method %d", method);
    int nmax = 0;
    HIPCHK(hipMemcpyAsync(c->host_ints, max_count_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    nmax = c->host_ints[0];
    if (nmax > c->max_cap) return fail(XB_E_LIMIT, "%d maxima exceed the table capacity %d", nmax, c->max_cap);
    c->local_max.resize(nmax);
    c->local_first.resize(nmax);
    if (nmax) {
        k_gather_first<<<(nmax + 255) / 256, 256, 0, c->stream>>>(c->first, c->max_list, nmax, c->max_aux);
        HIPCHK(hipGetLastError());
        if (int rc = download_pinned(c, c->local_max.data(), c->max_list, nmax * sizeof(int))) return rc;
        if (int rc = download_pinned(c, c->local_first.data(), c->max_aux, nmax * sizeof(int))) return rc;
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
        if (int rc = upload_pinned(c, c->max_aux, c->maxima_sorted.data(), n_global * sizeof(int))) return rc;
        k_set_rank<<<(unsigned)((n_global + 255) / 256), 256, 0, c->stream>>>(c->first, c->max_aux, (int)n_global);
        HIPCHK(hipGetLastError());
    }
    c->buni_valid = false; c->regions_labels = false;
    if (c->regions_pending && c->blab) {
        if (g.nz % 4 == 0 && g.ny % 8 == 0 && g.x0 % 8 == 0 && g.x1 % 8 == 0 && c->nbk[1] == g.ny / 8)   // whole bricks: one brick-label lookup per 8 rows
            k_relabel_regions_brick<<<dim3((g.nz / 4 + 63) / 64, c->nbk[1], (g.x1 - g.x0 + 3) / 4), TPB, 0, c->stream>>>(
                light(g), c->labels, c->first, c->blab, c->nbk[1], c->nbk[2], (c->box_max_tab ? c->box_max_tab : c->boxbuf + BB_BOXMAX),
                nullptr, nullptr, c->n_boxes);
        else if (g.nz % 4 == 0)
            k_relabel_regions4<<<nblocks(own / 4), TPB, 0, c->stream>>>(light(g), c->labels, c->first, c->blab, c->nbk[1],
                                                                    c->nbk[2], (c->box_max_tab ? c->box_max_tab : c->boxbuf + BB_BOXMAX), nullptr);
        else
            k_relabel_regions<<<nblocks(own), TPB, 0, c->stream>>>(light(g), c->labels, c->first, c->blab, c->nbk[1], c->nbk[2],
                                                                   (c->box_max_tab ? c->box_max_tab : c->boxbuf + BB_BOXMAX), nullptr);
        if (g.x1 - g.x0 == g.nx) {  // one slab: the per-brick label uniformity edge_find wants comes for free
            const int nbr = c->nbk[0] * c->nbk[1] * c->nbk[2];
            int *buni = reinterpret_cast<int *>(c->st);
            k_buni_from_regions<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, c->blab, (c->box_max_tab ? c->box_max_tab : c->boxbuf + BB_BOXMAX), c->first, buni, nullptr);
            if (c->n_walk)
                k_label_uniform_list<<<(c->n_walk + 3) / 4, TPB, 0, c->stream>>>(light(g), c->labels, c->nbk[1], c->nbk[2],
                                                                                c->walk, c->n_walk, nullptr, nullptr, buni);
            c->buni_valid = true;
            c->buni_halo_safe = false;
        } else if (!c->has_vacuum && g.nx % 8 == 0 && g.ny % 8 == 0 && g.nz % 8 == 0 && g.x0 % 8 == 0 && g.x1 % 8 == 0) {
            // a slab: the regions' bricks are uniform on every rank, the owned walk-list bricks are scanned, every other
            // brick counts as mixed -- right whatever the peers' halo planes bring, and no pass over the labels
            const int nbr = c->nbk[0] * c->nbk[1] * c->nbk[2];
            int *buni = reinterpret_cast<int *>(c->st);
            k_fill<int><<<64, TPB, 0, c->stream>>>(buni, XB_MIXED, nbr);
            k_buni_from_regions<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, c->blab, (c->box_max_tab ? c->box_max_tab : c->boxbuf + BB_BOXMAX), c->first, buni, nullptr);
            if (c->n_walk)
                k_label_uniform_list<<<(c->n_walk + 3) / 4, TPB, 0, c->stream>>>(light(g), c->labels, c->nbk[1], c->nbk[2],
                                                                                c->walk, c->n_walk, nullptr, nullptr, buni);
            c->buni_valid = true;
            c->buni_halo_safe = true;
        }
    } else
        k_relabel<<<nblocks(own), TPB, 0, c->stream>>>(g, c->labels, c->first, nullptr);
    c->regions_labels = c->regions_pending && c->blab && !c->has_vacuum;   // certain bricks carry their region's label now
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


// ---------------------------------------------------------------------------------------------------------------
// The single-GPU neargrid assignment with the control flow on the device (k_fused.h): one host wait at the end.
// Preconditions (checked by the caller): one slab, grid of whole 8^3 bricks, trapping regions enabled.
// ---------------------------------------------------------------------------------------------------------------
static bool fused_ok(const xb_ctx *c) {
    const Grid &g = c->g;
    return c->opt_fused && c->opt_boxes && c->opt_bricks && g.x0 == 0 && g.x1 == g.nx && !table_windowed(c) &&
           g.nx % BRK == 0 && g.ny % BRK == 0 && g.nz % BRK == 0 && 6LL * (c->N / (BRK * BRK * BRK)) <= c->N &&
           g.nx >= 16 && g.ny >= 16 && g.nz >= 16;
}
static int finish_numbering_on_host(xb_ctx *c, int nmax, int64_t *n_maxima);

static int assign_neargrid_fused(xb_ctx *c, int64_t *n_maxima) {
    if (int rc = need_grad(c)) return rc;
    Grid &g = c->g;
    const GridL gl0 = light(g);
    const int nb0 = g.nx / BRK, nb1 = g.ny / BRK, nb2 = g.nz / BRK, nbr = nb0 * nb1 * nb2;
    if (int rc = ensure_brick_bytes(c, nbr)) return rc;
    int *fs = c->fs;
    // scratch carved from `list` (free during an assignment): seed labels, brick masks, two label buffers, walk list
    int *seed = c->list, *bmask = c->list + nbr, *buf0 = c->list + 2 * nbr, *buf1 = c->list + 3 * nbr, *walk = c->list + 4 * nbr;
    const bool sparse = c->opt_sparse != 0;
    int *seeds = c->boxbuf + BB_SEEDS, *mxyz = c->boxbuf + BB_MXYZ, *rcap = c->boxbuf + BB_RCAP,
        *box_max = c->boxbuf + (sparse ? BB_REGMAX : BB_BOXMAX), *bx = c->boxbuf + BB_EXT, *br = c->boxbuf + BB_EXT + 3 * XB_BOXES_MAX,
        *box_first = c->boxbuf + (sparse ? BB_REGFIRST : BB_EXT + 4 * XB_BOXES_MAX), *bad = c->boxbuf + BB_BAD;
    int *bmaxv = walk;   // (free until the walk list is made)
    int *bpot = c->list + 5 * nbr;   // brick potentials of the region growth (k_grow_parent); buf1 doubles as the parent array
    const bool chase = sparse && c->opt_chase;
    c->box_max_tab = box_max;
    const int stride = XB_BOX_K + 4;
    HIPCHK(hipMemsetAsync(fs, 0, FS_TOTAL * sizeof(int), c->stream));
    HIPCHK(hipMemsetAsync(bad, 0, (size_t)XB_BOXES_MAX * stride * sizeof(int), c->stream));
    if (!c->first_clean) {  // a previous assignment did not finish: `first` may hold stale minima
        k_fill<int><<<4096, TPB, 0, c->stream>>>(c->first, XB_INT_MAX, c->N);
        HIPCHK(hipGetLastError());
    }
    c->first_clean = false;
    c->regions_pending = false;
    c->buni_valid = false; c->regions_labels = false;
    c->list_valid = false;
    g.main_ties = 1;   // methods.neargrid's tie test (methods.py:324)
    const GridL gl = light(g);
    (void)gl0;
    {   // brick masks + seeds (+ the full table on the round-1 route, opt_sparse = 0)
        ScopedTimer t4(c, 4);
        {
            ScopedTimer t5(c, 5);
            const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
            dim3 grid((g.nz + GT_Z - 1) / GT_Z, (g.ny + GT_Y - 1) / GT_Y, (g.nx + GT_X - 1) / GT_X);
            GridS gs;
            const bool sym = sym_grid(g, gs);
            if (sparse) {
                // the assignment's tie rule (methods.py:324) is the template argument
                int mirror = 0;
                double mu_scale = 0.;
                if (sym && c->opt_mirror) mirror_prefilter(g, mirror, mu_scale);
                // (an orthogonal lattice has a diagonal T_grad: exact zeros off the diagonal)
                const bool diag = c->opt_mask_diag && g.T[1] == 0. && g.T[2] == 0. && g.T[3] == 0. && g.T[5] == 0. && g.T[6] == 0. && g.T[7] == 0.;
                if (sym && diag) k_brick_masks<GridS, 1, true><<<grid, TPB, 0, c->stream>>>(gs, c->rho, small, bmask, bmaxv, fs + FS_TIES, 0, mu_scale, mirror, bpot);
                else if (sym) k_brick_masks<GridS, 1, false><<<grid, TPB, 0, c->stream>>>(gs, c->rho, small, bmask, bmaxv, fs + FS_TIES, 0, mu_scale, mirror, bpot);
                else k_brick_masks<Grid, 1, false><<<grid, TPB, 0, c->stream>>>(g, c->rho, small, bmask, bmaxv, fs + FS_TIES, 0, 0., 0, bpot);
            } else if (sym)
                k_grad_field<GridS><<<grid, TPB, 0, c->stream>>>(gs, c->rho, c->grad, seeds, fs + FS_N_SEEDS, BB_SEED_CAP, small,
                                                                bmask, fs + FS_TIES);
            else
                k_grad_field<Grid><<<grid, TPB, 0, c->stream>>>(g, c->rho, c->grad, seeds, fs + FS_N_SEEDS, BB_SEED_CAP, small,
                                                               bmask, fs + FS_TIES);
        }
        c->grad_valid = true;
        c->grad_rule = 1;
        c->grad_cover = sparse ? 1 : 0;
        if (sparse) {
            // seeds: the bricks that hold exactly one maximum (no cubes, no cap on the number of maxima); they are not
            // fixed: the kill iteration certifies them like every other brick
            k_seed_bricks<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, bmask, bmaxv, fs, seed, buf0, box_max);
            if (chase) {   // provisional labels by one chase along the brick potentials instead of ~6 propagation launches
                k_grow_parent<<<(nbr + TPB - 1) / TPB, TPB, 0, c->stream>>>(nb0, nb1, nb2, bmask, bpot, seed, buf1);
                k_grow_chase<<<(nbr + TPB - 1) / TPB, TPB, 0, c->stream>>>(nbr, buf1, seed, buf0, 4 * (nb0 + nb1 + nb2) + 64);
                k_seed_finish_kill<<<1, 1, 0, c->stream>>>(fs);
            } else
                k_seed_finish<<<1, 1, 0, c->stream>>>(fs);
        } else {
            // closed seed cubes around the maxima, then brick growth -- all decided on the device
            k_box_setup<<<1, XB_BOXES_MAX, 0, c->stream>>>(gl, fs, seeds, BB_SEED_CAP, XB_BOX_SEEDS_MAX, mxyz, rcap);
            const long long wmax = 2LL * XB_BOX_K + 1;
            k_box_shells_dev<false><<<dim3(nblocks(wmax * wmax * wmax), 8), TPB, 0, c->stream>>>(g, c->rho, c->grad, fs, mxyz, rcap, bad, stride);
            k_box_pick<<<1, XB_BOXES_MAX, 0, c->stream>>>(fs, seeds, mxyz, rcap, bad, stride, box_max, bx, br);
            k_brick_seed_dev<<<(nbr + 255) / 256, 256, 0, c->stream>>>(gl, nb0, nb1, nb2, fs, bx, br, seed, buf0);
        }
        // the worst-case schedule; after a chase only the kill iteration is left, which dies out within a few bricks of the
        // dividing surfaces: a short schedule first, and a repeat of the whole assignment with the long one (FS_GROW_RETRY)
        // for the rare density whose cascade runs deeper
        const int long_schedule = 2 * ((std::max(std::max(nb0, nb1), nb2) + BG - 1) / BG) + 12;
        const int launches = chase ? std::min(long_schedule, c->grow_kill_launches) : long_schedule;
        const dim3 ggrid((nb2 + BG - 1) / BG, (nb1 + BG - 1) / BG, (nb0 + BG - 1) / BG);
        for (int l = 0; l < launches; l++)   // each returns at once when the growth has finished (phase on the device)
            k_brick_grow_dev<<<ggrid, BG * BG * BG, 0, c->stream>>>(nb0, nb1, nb2, bmask, seed, buf0, buf1, fs, BG, sparse ? 0 : 1);
        if (chase && launches < long_schedule) k_grow_verdict<<<1, 1, 0, c->stream>>>(fs);
        k_fill<int><<<64, 256, 0, c->stream>>>(box_first, XB_INT_MAX, sparse ? XB_REGIONS_MAX : XB_BOXES_MAX);
        k_grow_finish<<<64, TPB, 0, c->stream>>>(nbr, seed, buf0, buf1, fs, c->blab_buf, box_first, bmask, sparse ? c->brick_rec : nullptr,
                                                 sparse ? 0 : 1);
        HIPCHK(hipGetLastError());
    }
    c->blab = c->blab_buf;
    c->walk = walk;
    c->nbk[0] = nb0; c->nbk[1] = nb1; c->nbk[2] = nb2;
    const long long own = c->N;
    {   // region fill / notes, then the walkers of the uncertain bricks
        ScopedTimer t0(c, 0);
        if (c->opt_morton) {
            int bits = 0;
            while ((1 << bits) < std::max(std::max(nb0, nb1), nb2)) bits++;
            const unsigned n_codes = 1u << (3 * bits);
            k_brick_walk_list_morton<<<(n_codes + 16 * TPB - 1) / (16 * TPB), TPB, 0, c->stream>>>(nb0, nb1, nb2, n_codes, c->blab, walk,
                                                                                                  fs + FS_N_WALK, fs + FS_GROW_RETRY);
        } else
            k_brick_walk_list<<<(nbr + 16 * TPB - 1) / (16 * TPB), TPB, 0, c->stream>>>(nbr, 0, nbr, c->blab, walk, fs + FS_N_WALK, fs + FS_GROW_RETRY);
        if (sparse) {   // pass B: records for the bricks of the walk list only
            ScopedTimer t7(c, 7);
            const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
            GridS gs;
            if (sym_grid(g, gs))
                k_brick_records<GridS><<<4096, TPB, 0, c->stream>>>(gs, c->rho, c->grad, walk, fs + FS_N_WALK, nbr, nb1, nb2, c->brick_rec, small);
            else
                k_brick_records<Grid><<<4096, TPB, 0, c->stream>>>(g, c->rho, c->grad, walk, fs + FS_N_WALK, nbr, nb1, nb2, c->brick_rec, small);
        }
        if (c->has_vacuum)
            k_fill_certain<<<nblocks(own), TPB, 0, c->stream>>>(gl, c->blab, nb1, nb2, box_max, c->labels, c->first, c->max_list,
                                                                fs + FS_N_MAX, c->max_cap, fs + FS_GROW_RETRY);
        else {
            k_note_regions<<<1, XB_BOXES_MAX, 0, c->stream>>>(gl, nb1, nb2, fs, box_first, box_max, c->first, c->max_list, fs + FS_N_MAX, c->max_cap);
            c->regions_pending = true;
        }
        {
            ScopedTimer t6(c, 6);
            const int maxsteps = 8 * (g.nx + g.ny + g.nz) + 64;
            // the lean walker needs 24-bit index products and nothing else the fused path does not already guarantee (whole-grid
            // table window, brick-label regions); 32-bit table offsets up to 2^27 voxels
            const int lean = (gl.use24 && c->opt_lean) ? (c->N <= (1LL << 27) ? 2 : 1) : 0;
            const int gw = c->opt_trace_group;   // waves per workgroup (1: one-wave workgroups, every wave pulls for itself)
#define XB_TRACE_ARGS gl, c->grad, box_max, c->blab, nb1, nb2, walk, fs, c->labels, c->first, c->max_list, c->max_cap, c->ovf_list, c->ovf_cap, \
                      maxsteps, c->has_vacuum ? 1 : 0
            if (gw > 1) {
                const int groups = std::max(1, c->opt_trace_grid / gw), ch = std::max(8, c->opt_trace_chunk);
                if (lean && gw == 8 && ch == 8 && c->opt_trace_cache) {   // one brick per pull: its records go through LDS
                    if (lean == 2) k_ng_trace_g<2, 4><<<groups, XB_WAVE * gw, 0, c->stream>>>(XB_TRACE_ARGS, ch, c->opt_trace_xcd);
                    else k_ng_trace_g<2, 3><<<groups, XB_WAVE * gw, 0, c->stream>>>(XB_TRACE_ARGS, ch, c->opt_trace_xcd);
                } else
                if (lean == 2) k_ng_trace_g<2, 2><<<groups, XB_WAVE * gw, 0, c->stream>>>(XB_TRACE_ARGS, ch, c->opt_trace_xcd);
                else if (lean == 1) k_ng_trace_g<2, 1><<<groups, XB_WAVE * gw, 0, c->stream>>>(XB_TRACE_ARGS, ch, c->opt_trace_xcd);
                else k_ng_trace_g<2, 0><<<groups, XB_WAVE * gw, 0, c->stream>>>(XB_TRACE_ARGS, ch, c->opt_trace_xcd);
            } else if (lean == 2)
                k_ng_trace_p<2, 2><<<c->opt_trace_grid, XB_WAVE, 0, c->stream>>>(XB_TRACE_ARGS, c->opt_trace_chunk, c->opt_trace_xcd);
            else if (lean == 1)
                k_ng_trace_p<2, 1><<<c->opt_trace_grid, XB_WAVE, 0, c->stream>>>(XB_TRACE_ARGS, c->opt_trace_chunk, c->opt_trace_xcd);
            else
                k_ng_trace_p<2, 0><<<c->opt_trace_grid, XB_WAVE, 0, c->stream>>>(XB_TRACE_ARGS, c->opt_trace_chunk, c->opt_trace_xcd);
#undef XB_TRACE_ARGS
        }
        HIPCHK(hipGetLastError());
    }
    // numbering + relabel on the device (skipped by their gate when the numbering has to be done on the host)
    k_number_maxima<<<1, 1024, 0, c->stream>>>(fs, c->first, c->max_list, c->max_cap, c->max_aux);
    int *buni = reinterpret_cast<int *>(c->st);
    if (c->regions_pending) {
        if (g.nz % 4 == 0)
            k_relabel_regions_brick<<<dim3((g.nz / 4 + 63) / 64, nb1, (g.nx + 3) / 4), TPB, 0, c->stream>>>(gl, c->labels, c->first, c->blab, nb1, nb2,
                                                                                                        box_max, fs, fs + FS_SORT_OK);
        else
            k_relabel_regions<<<nblocks(own), TPB, 0, c->stream>>>(gl, c->labels, c->first, c->blab, nb1, nb2, box_max, fs + FS_SORT_OK);
        k_buni_from_regions<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, c->blab, box_max, c->first, buni, fs + FS_SORT_OK);
        k_label_uniform_list<<<2048, TPB, 0, c->stream>>>(gl, c->labels, nb1, nb2, walk, 0, fs + FS_N_WALK, fs + FS_SORT_OK, buni);
    } else
        k_relabel<<<nblocks(own), TPB, 0, c->stream>>>(g, c->labels, c->first, fs + FS_SORT_OK);
    k_reset_first_dev<<<8, 256, 0, c->stream>>>(c->first, c->max_aux, fs + FS_N_MAX, fs + FS_SORT_OK);
    HIPCHK(hipGetLastError());
    // the ONE host wait of the assignment: state block + the sorted maxima
    HIPCHK(hipMemcpyAsync(c->host_ints, fs, FS_COUNT * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->host_ints + FS_COUNT, c->max_aux, XB_SORT_MAX * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const int *h = c->host_ints;
    g.main_ties = 0;
    if (h[FS_GROW_RETRY]) {   // the short kill schedule did not reach the fixpoint: once more, with the worst-case one from now on
        c->grow_kill_launches = 1 << 20;
        c->stat_grow_retries++;
        c->grad_valid = false;
        return assign_neargrid_fused(c, n_maxima);
    }
    if (h[FS_TIES] == 0) c->grad_rule = 2;
    c->n_boxes = h[FS_N_BOXES];
    c->box_voxels = (long long)h[FS_N_CERTAIN] * BRK * BRK * BRK;
    c->n_walk = h[FS_N_WALK];
    const int novf = h[FS_N_OVF];
    int nmax = h[FS_N_MAX];
    if (novf > c->ovf_cap) return fail(XB_E_LIMIT, "%d trajectories need the slow path (cap %d)", novf, c->ovf_cap);
    if (nmax > c->max_cap) return fail(XB_E_LIMIT, "%d maxima exceed the table capacity %d", nmax, c->max_cap);
    c->stat_ovf_assign += novf;
    if (h[FS_SORT_OK] && novf == 0) {
        c->maxima_sorted.assign(h + FS_COUNT, h + FS_COUNT + nmax);
        c->regions_pending = false;
        c->buni_valid = !c->has_vacuum;   // k_buni_from_regions + k_label_uniform_list ran
        c->regions_labels = !c->has_vacuum;
        c->first_clean = true;
        if (n_maxima) *n_maxima = nmax;
        return XB_OK;
    }
    // rare: trajectories for the exact slow kernel and/or more maxima than the device sort takes
    if (novf > 0) {
        g.main_ties = 1;
        const int rc = run_slow(c, novf, 0, fs + FS_N_MAX);
        g.main_ties = 0;
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(c->host_ints, fs + FS_N_MAX, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        nmax = c->host_ints[0];
        if (nmax > c->max_cap) return fail(XB_E_LIMIT, "%d maxima exceed the table capacity %d", nmax, c->max_cap);
    }
    return finish_numbering_on_host(c, nmax, n_maxima);
}

// maxima table -> host, sort by first voxel, rank + relabel (the tail of the host-driven path)
static int sort_and_finish(xb_ctx *c, int64_t n, int64_t *n_maxima);
static int finish_numbering_on_host(xb_ctx *c, int nmax, int64_t *n_maxima) {
    c->local_max.resize(nmax);
    c->local_first.resize(nmax);
    if (nmax) {
        k_gather_first<<<(nmax + 255) / 256, 256, 0, c->stream>>>(c->first, c->max_list, nmax, c->max_aux);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(c->local_max.data(), c->max_list, nmax * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(c->local_first.data(), c->max_aux, nmax * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return sort_and_finish(c, nmax, n_maxima);
}

int xb_assign(xb_ctx *c, int method, int64_t *n_maxima) {
    NEED_GRID_RAW("xb_assign");
    if (method == XB_METHOD_NEARGRID && fused_ok(c)) {
        if (c->has_vacuum) { if (int rc = settle_labels(c)) return rc; }
        else c->labels_zero_pending = false;   // every label is overwritten, none is read
        return assign_neargrid_fused(c, n_maxima);
    }
    if (method == XB_METHOD_ONGRID && !c->has_vacuum) c->labels_zero_pending = false;   // the pointer pass writes every label
    int64_t n = 0;
    if (int rc = xb_assign_trace(c, method, &n)) return rc;
    return sort_and_finish(c, n, n_maxima);
}
static int sort_and_finish(xb_ctx *c, int64_t n, int64_t *n_maxima) {
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

// the sweep's launches; the edge count stays on the device (counters[5]).  *dilate_owned: the owned edges still have to
// dilate from the list (k_edge_dilate_list over counters[5] entries)
static int edge_find_launch(xb_ctx *c, bool *dilate_owned) {
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
            const int nbr = (int)(c->N / 512), per_plane = (g.ny / 8) * (g.nz / 8);
            if (!c->buni_valid) {
                // a slab only scans the bricks its sweep can look at (the swept planes +- one brick)
                int b_off = 0, count = nbr;
                if (!all && np + 32 < g.nx) {
                    const int p0 = ((xa - 8) % g.nx + g.nx) % g.nx;
                    b_off = (p0 / 8) * per_plane;
                    count = ((np + 8 + 7 + (p0 % 8)) / 8 + 1) * per_plane;
                }
                k_label_uniform<<<(unsigned)count, TPB, 0, c->stream>>>(gl, c->labels, g.ny / 8, g.nz / 8, buni, b_off, nbr);
                c->buni_halo_safe = false;
            }
            k_buni3<<<(nbr + 255) / 256, 256, 0, c->stream>>>(g.nx / 8, g.ny / 8, g.nz / 8, buni, buni + nbr);
            buni += nbr;   // the sweep reads the 27-brick version
        }
        const GradRec *G = c->grad_valid ? c->grad : nullptr;
        const unsigned char *brec = c->grad_valid && c->grad_cover == 1 ? c->brick_rec : nullptr;
        if (whole || all) {
            dim3 grid((g.nz + ET_Z - 1) / ET_Z, (g.ny + ET_Y - 1) / ET_Y, (np + ET_X - 1) / ET_X);
            k_edge_flag_tiled<<<grid, TPB, 0, c->stream>>>(gl, c->rho, c->labels, c->known, xa, np, c->list,
                                                           c->counters + 5, small, buni, G, brec, c->has_vacuum ? 0 : 1);
            if (!whole)
                k_edge_dilate<<<nblocks((long long)npd * g.nyz), TPB, 0, c->stream>>>(g, c->known, xb_, npd, -2);
        } else if (buni && g.x0 % ET_X == 0 && g.x1 % ET_X == 0 && g.ny % ET_Y == 0 &&
                   (g.x1 - g.x0) + 2 * ((c->halo - 1 + ET_X - 1) / ET_X * ET_X) <= g.nx) {
            // a slab of whole bricks: as on one GPU only the tiles that are not of one label with their surroundings are swept
            // (`known` preset to 2 on the swept planes).  The owned planes' edges make the list; the halo planes each side
            // (rounded out to whole tiles: the extra planes lie beyond the ones whose flags anything reads) give a second
            // list (in `stage`, length on the device) that only serves the dilation (refinement.py:385-404)
            const int own = g.x1 - g.x0, side4 = (c->halo - 1 + ET_X - 1) / ET_X * ET_X;
            const int nty = g.ny / ET_Y, ntz = (g.nz + ET_Z - 1) / ET_Z;
            const int left0 = ((g.x0 - side4) % g.nx + g.nx) % g.nx, right0 = g.x1 % g.nx;
            auto preset = [&](int p0, int np_) -> int {
                const int run1 = std::min(np_, g.nx - p0);
                HIPCHK(hipMemsetAsync(c->known + (size_t)p0 * g.nyz, 2, (size_t)run1 * g.nyz, c->stream));
                if (np_ > run1) HIPCHK(hipMemsetAsync(c->known, 2, (size_t)(np_ - run1) * g.nyz, c->stream));
                return XB_OK;
            };
            if (int rc = preset(left0, side4 + own + side4)) return rc;
            const int n_own = (own / ET_X) * nty * ntz, n_halo = 2 * (side4 / ET_X) * nty * ntz;
            int *tiles_own = (int *)c->stage, *tiles_halo = tiles_own + n_own, *halo_list = tiles_halo + n_halo;
            HIPCHK(hipMemsetAsync(c->counters + 6, 0, sizeof(int), c->stream));
            HIPCHK(hipMemsetAsync(c->counters + 22, 0, 2 * sizeof(int), c->stream));
            GridL ga = gl;
            ga.x0 = 0; ga.x1 = g.nx;    // (lists every edge of the planes it sweeps)
            k_edge_tile_list<<<(n_own + TPB - 1) / TPB, TPB, 0, c->stream>>>(gl, buni, tiles_own, c->counters + 22, g.x0 / ET_X, own / ET_X);
            k_edge_tile_list<<<(n_halo / 2 + TPB - 1) / TPB, TPB, 0, c->stream>>>(gl, buni, tiles_halo, c->counters + 23, left0 / ET_X, side4 / ET_X);
            k_edge_tile_list<<<(n_halo / 2 + TPB - 1) / TPB, TPB, 0, c->stream>>>(gl, buni, tiles_halo, c->counters + 23, right0 / ET_X, side4 / ET_X);
            k_edge_flag_listed<<<n_own, TPB, 0, c->stream>>>(gl, c->rho, c->labels, c->known, c->list, c->counters + 5, small, G, brec,
                                                            c->has_vacuum ? 0 : 1, tiles_own, c->counters + 22);
            k_edge_flag_listed<<<n_halo, TPB, 0, c->stream>>>(ga, c->rho, c->labels, c->known, halo_list, c->counters + 6, small, G, brec,
                                                             c->has_vacuum ? 0 : 1, tiles_halo, c->counters + 23);
            k_edge_dilate_list<<<2048, TPB, 0, c->stream>>>(gl, c->known, halo_list, 0, c->counters + 6);
        } else {
            // a slab: the owned planes (their edges make the list), then the halo planes each side -- their edges go to a
            // second list (in `stage`, length on the device) that only serves the dilation (refinement.py:385-404)
            const int own = g.x1 - g.x0, side = c->halo - 1;
            int *halo_list = (int *)c->stage;
            HIPCHK(hipMemsetAsync(c->counters + 6, 0, sizeof(int), c->stream));
            GridL ga = gl;
            ga.x0 = 0; ga.x1 = g.nx;    // (lists every edge of the planes it sweeps)
            dim3 grid((g.nz + ET_Z - 1) / ET_Z, (g.ny + ET_Y - 1) / ET_Y, (own + ET_X - 1) / ET_X);
            k_edge_flag_tiled<<<grid, TPB, 0, c->stream>>>(gl, c->rho, c->labels, c->known, g.x0, own, c->list,
                                                           c->counters + 5, small, buni, G, brec, c->has_vacuum ? 0 : 1);
            grid.z = (side + ET_X - 1) / ET_X;
            k_edge_flag_tiled<<<grid, TPB, 0, c->stream>>>(ga, c->rho, c->labels, c->known, xa, side, halo_list,
                                                           c->counters + 6, small, buni, G, brec, c->has_vacuum ? 0 : 1);
            k_edge_flag_tiled<<<grid, TPB, 0, c->stream>>>(ga, c->rho, c->labels, c->known, g.x1 % g.nx, side, halo_list,
                                                           c->counters + 6, small, buni, G, brec, c->has_vacuum ? 0 : 1);
            k_edge_dilate_list<<<2048, TPB, 0, c->stream>>>(gl, c->known, halo_list, 0, c->counters + 6);
        }
    }
    HIPCHK(hipGetLastError());
    *dilate_owned = whole || !all;
    return XB_OK;
}
int xb_edge_find(xb_ctx *c, int64_t *edges) {
    NEED_GRID("xb_edge_find");
    const Grid &g = c->g;
    bool dilate_owned = false;
    if (int rc = edge_find_launch(c, &dilate_owned)) return rc;
    int n = 0;
    if (int rc = read_counter(c, 5, &n)) return rc;
    if (dilate_owned && n) {  // the list holds every owned edge: dilate from it
        ScopedTimer t(c, 2);
        k_edge_dilate_list<<<nblocks(n), TPB, 0, c->stream>>>(light(g), c->known, c->list, n, nullptr);
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
    k_compact_known16<<<nblocks((own + 16 * CK_CHUNKS - 1) / (16 * CK_CHUNKS)), TPB, 0, c->stream>>>(light(g), c->known, value, c->list,
                                                                                               c->counters + 5);
    HIPCHK(hipGetLastError());
    return read_counter(c, 5, n_out);
}

// ---- remote path queries (slab scheduler) -------------------------------------------------------
// A retrace that left the valid planes of its rank was parked (known == -6).  Its path depends on rho
// only (replicated), so the owner can record it in full; which voxel of the path stops the retrace
// (the first known == 2 one, refinement.py:294-303) is then asked of the ranks that own those voxels.
int xb_escaped_paths(xb_ctx *c, int64_t max_len, int64_t *n_paths, int64_t *n_voxels) {
    NEED_GRID("xb_escaped_paths");
    c->esc_starts.clear(); c->esc_offsets.assign(1, 0); c->esc_vox.clear(); c->esc_complete.clear();
    c->g.main_ties = 0;   // retraces follow refinement.py's rule
    int n = 0;
    if (int rc = compact(c, -6, &n)) return rc;
    c->list_valid = false;
    if (n) {
        if (max_len < 2 || max_len > (1 << 15)) return fail(XB_E_ARG, "xb_escaped_paths: max_len out of range");
        const int lmax = (int)max_len, chunk = (int)std::max<int64_t>(256, std::min<int64_t>(8192, (32LL << 20) / max_len));
        DevBuf<int> bpath, blen;
        HIPCHK(bpath.alloc((size_t)chunk * lmax));
        HIPCHK(blen.alloc(3 * (size_t)chunk));   // lengths, first out-of-range indices, offsets
        int *path = bpath.p, *dlen = blen.p, *packed = nullptr;
        std::vector<int> starts(n), len(2 * chunk), off(chunk), buf;
        HIPCHK(hipMemcpyAsync(starts.data(), c->list, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        int rc = XB_OK;
        for (int o = 0; o < n && rc == XB_OK; o += chunk) {
            const int m = std::min(chunk, n - o);
            int *dfirst = dlen + m, *doff = dlen + 2 * chunk;
            k_trace_slow<<<(m + 63) / 64, 64, 0, c->stream>>>(c->g, c->rho, c->labels, c->known, c->known, c->list + o, m, path,
                                                             lmax, 2, c->first, c->max_list, c->counters + 0, c->max_cap,
                                                             c->counters + 2, c->counters + 3, c->counters + 8, dlen);
            hipError_t e = hipMemcpyAsync(len.data(), dlen, 2 * m * sizeof(int), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) { rc = fail(XB_E_HIP, "xb_escaped_paths: %s", hipGetErrorString(e)); break; }
            int total = 0;
            std::vector<int> alen(m);
            for (int i = 0; i < m; i++) {
                alen[i] = std::abs(len[i]);          // negative: cut at max_len
                off[i] = total;
                total += 1 + alen[i] - len[m + i];   // start voxel + the part from the first out-of-range voxel on
            }
            buf.resize(total);
            e = hipMalloc(&packed, (size_t)std::max(total, 1) * sizeof(int));
            if (e == hipSuccess) e = hipMemcpyAsync(doff, off.data(), m * sizeof(int), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(dlen, alen.data(), m * sizeof(int), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) {
                k_path_pack<<<m, 64, 0, c->stream>>>(path, lmax, doff, dlen, dfirst, packed);
                e = hipMemcpyAsync(buf.data(), packed, (size_t)total * sizeof(int), hipMemcpyDeviceToHost, c->stream);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            hipFree(packed); packed = nullptr;
            if (e != hipSuccess) { rc = fail(XB_E_HIP, "xb_escaped_paths: %s", hipGetErrorString(e)); break; }
            for (int i = 0; i < m; i++) {
                c->esc_starts.push_back(starts[o + i]);
                const int cnt = 1 + alen[i] - len[m + i];
                for (int k = 0; k < cnt; k++) c->esc_vox.push_back(buf[off[i] + k]);
                c->esc_offsets.push_back((int64_t)c->esc_vox.size());
                c->esc_complete.push_back(len[i] > 0 ? 1 : 0);
            }
        }
        if (rc != XB_OK) return rc;
    }
    if (n_paths) *n_paths = (int64_t)c->esc_starts.size();
    if (n_voxels) *n_voxels = (int64_t)c->esc_vox.size();
    return XB_OK;
}
int xb_escaped_paths_fetch(xb_ctx *c, int64_t *starts, int64_t *offsets, int64_t *voxels, int8_t *complete) {
    if (!c) return fail(XB_E_ARG, "null ctx");
    std::copy(c->esc_starts.begin(), c->esc_starts.end(), starts);
    std::copy(c->esc_offsets.begin(), c->esc_offsets.end(), offsets);
    std::copy(c->esc_vox.begin(), c->esc_vox.end(), voxels);
    std::copy(c->esc_complete.begin(), c->esc_complete.end(), complete);
    return XB_OK;
}
// labels / known at arbitrary voxels (linear indices), and the write-back of retrace results
static int voxel_io(xb_ctx *c, const int64_t *idx, int64_t n, int32_t *lab, int8_t *kn, bool scatter) {
    NEED_GRID("xb_gather_voxels");
    if (n <= 0) return XB_OK;
    std::vector<int> i32(n);
    for (int64_t k = 0; k < n; k++) {
        if (idx[k] < 0 || idx[k] >= c->N) return fail(XB_E_ARG, "voxel index out of range");
        i32[k] = (int)idx[k];
    }
    DevBuf<int> buf;
    HIPCHK(buf.alloc(2 * (size_t)n + (size_t)n / 4 + 1));   // indices, labels, known bytes
    int *d = buf.p;
    int *dlab = d + n;
    int8_t *dkn = reinterpret_cast<int8_t *>(d + 2 * n);
    hipError_t e = hipMemcpyAsync(d, i32.data(), n * sizeof(int), hipMemcpyHostToDevice, c->stream);
    if (scatter) {
        if (e == hipSuccess) e = hipMemcpyAsync(dlab, lab, n * sizeof(int), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dkn, kn, n, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) k_scatter_voxels<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(d, (int)n, dlab, dkn, c->labels, c->known);
        c->list_valid = false;
        c->buni_valid = false; c->regions_labels = false;
        c->zero_outside[0] = -1;
    } else {
        if (e == hipSuccess) k_gather_voxels<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(d, (int)n, c->labels, c->known, dlab, dkn);
        if (e == hipSuccess) e = hipMemcpyAsync(lab, dlab, n * sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(kn, dkn, n, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(XB_E_HIP, "xb_gather/scatter_voxels: %s", hipGetErrorString(e));
    return XB_OK;
}
int xb_gather_voxels(xb_ctx *c, const int64_t *idx, int64_t n, int32_t *labels_out, int8_t *known_out) {
    return voxel_io(c, idx, n, labels_out, known_out, false);
}
int xb_scatter_voxels(xb_ctx *c, const int64_t *idx, int64_t n, const int32_t *labels_in, const int8_t *known_in) {
    return voxel_io(c, idx, n, const_cast<int32_t *>(labels_in), const_cast<int8_t *>(known_in), true);
}

static int refine_trace_impl(xb_ctx *c, int flag, int64_t *changed, int64_t *escaped);
int xb_refine_trace(xb_ctx *c, int64_t *changed, int64_t *escaped) { return refine_trace_impl(c, -2, changed, escaped); }
int xb_refine_trace_escaped(xb_ctx *c, int64_t *changed, int64_t *escaped) { return refine_trace_impl(c, -6, changed, escaped); }
static int refine_trace_impl(xb_ctx *c, int flag, int64_t *changed, int64_t *escaped) {
    NEED_GRID("xb_refine_trace");
    const Grid &g = c->g;
    c->g.main_ties = 0;
    int n = 0;
    if (c->list_valid && flag == -2) n = c->list_n;
    else if (int rc = compact(c, flag, &n)) return rc;
    c->list_valid = false;  // the retrace rewrites known
    c->buni_valid = false;  // ... and may relabel edge voxels; st is also edge_check's scratch
    c->walk_n_out = 0; c->walk_n_res = 0; c->walk_out_dev = nullptr;
    c->walk_host.clear(); c->res_host.clear();
    HIPCHK(hipMemsetAsync(c->counters, 0, 4 * sizeof(int), c->stream));
    int n_changed = 0, n_escaped = 0;
    if (n) {
        const int maxsteps = 8 * (g.nx + g.ny + g.nz) + 64;
        Walker *wio_out = nullptr;
        int wio_cap = 0;
        if (int rc = ensure_grad(c, false, false, false)) return rc;
        {
            ScopedTimer t(c, 3);
            const unsigned char *brec = c->grad_cover == 1 ? c->brick_rec : nullptr;
            const int regions_ok = brec && c->regions_labels && !c->has_vacuum ? 1 : 0;
            // slabs: the regions' brick labels stop a retrace when the labels are this assignment's, there is no vacuum and
            // the density has no tie voxel (the windowed masks are built under the assignment's tie rule only)
            const int *slab_regions = (table_windowed(c) && c->blab && c->regions_labels && !c->has_vacuum && (c->grad_rule == 2 || c->slab_sparse) &&
                                       g.nx % 8 == 0 && g.ny % 8 == 0 && g.nz % 8 == 0) ? c->blab : nullptr;
            // the lean kernel; the retraces whose walk goes on through a voxel without a record (a brick without records,
            // a voxel outside the table window of a slab) or out of the valid planes of a slab are redone by the from-rho
            // kernel (their count stays on the device: its grid strides over it).  On a slab that kernel parks the
            // retraces that leave the valid planes AND exports them as walkers (xb_walkers_*).
            const bool slab = g.vlen < g.nx;
            int *defer = (int *)c->stage;
            WalkerIO wio{};
            if (flag == -2 && slab) {
                const size_t off = (((size_t)n * sizeof(int)) + 255) & ~(size_t)255;
                if (off + sizeof(Walker) <= c->stage_bytes) {
                    HIPCHK(hipMemsetAsync(c->counters + 16, 0, sizeof(int), c->stream));
                    wio.out = (Walker *)((char *)c->stage + off); wio.out_count = c->counters + 16;
                    wio.out_cap = (int)std::min<size_t>((c->stage_bytes - off) / sizeof(Walker), 1u << 30);
                    c->walk_out_dev = wio.out;
                }
            }
            HIPCHK(hipMemsetAsync(c->counters + 15, 0, sizeof(int), c->stream));
            k_refine_trace<2, false><<<nblocks(n), TPB, 0, c->stream>>>(light(g), c->grad, c->labels, c->known, c->list, n, nullptr,
                                                                        c->counters + 2, c->counters + 3, c->ovf_list, c->counters + 1,
                                                                        c->ovf_cap, maxsteps, c->rho, c->dist_dev, brec, defer,
                                                                        c->counters + 15, regions_ok, slab_regions, WalkerIO{});
            if (brec || slab || table_windowed(c))
                k_refine_trace<2, true><<<512, TPB, 0, c->stream>>>(light(g), c->grad, c->labels, c->known, defer, 0, c->counters + 15,
                                                                    c->counters + 2, c->counters + 3, c->ovf_list, c->counters + 1,
                                                                    c->ovf_cap, maxsteps, c->rho, c->dist_dev, brec, nullptr, nullptr, 0,
                                                                    slab_regions, wio);
            wio_out = wio.out; wio_cap = wio.out_cap;
        }
        HIPCHK(hipGetLastError());
        // one wait for everything the kernels counted: overflows [1], changed [2], escaped [3], exported walkers [16]
        HIPCHK(hipMemcpyAsync(c->host_ints, c->counters, 17 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        const int novf = c->host_ints[1];
        n_changed = c->host_ints[2]; n_escaped = c->host_ints[3];
        if (wio_out) {
            c->walk_n_out = std::min(c->host_ints[16], wio_cap);
            c->walk_host.resize((size_t)c->walk_n_out * (sizeof(Walker) / 8));
            if (int rc = download_pinned(c, c->walk_host.data(), wio_out, (size_t)c->walk_n_out * sizeof(Walker))) return rc;
        }
        if (novf > c->ovf_cap) return fail(XB_E_LIMIT, "%d retraces need the slow path (cap %d)", novf, c->ovf_cap);
        c->stat_ovf_refine += novf;
        if (novf > 0) {
            if (int rc = run_slow(c, novf, 1)) return rc;
            HIPCHK(hipMemcpyAsync(c->host_ints, c->counters + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            n_changed = c->host_ints[0]; n_escaped = c->host_ints[1];
        }
    }
    if (changed) *changed = n_changed;
    if (escaped) *escaped = n_escaped;
    return XB_OK;
}

// ---- walkers: retraces that left this rank's valid planes, carried on by the rank that owns the plane they entered ----
static int ensure_walker_bufs(xb_ctx *c, int64_t n) {
    if (c->walk_cap >= n) return XB_OK;
    (void)hipFree(c->walk_in); (void)hipFree(c->walk_out2); (void)hipFree(c->walk_res);
    c->walk_in = c->walk_out2 = c->walk_res = nullptr; c->walk_cap = 0;
    const size_t cap = (size_t)n + n / 2 + 4096;
    HIPCHK(hipMalloc(&c->walk_in, cap * sizeof(Walker)));
    HIPCHK(hipMalloc(&c->walk_out2, cap * sizeof(Walker)));
    HIPCHK(hipMalloc(&c->walk_res, cap * 2 * sizeof(int)));
    c->walk_cap = (long long)cap;
    return XB_OK;
}
int xb_walkers_count(xb_ctx *c, int64_t *n_walkers, int64_t *n_results) {
    NEED_GRID("xb_walkers_count");
    if (n_walkers) *n_walkers = c->walk_n_out;
    if (n_results) *n_results = c->walk_n_res;
    return XB_OK;
}
int xb_walkers_fetch(xb_ctx *c, int64_t *walkers, int64_t *results) {
    NEED_GRID("xb_walkers_fetch");
    if (walkers && c->walk_n_out) memcpy(walkers, c->walk_host.data(), (size_t)c->walk_n_out * sizeof(Walker));
    if (results && c->walk_n_res) memcpy(results, c->res_host.data(), (size_t)c->walk_n_res * sizeof(int64_t));
    return XB_OK;
}
// `walkers`: n records of XB_WALKER_WORDS int64 (every rank's exports, any order).  The ones that arrive on a plane this
// rank owns are carried on with this rank's labels / known: results = (start voxel, final label) pairs, the others
// that leave the valid planes again are exported anew (xb_walkers_count / xb_walkers_fetch).
int xb_walkers_continue(xb_ctx *c, const int64_t *walkers, int64_t n) {
    NEED_GRID("xb_walkers_continue");
    if (n < 0 || (n && !walkers)) return fail(XB_E_ARG, "xb_walkers_continue: bad arguments");
    if (c->g.vlen >= c->g.nx) return fail(XB_E_STATE, "xb_walkers_continue: every plane is valid on this rank (no slab halo)");
    const Grid &g = c->g;
    c->walk_n_out = 0; c->walk_n_res = 0;
    c->walk_host.clear(); c->res_host.clear();
    if (!n) return XB_OK;
    if (n > (1 << 28)) return fail(XB_E_LIMIT, "xb_walkers_continue: too many walkers");
    c->g.main_ties = 0;
    if (int rc = ensure_grad(c, false, false, false)) return rc;
    const size_t bytes = (size_t)n * sizeof(Walker);
    if (int rc = ensure_walker_bufs(c, n)) return rc;
    if (int rc = upload_pinned(c, c->walk_in, walkers, bytes)) return rc;
    HIPCHK(hipMemsetAsync(c->counters + 16, 0, 4 * sizeof(int), c->stream));
    HIPCHK(hipMemsetAsync(c->counters + 1, 0, 3 * sizeof(int), c->stream));
    WalkerIO wio{};
    wio.in = (const Walker *)c->walk_in;
    wio.out = (Walker *)c->walk_out2; wio.out_count = c->counters + 16; wio.out_cap = (int)n;
    wio.res = (int *)c->walk_res; wio.res_count = c->counters + 17;
    wio.own0 = g.x0; wio.own1 = g.x1;
    const unsigned char *brec = c->grad_cover == 1 ? c->brick_rec : nullptr;
    const int *slab_regions = (table_windowed(c) && c->blab && c->regions_labels && !c->has_vacuum && (c->grad_rule == 2 || c->slab_sparse) &&
                               g.nx % 8 == 0 && g.ny % 8 == 0 && g.nz % 8 == 0) ? c->blab : nullptr;
    const int maxsteps = 8 * (g.nx + g.ny + g.nz) + 64;
    k_refine_trace<2, true, true><<<nblocks((int)n), TPB, 0, c->stream>>>(light(g), c->grad, c->labels, c->known, nullptr, (int)n, nullptr,
                                                                         c->counters + 2, c->counters + 3, c->ovf_list, c->counters + 1,
                                                                         c->ovf_cap, maxsteps, c->rho, c->dist_dev, brec, nullptr, nullptr, 0,
                                                                         slab_regions, wio);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->host_ints, c->counters + 16, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->walk_n_out = c->host_ints[0]; c->walk_n_res = c->host_ints[1];
    c->walk_out_dev = wio.out;
    c->walk_host.resize((size_t)c->walk_n_out * (sizeof(Walker) / 8));
    std::vector<int> pairs(2 * (size_t)c->walk_n_res);
    if (int rc = download_pinned(c, c->walk_host.data(), wio.out, (size_t)c->walk_n_out * sizeof(Walker))) return rc;
    if (int rc = download_pinned(c, pairs.data(), wio.res, pairs.size() * sizeof(int))) return rc;
    c->res_host.resize(c->walk_n_res);
    for (int i = 0; i < c->walk_n_res; i++)
        c->res_host[i] = (int64_t)(uint32_t)pairs[2 * i] | ((int64_t)pairs[2 * i + 1] << 32);
    return XB_OK;
}
// `results`: n pairs (voxel | label << 32), every rank's.  The pairs whose voxel this rank owns are applied as the retrace
// would have (refinement.py:288-291); stuck ones (the exact slow path is needed) stay parked for xb_escaped_paths.
int xb_walkers_apply(xb_ctx *c, const int64_t *results, int64_t n, int64_t *changed, int64_t *stuck) {
    NEED_GRID("xb_walkers_apply");
    if (changed) *changed = 0;
    if (stuck) *stuck = 0;
    if (n < 0 || (n && !results)) return fail(XB_E_ARG, "xb_walkers_apply: bad arguments");
    if (!n) return XB_OK;
    if (n > (1 << 28)) return fail(XB_E_LIMIT, "xb_walkers_apply: too many results");
    const Grid &g = c->g;
    std::vector<int> pairs(2 * (size_t)n);
    for (int64_t i = 0; i < n; i++) {
        const int64_t v = results[i] & 0xffffffffLL;
        if (v >= c->N) return fail(XB_E_ARG, "xb_walkers_apply: voxel out of range");
        pairs[2 * i] = (int)v; pairs[2 * i + 1] = (int)(results[i] >> 32);
    }
    if (int rc = ensure_walker_bufs(c, n)) return rc;
    int *d = (int *)c->walk_res;    // (the results of the last xb_walkers_continue were fetched to the host already)
    if (int rc = upload_pinned(c, d, pairs.data(), pairs.size() * sizeof(int))) return rc;
    HIPCHK(hipMemsetAsync(c->counters + 18, 0, 2 * sizeof(int), c->stream));
    k_walkers_apply<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(light(g), d, (int)n, g.x0, g.x1, c->labels, c->known,
                                                                      c->counters + 18, c->counters + 19);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->host_ints, c->counters + 18, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->list_valid = false; c->buni_valid = false;
    if (changed) *changed = c->host_ints[0];
    if (stuck) *stuck = c->host_ints[1];
    return XB_OK;
}

// edge_check on the listed voxels c->list[0..n) (all flagged -2 in `known`): the greedy resolution by dependency
// counters (k_edges.h), then apply / restore / ring / finish.  `cls`: per list entry the edge&maximum class computed
// elsewhere (slabs: by the owner of the voxel), or null to derive it here.  Only entries within `near_np` planes
// from plane `near_xa` re-classify their boxes (slabs: the boxes that can touch this rank's valid planes), new
// edges are counted in the linear index range [count_lo, count_hi) (slabs: the owned planes).
static int edge_check_resolve(xb_ctx *c, int n, const int8_t *cls, int near_xa, int near_np, long long count_lo,
                              long long count_hi, int64_t *checked, int64_t *edges) {
    const Grid &g = c->g;
    if (checked) *checked = 0;
    if (edges) *edges = 0;
    if (!n) return XB_OK;
    {
        // counters + classes for the whole list, round 1 over the whole list, then the dependency chains are
        // chased asynchronously by a small grid of workgroups; queue overflows seed another launch.  Scratch: two
        // seed / overflow lists of N ints in the staging buffer, 16 bits per voxel for the counters (only
        // 'changed' refinement needs them).
        // (the seed / overflow lists: in `stage` when it is grid sized, else in a buffer of their own -- at most every listed
        // voxel is queued at once)
        int cap = (int)std::min<long long>(c->N, 1LL << 30);
        int *buf[2] = {(int *)c->stage, (int *)c->stage + c->N};
        if (c->stage_bytes < 8 * (size_t)c->N) {
            cap = (int)std::min<long long>(c->N, std::max<long long>(2LL * n + 65536, 1 << 20));
            if (c->ec_buf_cap < 2LL * cap) {
                hipFree(c->ec_buf); c->ec_buf = nullptr; c->ec_buf_cap = 0;
                HIPCHK(hipMalloc(&c->ec_buf, 2 * (size_t)cap * sizeof(int)));
                c->ec_buf_cap = 2LL * cap;
            }
            buf[0] = c->ec_buf; buf[1] = c->ec_buf + cap;
        }
        if (!c->ec_pend) HIPCHK(hipMalloc(&c->ec_pend, 8 * (size_t)c->N + 16));
        ec_word *pend_w = c->ec_pend;
        HIPCHK(hipMemsetAsync(c->counters + 6, 0, sizeof(int), c->stream));
        if (cls) k_ec_init_cls<<<nblocks(n), TPB, 0, c->stream>>>(g, c->known, c->list, n, cls, pend_w);
        else k_ec_init<<<nblocks(n), TPB, 0, c->stream>>>(g, c->rho, c->labels, c->known, c->list, n, pend_w);
        k_ec_first<<<(unsigned)std::min<long long>(nblocks(n), 4096), TPB, 0, c->stream>>>(g, c->known, pend_w, c->list, n, buf[0],
                                                                                         c->counters + 6, cap);
        HIPCHK(hipGetLastError());
        int n_seeds = 0;
        if (int rc = read_counter(c, 6, &n_seeds)) return rc;
        for (int pass = 0; n_seeds > 0; pass++) {
            if (n_seeds > cap) return fail(XB_E_LIMIT, "xb_edge_check: seed list too small");
            if (pass > 256) return fail(XB_E_LIMIT, "xb_edge_check: queue overflow passes did not drain");
            HIPCHK(hipMemsetAsync(c->counters + 6, 0, sizeof(int), c->stream));
            const int groups = (int)std::min<long long>(std::max(1, n_seeds / 64), c->opt_ec_groups);
            k_ec_chase<<<groups, EC_CHASE_THREADS, 0, c->stream>>>(g, c->known, pend_w, buf[pass & 1], n_seeds, buf[1 - (pass & 1)],
                                                                   c->counters + 6, cap, c->opt_ec_qcap);
            HIPCHK(hipGetLastError());
            const int before = n_seeds;
            if (int rc = read_counter(c, 6, &n_seeds)) return rc;
            if (c->opt_dbg & 4) fprintf(stderr, "edge_check pass %d: %d seeds, %d overflowed (%d groups)\n", pass, before, n_seeds, groups);
        }
    }
    HIPCHK(hipMemsetAsync(c->counters + 6, 0, sizeof(int), c->stream));
    k_ec_collect<<<nblocks(n), TPB, 0, c->stream>>>(c->known, c->list, n, c->st, c->counters + 6);
    {
        int undecided = 0;
        if (int rc = read_counter(c, 6, &undecided)) return rc;
        if (undecided) return fail(XB_E_STATE, "xb_edge_check: %d edge voxels left undecided", undecided);
    }
    if (near_np < g.nx) k_ec_keep_near<<<nblocks(n), TPB, 0, c->stream>>>(g, c->list, n, c->st, near_xa, near_np);
    HIPCHK(hipMemsetAsync(c->counters64, 0, 2 * sizeof(unsigned long long), c->stream));
    const int new_cap = (int)std::min<long long>(c->list_cap - n, 1LL << 30);   // the rest of `list` behind the compacted edges
    HIPCHK(hipMemsetAsync(c->counters + 7, 0, sizeof(int), c->stream));
    k_ec_apply<<<nblocks(n), TPB, 0, c->stream>>>(g, c->rho, c->labels, c->known, c->list, n, c->st, c->counters64 + 1,
                                                  c->list + n, c->counters + 7, new_cap);
    k_ec_restore<<<nblocks(n), TPB, 0, c->stream>>>(c->known, c->list, n);
    {   // -1 ring around the new edges (-3): from their list, or by a full-grid sweep if the list did not fit
        int n_new = 0;
        if (int rc = read_counter(c, 7, &n_new)) return rc;
        if (n_new > new_cap) k_edge_dilate<<<nblocks(c->N), TPB, 0, c->stream>>>(g, c->known, 0, g.nx, -3);
        else if (n_new) k_edge_dilate_list<<<nblocks(n_new), TPB, 0, c->stream>>>(light(g), c->known, c->list + n, n_new, nullptr);
    }
    k_ec_finish<<<(unsigned)std::min<long long>(nblocks((c->N + 15) / 16), 2048), TPB, 0, c->stream>>>(c->known, c->N, c->counters64,
                                                                                                      count_lo, count_hi);
    HIPCHK(hipGetLastError());
    unsigned long long r[2];
    HIPCHK(hipMemcpyAsync(r, c->counters64, sizeof r, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (edges) *edges = (int64_t)r[0];
    if (checked) *checked = (int64_t)(r[1] + r[0]);  // refinement.py:479 + 504
    return XB_OK;
}

int xb_edge_check(xb_ctx *c, int64_t *checked, int64_t *edges) {
    NEED_GRID("xb_edge_check");
    const Grid &g = c->g;
    c->list_valid = false;
    c->buni_valid = false;
    if (g.x1 - g.x0 != g.nx) return fail(XB_E_STATE, "xb_edge_check: one slab only; slabs use xb_edge_check_local + xb_edge_check_global");
    if (c->N > (1LL << 30)) return fail(XB_E_LIMIT, "xb_edge_check: more than 2^30 voxels (queue entries keep two flag bits)");
    int n = 0;
    if (int rc = compact(c, -2, &n)) return rc;
    return edge_check_resolve(c, n, nullptr, 0, g.nx, 0, c->N, checked, edges);
}

// ---- 'changed' refinement across slabs ------------------------------------------------------------------------
// refinement.edge_check is ONE lexicographic greedy scan of the whole grid (refinement.py:420-427): whether a changed
// voxel is processed depends on its C-order earlier changed neighbours, in chains that run through slab boundaries.
// The chains only involve the changed voxels themselves (a few 10^5 at 512^3) and one class bit each, so every rank
// resolves the GLOBAL list: (1) each rank lists its owned changed voxels with their class (xb_edge_check_local);
// (2) the scheduler all-gathers the lists; (3) each rank flags the whole list in its full-size `known`, resolves it
// with the same dependency-counter kernels as one GPU, and applies the boxes that touch its own valid planes
// (xb_edge_check_global).  Needs label AND known halos refreshed beforehand.
int xb_edge_check_local(xb_ctx *c, int64_t *n_out) {
    NEED_GRID("xb_edge_check_local");
    c->list_valid = false;
    int n = 0;
    if (int rc = compact(c, -2, &n)) return rc;   // owned planes only
    if (n) {
        k_ec_class<<<nblocks(n), TPB, 0, c->stream>>>(c->g, c->rho, c->labels, c->list, n, c->st);
        HIPCHK(hipGetLastError());
    }
    c->ec_local_n = n;
    if (n_out) *n_out = n;
    return XB_OK;
}
int xb_edge_check_local_fetch(xb_ctx *c, int64_t *idx_out, int8_t *cls_out) {
    NEED_GRID("xb_edge_check_local_fetch");
    const int n = c->ec_local_n;
    if (!n) return XB_OK;
    std::vector<int> tmp(n);
    HIPCHK(hipMemcpyAsync(tmp.data(), c->list, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(cls_out, c->st, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; i++) idx_out[i] = tmp[i];
    return XB_OK;
}
int xb_edge_check_global(xb_ctx *c, const int64_t *idx, const int8_t *cls, int64_t n, int64_t *checked, int64_t *edges) {
    NEED_GRID("xb_edge_check_global");
    const Grid &g = c->g;
    c->list_valid = false;
    c->buni_valid = false;
    if (checked) *checked = 0;
    if (edges) *edges = 0;
    if (n < 0 || n > c->N) return fail(XB_E_ARG, "xb_edge_check_global: bad list length");
    if (c->N > (1LL << 30)) return fail(XB_E_LIMIT, "xb_edge_check: more than 2^30 voxels (queue entries keep two flag bits)");
    if (c->halo < 3 && g.vlen < g.nx) return fail(XB_E_STATE, "xb_edge_check_global: needs a halo of at least 3 planes");
    if (!n) return XB_OK;
    std::vector<int> i32(n);
    for (int64_t k = 0; k < n; k++) {
        if (idx[k] < 0 || idx[k] >= c->N) return fail(XB_E_ARG, "xb_edge_check_global: voxel index out of range");
        i32[k] = (int)idx[k];
    }
    // planes outside this rank's valid range hold stale flags: neutralise them, then flag the whole global list
    if (g.vlen < g.nx) {
        const int a = g.vx0 + g.vlen;   // invalid planes: [a, a + nx - vlen) modulo nx
        const int len = g.nx - g.vlen, first = a % g.nx, run1 = std::min(len, g.nx - first);
        HIPCHK(hipMemsetAsync(c->known + (size_t)first * g.nyz, 2, (size_t)run1 * g.nyz, c->stream));
        if (len > run1) HIPCHK(hipMemsetAsync(c->known, 2, (size_t)(len - run1) * g.nyz, c->stream));
    }
    int8_t *dcls = c->st + (c->N - n);   // the tail of `st` (its head receives the decisions of k_ec_collect)
    if (2 * n > c->N || 2 * n > c->list_cap) return fail(XB_E_LIMIT, "xb_edge_check_global: list longer than half the grid / the list buffer");
    if (int rc = upload_pinned(c, c->list, i32.data(), n * sizeof(int))) return rc;
    if (int rc = upload_pinned(c, dcls, cls, n, (n * sizeof(int) + 255) & ~(size_t)255)) return rc;
    k_scatter_byte<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(c->known, c->list, (int)n, (int8_t)-2);
    HIPCHK(hipGetLastError());
    // a processed voxel re-classifies its box (one plane each side) and a new edge among those rings its own box
    // (one more plane): voxels within two planes of the known-valid range [vx0, vx0 + vlen) can reach it
    int near_xa = 0, near_np = g.nx;
    if (g.vlen + 4 < g.nx) { near_xa = (g.vx0 - 2 + g.nx) % g.nx; near_np = g.vlen + 4; }
    const int rc = edge_check_resolve(c, (int)n, dcls, near_xa, near_np, (long long)g.x0 * g.nyz, (long long)g.x1 * g.nyz, checked, edges);
    return rc;   // (host vectors outlive the copies: edge_check_resolve waits on the stream)
}

// The retraces of a refinement need the gradient-field table anyway; built before the first edge sweep it also
// lets edge_find read "not a maximum" off the tabulated ongrid successor instead of a 27-point density test
// per edge voxel (2.0 -> 0.9 ms at 512^3 after an ongrid assignment).
int xb_prepare_refine(xb_ctx *c) {
    NEED_GRID("xb_prepare_refine");
    return ensure_grad(c, false, false, false);
}

// edge_find + retrace of one refinement iteration on one slab with ONE host wait: the edge count stays on the
// device (the list kernels stride over it), the counters come back together at the end.
static int refine_iteration_fused(xb_ctx *c, int64_t *edges, int64_t *changed) {
    const Grid &g = c->g;
    int *fs = c->fs;
    c->g.main_ties = 0;
    const GridL gl = light(g);
    HIPCHK(hipMemsetAsync(fs + FS_N_EDGES, 0, 5 * sizeof(int), c->stream));   // edges, changed, escaped, overflows, deferred
    {
        ScopedTimer t(c, 2);
        const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
        int *buni = nullptr;
        if (g.nx % 8 == 0 && g.ny % 8 == 0 && g.nz % 8 == 0) {
            buni = reinterpret_cast<int *>(c->st);
            if (!c->buni_valid)
                k_label_uniform<<<(unsigned)(c->N / 512), TPB, 0, c->stream>>>(gl, c->labels, g.ny / 8, g.nz / 8, buni, 0, (int)(c->N / 512));
            const int nbr = (int)(c->N / 512);
            k_buni3<<<(nbr + 255) / 256, 256, 0, c->stream>>>(g.nx / 8, g.ny / 8, g.nz / 8, buni, buni + nbr);
            buni += nbr;
        }
        const unsigned char *brec = c->grad_valid && c->grad_cover == 1 ? c->brick_rec : nullptr;
        if (buni) {
            // flags preset to "known", then only the tiles that are not of one non-vacuum label with their surroundings
            const int ntiles = ((g.nz + ET_Z - 1) / ET_Z) * (g.ny / ET_Y) * (g.nx / ET_X);
            int *tiles = (int *)c->stage;
            HIPCHK(hipMemsetAsync(c->known, 2, (size_t)c->N, c->stream));
            HIPCHK(hipMemsetAsync(fs + FS_N_TILES, 0, sizeof(int), c->stream));
            k_edge_tile_list<<<(ntiles + TPB - 1) / TPB, TPB, 0, c->stream>>>(gl, buni, tiles, fs + FS_N_TILES, 0, g.nx / ET_X);
            k_edge_flag_listed<<<ntiles, TPB, 0, c->stream>>>(gl, c->rho, c->labels, c->known, c->list, fs + FS_N_EDGES, small,
                                                           c->grad_valid ? c->grad : nullptr, brec, c->has_vacuum ? 0 : 1, tiles, fs + FS_N_TILES);
        } else {
            dim3 grid((g.nz + ET_Z - 1) / ET_Z, (g.ny + ET_Y - 1) / ET_Y, (g.nx + ET_X - 1) / ET_X);
            k_edge_flag_tiled<<<grid, TPB, 0, c->stream>>>(gl, c->rho, c->labels, c->known, 0, g.nx, c->list, fs + FS_N_EDGES, small, buni,
                                                           c->grad_valid ? c->grad : nullptr, brec, c->has_vacuum ? 0 : 1);
        }
        k_edge_dilate_list<<<nblocks(c->N / 16), TPB, 0, c->stream>>>(gl, c->known, c->list, 0, fs + FS_N_EDGES);
    }
    c->list_valid = false;
    c->buni_valid = false;
    {
        ScopedTimer t(c, 3);
        const int maxsteps = 8 * (g.nx + g.ny + g.nz) + 64;
        k_refine_trace<2, false><<<nblocks(c->N / 16), TPB, 0, c->stream>>>(gl, c->grad, c->labels, c->known, c->list, 0, fs + FS_N_EDGES,
                                                              fs + FS_CHANGED, fs + FS_ESCAPED, c->ovf_list, fs + FS_R_OVF,
                                                              c->ovf_cap, maxsteps, c->rho, c->dist_dev,
                                                              c->grad_cover == 1 ? c->brick_rec : nullptr, (int *)c->stage, fs + FS_R_DEFER,
                                                              c->grad_cover == 1 && c->regions_labels && !c->has_vacuum ? 1 : 0, nullptr, WalkerIO{});
        if (c->grad_cover == 1)   // the few retraces whose walk goes on through a brick without records (count on the device)
            k_refine_trace<2, true><<<512, TPB, 0, c->stream>>>(gl, c->grad, c->labels, c->known, (int *)c->stage, 0, fs + FS_R_DEFER,
                                                               fs + FS_CHANGED, fs + FS_ESCAPED, c->ovf_list, fs + FS_R_OVF,
                                                               c->ovf_cap, maxsteps, c->rho, c->dist_dev, c->brick_rec, nullptr, nullptr, 0, nullptr, WalkerIO{});
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->host_ints, fs + FS_N_EDGES, 5 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *edges = c->host_ints[0];
    *changed = c->host_ints[1];
    const int novf = c->host_ints[3];
    c->stat_deferred += c->host_ints[4];
    if (c->host_ints[2]) return fail(XB_E_STATE, "xb_refine: %d traces left the grid", c->host_ints[2]);
    if (novf > c->ovf_cap) return fail(XB_E_LIMIT, "%d retraces need the slow path (cap %d)", novf, c->ovf_cap);
    c->stat_ovf_refine += novf;
    if (novf > 0) {
        if (int rc = run_slow(c, novf, 1, nullptr, fs + FS_CHANGED, fs + FS_ESCAPED)) return rc;
        HIPCHK(hipMemcpyAsync(c->host_ints, fs + FS_CHANGED, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        *changed = c->host_ints[0];
    }
    return XB_OK;
}

int xb_refine(xb_ctx *c, int mode, int64_t iters, int64_t *log, int64_t log_capacity, int64_t *n_iters) {
    NEED_GRID("xb_refine");
    if (n_iters) *n_iters = 0;
    if (iters == 0) return XB_OK;  // thread_handlers.py:146-147
    if (int rc = xb_prepare_refine(c)) return rc;
    int64_t edges = 0, changed = 0, esc = 0, checked = 0;
    const bool fused = c->opt_fused && c->g.x0 == 0 && c->g.x1 == c->g.nx && !table_windowed(c);
    int64_t k = 0;
    auto put = [&](int64_t e, int64_t ch) {
        if (log && 2 * k + 1 < log_capacity) { log[2 * k] = e; log[2 * k + 1] = ch; }
        k++;
        if (n_iters) *n_iters = k;
    };
    if (fused) {
        if (int rc = refine_iteration_fused(c, &edges, &changed)) return rc;
        if (edges == 0) return XB_OK;  // thread_handlers.py:151-153 (no edge: the retrace had nothing to do)
    } else {
        if (int rc = xb_edge_find(c, &edges)) return rc;
        if (edges == 0) return XB_OK;  // thread_handlers.py:151-153
        if (int rc = xb_refine_trace(c, &changed, &esc)) return rc;
        if (esc) return fail(XB_E_STATE, "xb_refine: %lld traces left the valid slab", (long long)esc);
    }
    put(edges, changed);
    for (int64_t it = 2; iters < 0 || it <= iters; it++) {  // thread_handlers.py:194-236
        if (fused && mode != XB_REFINE_ALL && changed == 0) {
            // edge_check re-classifies the boxes of the voxels still flagged -2, i.e. the CHANGED ones
            // (refinement.py:425-427): none is left, so it reports 0 edges and the retrace has no work
            put(0, 0);
            break;
        }
        if (fused && mode == XB_REFINE_ALL) {
            if (int rc = refine_iteration_fused(c, &edges, &changed)) return rc;
            put(edges, changed);
            if (changed == 0) break;
            continue;
        }
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
    DevBuf<double> bch;
    DevBuf<unsigned long long> bcn;
    HIPCHK(bch.alloc(n_labels));
    HIPCHK(bcn.alloc(n_labels));
    double *dch = bch.p;
    unsigned long long *dcn = bcn.p;
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
    if (e != hipSuccess) return fail(XB_E_HIP, "xb_charge_sum: %s", hipGetErrorString(e));
    for (int64_t i = 0; i < n_labels; i++) {
        charge[i] *= voxel_volume;  // utils.py:251-252
        volume[i] = (double)cn[i] * voxel_volume;
    }
    return XB_OK;
}

int xb_volume_assign(xb_ctx *c, const int64_t *swap, int64_t n_swap) {
    NEED_GRID("xb_volume_assign");
    c->zero_outside[0] = -1;
    c->buni_valid = false; c->regions_labels = false;
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
    if (n_max <= 0) return XB_OK;
    if (n_max > (1LL << 30) || n_atoms > (1LL << 24)) return fail(XB_E_LIMIT, "xb_atom_assign: too many maxima / atoms");
    // context free (the reference calls it without a grid): buffers on the current device, default stream
    double *d = nullptr;
    const size_t nd = 3 * (size_t)n_max + 3 * (size_t)n_atoms + 9 + (size_t)n_max;   // maxima, atoms, lattice, distances
    hipError_t e = hipMalloc(&d, nd * sizeof(double) + (size_t)n_max * sizeof(long long));
    if (e != hipSuccess) return fail(XB_E_HIP, "xb_atom_assign: %s", hipGetErrorString(e));
    double *dmax = d, *datoms = d + 3 * n_max, *dlat = datoms + 3 * n_atoms, *ddist = dlat + 9;
    long long *dwho = reinterpret_cast<long long *>(ddist + n_max);
    e = hipMemcpy(dmax, b_max, 3 * n_max * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(datoms, atoms, 3 * n_atoms * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dlat, lattice, 9 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_atom_assign<<<(unsigned)((n_max + 63) / 64), 64>>>(dmax, (int)n_max, datoms, (int)n_atoms, dlat, dwho, ddist);
        e = hipGetLastError();
    }
    static_assert(sizeof(long long) == sizeof(int64_t), "label width");
    if (e == hipSuccess) e = hipMemcpy(atom_out, dwho, n_max * sizeof(long long), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(dist_out, ddist, n_max * sizeof(double), hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) return fail(XB_E_HIP, "xb_atom_assign: %s", hipGetErrorString(e));
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
    if (margin < 0 || own == g.nx) { g.wx0 = 0; g.wlen = g.nx; c->table_margin = -1; g.wbase = 0; return XB_OK; }
    if (g.nx % 8 || g.ny % 8 || g.nz % 8 || g.x0 % 8 || g.x1 % 8)
        return fail(XB_E_ARG, "xb_set_table_window: grid and slab must be made of whole 8^3 bricks");
    const int m8 = (int)((std::max<int64_t>(margin, c->halo) + 7) / 8) * 8;
    if (own + 2 * m8 >= g.nx) { g.wx0 = 0; g.wlen = g.nx; c->table_margin = -1; g.wbase = 0; return XB_OK; }
    g.wx0 = ((g.x0 - m8) % g.nx + g.nx) % g.nx;
    g.wlen = own + 2 * m8;
    c->table_margin = m8;
    return need_grad(c);     // the table shrinks to the window: 32 B per voxel of slab + margins instead of the grid
}
static bool slab_sparse_ok(const xb_ctx *c) {
    const Grid &g = c->g;
    return c->opt_sparse && c->opt_boxes && c->opt_bricks && table_windowed(c) && g.nx % BRK == 0 && g.ny % BRK == 0 && g.nz % BRK == 0 &&
           g.x0 % BRK == 0 && g.x1 % BRK == 0 && g.ny >= 16 && g.nz >= 16 && 7LL * (c->N / (BRK * BRK * BRK)) <= c->N;
}
int xb_table_build(xb_ctx *c, int64_t *n_local_seeds) {
    NEED_GRID_RAW("xb_table_build");   // (no label is read here: a deferred labels := 0 stays deferred)
    if (int rc = need_grad(c)) return rc;
    c->slab_sparse = false;
    if (slab_sparse_ok(c)) {
        // pass A over the OWN planes: move masks, maxima count and the single maximum of every own brick (k_brick_masks);
        // the scheduler shares both arrays, xb_table_finish grows the regions and builds the records of the window
        Grid &g = c->g;
        const int nb1 = g.ny / BRK, nb2 = g.nz / BRK, nbr = (g.nx / BRK) * nb1 * nb2;
        if (int rc = ensure_brick_bytes(c, nbr)) return rc;
        int *fs = c->fs, *bmask = c->list + nbr, *bmaxv = c->list + 4 * nbr;
        HIPCHK(hipMemsetAsync(fs, 0, FS_TOTAL * sizeof(int), c->stream));
        g.main_ties = 1;
        {
            ScopedTimer t4(c, 4);
            ScopedTimer t5(c, 5);
            const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
            dim3 grid((g.nz + GT_Z - 1) / GT_Z, (g.ny + GT_Y - 1) / GT_Y, (g.x1 - g.x0) / GT_X);
            GridS gs;
            int mirror = 0;
            double mu_scale = 0.;
            if (c->opt_mirror) mirror_prefilter(g, mirror, mu_scale);
            if (sym_grid(g, gs)) k_brick_masks<GridS, 1, false><<<grid, TPB, 0, c->stream>>>(gs, c->rho, small, bmask, bmaxv, fs + FS_TIES, g.x0, mu_scale, mirror, nullptr);
            else k_brick_masks<Grid, 1, false><<<grid, TPB, 0, c->stream>>>(g, c->rho, small, bmask, bmaxv, fs + FS_TIES, g.x0, 0., 0, nullptr);
        }
        HIPCHK(hipGetLastError());
        int ties = 0;
        HIPCHK(hipMemcpyAsync(c->host_ints, fs + FS_TIES, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        ties = c->host_ints[0];
        c->window_ties = ties != 0;
        c->window_seeds.clear();
        c->grad_valid = true;     // (records follow in xb_table_finish)
        c->grad_cover = 1;
        c->grad_rule = 1;
        c->blab = nullptr;
        c->n_boxes = 0; c->box_voxels = 0;
        c->table_stage = 1;
        c->slab_sparse = true;
        if (n_local_seeds) *n_local_seeds = 0;
        return XB_OK;
    }
    if (int rc = ensure_grad(c, true, true, true)) return rc;
    if (n_local_seeds) *n_local_seeds = table_windowed(c) ? (int64_t)c->window_seeds.size() : 0;
    return XB_OK;
}
int xb_table_local_seeds(xb_ctx *c, int64_t *out, int64_t capacity) {
    NEED_GRID_RAW("xb_table_local_seeds");   // (no label is read here: a deferred labels := 0 stays deferred)
    if ((int64_t)c->window_seeds.size() > capacity) return fail(XB_E_ARG, "xb_table_local_seeds: capacity too small");
    for (size_t i = 0; i < c->window_seeds.size(); i++) out[i] = c->window_seeds[i];
    return XB_OK;
}
int xb_brick_masks(xb_ctx *c, void **dev_ptr, int64_t *n_bricks, int64_t *own_first, int64_t *own_count) {
    NEED_GRID_RAW("xb_brick_masks");   // (no label is read here: a deferred labels := 0 stays deferred)
    const Grid &g = c->g;
    if (g.nx % 8 || g.ny % 8 || g.nz % 8) return fail(XB_E_STATE, "xb_brick_masks: grid is not made of whole bricks");
    const int64_t nbr = c->N / 512, per_plane = (int64_t)(g.ny / 8) * (g.nz / 8);
    if (dev_ptr) *dev_ptr = (void *)(c->list + nbr);
    if (n_bricks) *n_bricks = nbr;
    if (own_first) *own_first = (g.x0 / 8) * per_plane;
    if (own_count) *own_count = ((g.x1 - g.x0) / 8) * per_plane;
    return XB_OK;
}
int xb_table_ties(xb_ctx *c, int64_t *has_ties) {
    NEED_GRID_RAW("xb_table_ties");   // (no label is read here: a deferred labels := 0 stays deferred)
    if (c->table_stage < 1 || !c->grad_valid) return fail(XB_E_STATE, "xb_table_ties: call xb_table_build first");
    if (has_ties) *has_ties = c->window_ties ? 1 : 0;
    return XB_OK;
}
int xb_table_finish(xb_ctx *c, const int64_t *seeds, int64_t n_seeds, int64_t any_ties) {
    NEED_GRID_RAW("xb_table_finish");   // (no label is read here: a deferred labels := 0 stays deferred)
    if (c->table_stage < 1 || !c->grad_valid) return fail(XB_E_STATE, "xb_table_finish: call xb_table_build first");
    // the records serve both tie rules (and the regions are closed for the refinement's retraces too) only when NO
    // rank's window holds a tie voxel
    c->grad_rule = any_ties ? 1 : 2;
    if (c->slab_sparse) {
        // every rank holds every brick's mask / maximum now: the same seeding + growth as on one GPU (replicated: the brick
        // arrays are tiny), then the 32-byte records for the uncertain bricks of THIS rank's window
        Grid &g = c->g;
        const GridL gl = light(g);
        const int nb0 = g.nx / BRK, nb1 = g.ny / BRK, nb2 = g.nz / BRK, nbr = nb0 * nb1 * nb2;
        int *fs = c->fs;
        int *seed = c->list, *bmask = c->list + nbr, *buf0 = c->list + 2 * nbr, *buf1 = c->list + 3 * nbr, *bmaxv = c->list + 4 * nbr,
            *reclist = c->list + 5 * nbr;
        int *box_max = c->boxbuf + BB_REGMAX, *box_first = c->boxbuf + BB_REGFIRST;
        c->box_max_tab = box_max;
        ScopedTimer t4(c, 4);
        k_seed_bricks<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, bmask, bmaxv, fs, seed, buf0, box_max);
        k_seed_finish<<<1, 1, 0, c->stream>>>(fs);
        const int launches = 2 * ((std::max(std::max(nb0, nb1), nb2) + BG - 1) / BG) + 12;
        const dim3 ggrid((nb2 + BG - 1) / BG, (nb1 + BG - 1) / BG, (nb0 + BG - 1) / BG);
        for (int l = 0; l < launches; l++)
            k_brick_grow_dev<<<ggrid, BG * BG * BG, 0, c->stream>>>(nb0, nb1, nb2, bmask, seed, buf0, buf1, fs, BG, 0);
        k_fill<int><<<64, 256, 0, c->stream>>>(box_first, XB_INT_MAX, XB_REGIONS_MAX);
        k_grow_finish<<<64, TPB, 0, c->stream>>>(nbr, seed, buf0, buf1, fs, c->blab_buf, box_first, bmask, c->brick_rec, 0);
        c->blab = c->blab_buf;
        c->nbk[0] = nb0; c->nbk[1] = nb1; c->nbk[2] = nb2;
        // the bricks of the window (it may wrap round the grid) that lie outside the regions get their records
        const int per_plane = nb1 * nb2, w0 = g.wx0 / BRK, wn = g.wlen / BRK;
        const int run1 = std::min(wn, nb0 - w0);
        k_brick_walk_list<<<(nbr + 16 * TPB - 1) / (16 * TPB), TPB, 0, c->stream>>>(nbr, w0 * per_plane, (w0 + run1) * per_plane, c->blab, reclist,
                                                                                   fs + FS_N_WALK);
        if (wn > run1)
            k_brick_walk_list<<<(nbr + 16 * TPB - 1) / (16 * TPB), TPB, 0, c->stream>>>(nbr, 0, (wn - run1) * per_plane, c->blab, reclist, fs + FS_N_WALK);
        {
            ScopedTimer t7(c, 7);
            const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
            g.main_ties = 1;
            GridS gs;
            if (sym_grid(g, gs))
                k_brick_records<GridS><<<4096, TPB, 0, c->stream>>>(gs, c->rho, c->grad, reclist, fs + FS_N_WALK, nbr, nb1, nb2, c->brick_rec, small);
            else
                k_brick_records<Grid><<<4096, TPB, 0, c->stream>>>(g, c->rho, c->grad, reclist, fs + FS_N_WALK, nbr, nb1, nb2, c->brick_rec, small);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(c->host_ints, fs, FS_COUNT * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->n_boxes = c->host_ints[FS_N_BOXES];
        c->box_voxels = (long long)c->host_ints[FS_N_CERTAIN] * BRK * BRK * BRK;
        if (!c->host_ints[FS_GROW_CONVERGED] || c->n_boxes == 0) c->blab = nullptr;   // no regions: plain tracing of the slab
        (void)gl;
        c->table_stage = 2;
        c->table_prebuilt = true;
        return XB_OK;
    }
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

void *xb_labels_ptr(xb_ctx *c) {
    if (!c) return nullptr;
    settle_labels(c);
    return (void *)c->labels;
}
void *xb_known_ptr(xb_ctx *c) { return c ? (void *)c->known : nullptr; }
void *xb_density_ptr(xb_ctx *c) { return c ? (void *)c->rho : nullptr; }
int64_t xb_plane_elems(xb_ctx *c) { return c ? c->g.nyz : 0; }

int xb_copy_planes(xb_ctx *c, int which, int to_device, void *host, int64_t xa, int64_t xb) {
    NEED_GRID("xb_copy_planes");
    if (xa < 0 || xb > c->g.nx || xa > xb) return fail(XB_E_ARG, "xb_copy_planes: bad plane range");
    const size_t es = which == 0 ? 4 : 1;
    char *dev = which == 0 ? (char *)c->labels : (char *)c->known;
    const size_t off = (size_t)xa * c->g.nyz * es, bytes = (size_t)(xb - xa) * c->g.nyz * es;
    // (a slab's halo planes come from peers that ran the same assignment: the regions' labels stay what they are)
    if (to_device && which == 0) c->zero_outside[0] = -1;
    if (to_device) { c->list_valid = false; c->has_vacuum = c->has_vacuum || c->g.x1 - c->g.x0 == c->g.nx;
                     c->buni_valid = c->buni_valid && c->buni_halo_safe && c->g.x1 - c->g.x0 < c->g.nx;
                     if (c->g.x1 - c->g.x0 == c->g.nx) c->regions_labels = false; }
    if (to_device) HIPCHK(hipMemcpyAsync(dev + off, host, bytes, hipMemcpyHostToDevice, c->stream));
    else HIPCHK(hipMemcpyAsync(host, dev + off, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}

int xb_brick_masks_copy(xb_ctx *c, int to_device, int32_t *host, int64_t first, int64_t count) {
    NEED_GRID_RAW("xb_brick_masks_copy");   // (no label is read here: a deferred labels := 0 stays deferred)
    const int64_t nbr = c->N / 512;
    if (!host || first < 0 || count < 0 || first + count > nbr) return fail(XB_E_ARG, "xb_brick_masks_copy: bad chunk");
    // host holds 2 * count ints: the move masks of the chunk, then the single-maximum voxels (k_brick_masks)
    int *masks = c->list + nbr, *maxvox = c->list + 4 * nbr;
    if (to_device) {
        HIPCHK(hipMemcpyAsync(masks + first, host, count * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(maxvox + first, host + count, count * sizeof(int), hipMemcpyHostToDevice, c->stream));
    } else {
        HIPCHK(hipMemcpyAsync(host, masks + first, count * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(host + count, maxvox + first, count * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}

int xb_set_option(xb_ctx *c, int key, int value) {
    if (!c) return fail(XB_E_ARG, "null ctx");
    if (key == 0) c->opt_trace = value;
    else if (key == 1) { c->opt_boxes = value & 1; c->opt_bricks = (value >> 1) & 1; }
    else if (key == 3) c->opt_dbg = value;
    else if (key == 4 && value >= 1 && value <= 4096) c->opt_ec_groups = value;
    else if (key == 5 && value >= 2 && value <= EC_Q) c->opt_ec_qcap = value;
    else if (key == 6) c->grad_valid = false;  // drop the cached gradient-field table (a refinement rebuilds it)
    else if (key == 2 && (value == 64 || value == 128 || value == 256)) c->opt_trace_tpb = value;
    else if (key == 7) c->opt_fused = value != 0;  // 0: the host-driven orchestration on one GPU too (tests compare the two)
    else if (key == 8 && value >= 64 && value <= (1 << 22)) c->opt_trace_grid = value;
    else if (key == 9 && value >= 1 && value <= 4096) c->opt_trace_chunk = value;
    else if (key == 10) c->opt_trace_xcd = value != 0;
    else if (key == 11) c->opt_morton = value != 0;
    else if (key == 12) c->opt_sparse = value != 0;   // 0: the round-1 route (a 32-byte record for every voxel)
    else if (key == 14) c->opt_lean = value != 0;
    else if (key == 16) c->opt_chase = value != 0;
    else if (key == 18) c->opt_narrow_halo = value != 0;
    else if (key == 19) c->opt_self_exchange = value != 0;
    else if (key == 20) c->opt_mask_diag = value != 0;
    else if (key == 21) c->opt_trace_cache = value != 0;
    else if (key == 22) c->opt_lean_mem = value != 0;   // (before xb_set_grid)
    else if (key == 24) c->opt_async_comm = value != 0;
    else if (key == 17 && value >= 1) c->grow_kill_launches = value;
    else if (key == 15 && (value == 1 || value == 2 || value == 4 || value == 8)) c->opt_trace_group = value;
    else if (key == 13) c->opt_mirror = value != 0;   // 0: pass A runs the exact ongrid plane test for every open face (tests compare)
    else return fail(XB_E_ARG, "xb_set_option: unknown key %d", key);
    return XB_OK;
}
int xb_deferred_stats(xb_ctx *c, int64_t *refine_total) {
    if (!c) return fail(XB_E_ARG, "null ctx");
    if (refine_total) *refine_total = c->stat_deferred;
    return XB_OK;
}
int xb_slow_path_stats(xb_ctx *c, int64_t *assign_total, int64_t *refine_total) {
    if (!c) return fail(XB_E_ARG, "null ctx");
    if (assign_total) *assign_total = c->stat_ovf_assign;
    if (refine_total) *refine_total = c->stat_ovf_refine;
    return XB_OK;
}
// device bytes this context holds for the grid: density, labels, flags, numbering, the table (its window), scratch
int xb_memory_stats(xb_ctx *c, int64_t *bytes_total, int64_t *bytes_table, int64_t *bytes_scratch) {
    if (!c || !c->has_grid) return fail(XB_E_STATE, "xb_memory_stats: no grid");
    const long long N = c->N;
    const long long table = c->grad_cap * (long long)sizeof(GradRec);
    const long long scratch = c->list_cap * 4 + (long long)c->stage_bytes + c->ec_buf_cap * 4 + (c->ec_pend ? 8 * N + 16 : 0);
    const long long fixed = 8 * N /* rho */ + 4 * N /* labels */ + (N + 16) /* known */ + 4 * N /* first */ + N /* st */ +
                            2LL * c->max_cap * 4 + (long long)c->ovf_cap * 4 + c->blab_alloc * 5 + (long long)c->walk_cap * 3 * 80 +
                            (1 << 22) /* boxbuf */;
    if (bytes_total) *bytes_total = fixed + table + scratch;
    if (bytes_table) *bytes_table = table;
    if (bytes_scratch) *bytes_scratch = scratch;
    return XB_OK;
}
int xb_box_stats(xb_ctx *c, int64_t *n_boxes, int64_t *box_voxels) {
    if (!c) return fail(XB_E_ARG, "null ctx");
    if (n_boxes) *n_boxes = c->n_boxes;
    if (box_voxels) *box_voxels = c->box_voxels;
    return XB_OK;
}
#ifdef XB_DEBUG_COUNT
int xb_debug_counts(unsigned long long *out, int reset) {
    hipDeviceSynchronize();
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(xb_dbg), sizeof(unsigned long long) * 16) != hipSuccess) return XB_E_HIP;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(xb_dbg), z, sizeof z) != hipSuccess) return XB_E_HIP;
    }
    return XB_OK;
}
#endif
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
    if (!c || which < 0 || which > 7) return fail(XB_E_ARG, "xb_kernel_time: bad argument");
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

#include "comm.h"
#include "slab_step.h"
