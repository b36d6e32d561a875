// bader_hip.hip -- libbader_hip.so: HIP kernels + C ABI (include/bader_hip.h) for gfx950.  ONE translation unit:
//   kernels    k_common.h k_masks.h k_fused.h k_trace.h k_ongrid.h k_edges.h k_sums.h k_text.h
//   host side  this file (context struct, options, statistics, timing) + host_context.h (life cycle, transfers)
//              + host_assign.h (the table outside an assignment, the assignments) + host_refine.h + host_sums.h + host_slab_table.h (host-driven slab calls) + comm.h (RCCL through the ABI)
//              + slab_step.h (the slab step with its control flow on the device)
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
    int opt_ec_share = 1;       // k_ec_chase: a long queue sheds its surplus into other workgroups' mailboxes (0: every workgroup keeps what it wakes; A/B, option 29)
    int *ec_share = nullptr;    // the sharing block of k_ec_chase (activity count, mailbox tails and slots), allocated on first use
    int opt_ec_qcap = EC_Q;     // LDS queue entries used per buffer (smaller only in tests)
    std::vector<int64_t> esc_starts, esc_offsets, esc_vox;  // xb_escaped_paths -> xb_escaped_paths_fetch
    unsigned long long *ec_pend = nullptr;  // edge_check's counter word per voxel (8 N bytes, allocated on first use)
    int8_t *ec_pflag = nullptr;             // ... and a flag byte per voxel, set on the processed voxels while their boxes are applied (zero otherwise)
    std::vector<int8_t> esc_complete;
    bool has_vacuum = true;    // false only when volumes_init proved there is no -1 label
    bool vac_by_tol = false;   // the -1 labels are exactly the voxels with rho <= vac_tol (xb_vacuum_assign wrote them, nobody since)
    double vac_tol = 0.;
    bool regions_pending = false;  // labels of certain bricks are written by the relabel pass
    bool buni_valid = false;       // per-brick label uniformity (in `st`) matches the resident labels
    bool buni_halo_safe = false;   // ... and marks every brick outside the owned planes that is not of a trapping region as mixed:
                                   // it stays right when the peers' halo planes arrive (slabs, xb_assign_finish)
    bool regions_neargrid = false; // the regions in `blab` are closed under NEARGRID moves (k_brick_masks); false: the ongrid pointer field's
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
    int *bres_last = nullptr;   // the fused neargrid assignment's per-brick trace results (inside `list`), or null
    int *blab = nullptr;        // brick labels of the trapping regions (inside `list`), or null
    int nbk[3] = {0, 0, 0};
    bool grad_valid = false;
    bool brick_max_valid = false;   // brick_rec bit 1 (the brick holds a 26-neighbour maximum) is right for the density on the card
    int grad_cover = 0;        // 0: the table holds a record for every voxel (of the window); 1: only for the bricks flagged in brick_rec
    unsigned char *brick_rec = nullptr;   // per 8^3 brick: its records exist (k_brick_records), nbr bytes inside blab_buf's allocation
    void *xbuf = nullptr;      // the device-driven slab step's exchange blocks 3-5 (slab_step.h): tie flags, counters, maxima tables
    int slab_rank = 0, slab_nranks = 0, slab_stage = 0;
    void *wbuf[2] = {nullptr, nullptr};   // ... blocks 6 / 7: the walkers of a refinement pass and their results, one part per rank
    void *wk_in = nullptr;                // the walkers this rank carries on in a round (+ their count)
    int wbuf_ranks = 0, walk_last = -1, walk_round = 0, wcap = 0;
    int walk_send = 0;          // walkers of a rank's part that travel in the pass's own gather (0: the capacity; xb_slab_walk_send)
    int opt_async_comm = 0;    // collectives return without waiting (they are ordered on the context's stream); the device-driven slab step sets it
    int opt_tile_dilate = 1;   // the dilation of the edge sweep tile by tile (k_edge_dilate_tiles) instead of from the edge list (tests compare)
    int opt_self_exchange = 0; // tests only: xb_comm_exchange_planes accepts this rank as its own peer (one GPU exercises pack / send / recv / unpack)
    int opt_mask_diag = 1;     // pass A: the three-product form of T_grad . grad on orthogonal lattices (tests compare)
    int opt_narrow_halo = 1;   // label halos travel as dtype_calc(-n_maxima) (int8 / int16) instead of int32 (comm.h)
    int grow_kill_launches = 6;   // kill launches scheduled after a chase (raised to the worst case by the first assignment that needs more)
    long long stat_grow_retries = 0;
    bool defer_wait = false;     // xb_assign_refine: the assignment queues its result transfer and returns without waiting ...
    bool pending_assign = false; // ... and its results are still to be read (after the refinement's first wait)
    int64_t pending_n_maxima = 0;
    int opt_lean = 1;          // persistent trace: the lean walker (k_trace.h, ng_walk_lean); 0: ng_walk_wave (tests compare)
    int opt_mirror = 1;        // pass A: mirror prefilter of the ongrid face test (k_masks.h, bm_mirror)
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
    static constexpr int trace_waves = 8192;   // waves of the persistent trace: 2048 workgroups of XB_TRACE_WAVES = 4, eight per compute unit
    unsigned long long *counters64 = nullptr;
    double *dsum = nullptr;
    int *host_ints = nullptr;  // pinned
    char *big_pin[2] = {nullptr, nullptr};   // two pinned chunks for the large transfers (staged_h2d / staged_d2h)
    hipEvent_t big_ev[2] = {nullptr, nullptr};
    char *pin = nullptr;       // pinned staging for the small host arrays a step uploads (pageable copies pin pages on the fly)
    size_t pin_bytes = 0;
    std::vector<int> maxima_sorted;  // global, label order
    int label_wire = 4;        // bytes per label that hold EVERY resident label (1 / 2 / 4): what the narrowed halo may travel in.
                               // Follows whoever wrote the labels last: an assignment (dtype_calc(-n_maxima)), an upload (its dtype),
                               // volume_assign (the largest atom index) -- calls every rank of a slab run makes alike, so the
                               // ranks agree on it (send / receive sizes).  Planes or voxels written from outside widen it only
                               // if one of their labels does not fit; xb_label_wire lets a scheduler agree on the maximum
    std::vector<int> local_max, local_first;
    bool first_clean = false;
    int chg_n = -1;            // >= 0: the upper half of `stage` lists the chg_n voxels the last retrace pass relabelled (all known == -2 voxels)
    int list_n = 0;            // entries of `list` that hold the owned known == -2 voxels ...
    bool list_valid = false;   // ... when this is set (by xb_edge_find)
    unsigned timing = 0;   // bit k: timer k records its events (xb_enable_timing)
    TimedKernel tk[8];
    long long n_alloc = 0;
};

// utils.dtype_calc(-n) as a byte width (utils.py:25-37): the narrowest signed type for labels 0..n-1 and -1
static inline int label_wire_for(long long n) { return 2 * n <= 255 ? 1 : (2 * n <= 65535 ? 2 : 4); }
// the narrowest signed width that holds each of n labels (host array)
static inline int labels_fit_wire(const int32_t *lab, long long n) {
    int32_t lo = 0, hi = 0;
    for (long long i = 0; i < n; i++) { lo = std::min(lo, lab[i]); hi = std::max(hi, lab[i]); }
    return (lo >= -128 && hi <= 127) ? 1 : ((lo >= -32768 && hi <= 32767) ? 2 : 4);
}
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
        if ((c->timing >> which) & 1u) {
            hipEventCreate(&a);
            hipEventCreate(&b);
            hipEventRecord(a, c->stream);
        }
    }
    ~ScopedTimer() {
        if ((c->timing >> which) & 1u) {
            hipEventRecord(b, c->stream);
            c->tk[which].pending.push_back({a, b});
        }
    }
};

extern "C" {

const char *xb_last_error(void) { return g_err.c_str(); }

#include "host_context.h"
#include "host_assign.h"
#include "host_refine.h"
#include "host_sums.h"
#include "host_slab_table.h"

int xb_set_option(xb_ctx *c, int key, int value) {
    if (!c) return fail(XB_E_ARG, "null ctx");
    if (key == 1) { c->opt_boxes = value & 1; c->opt_bricks = (value >> 1) & 1; }
    else if (key == 2) {   // cross-checks: a second implementation of the same step, for the tests that compare the two
        c->opt_mirror = !(value & 1);        // 1: pass A runs the exact ongrid plane test for every open face (no mirror prefilter)
        c->opt_lean = !(value & 2);          // 2: the generic walker ng_walk_wave instead of the lean one
        c->opt_mask_diag = !(value & 4);     // 4: the full T_grad . grad product on orthogonal lattices too
        c->opt_tile_dilate = !(value & 8);   // 8: the dilation of the edge sweep from the edge list instead of tile by tile
        c->opt_narrow_halo = !(value & 16);  // 16: label halos travel as int32
        c->opt_ec_share = !(value & 32);     // 32: every workgroup of the edge_check chase keeps what it wakes
    }
    else if (key == 3) c->opt_dbg = value;
    else if (key == 4 && value >= 1 && value <= 4096) c->opt_ec_groups = value;
    else if (key == 5 && value >= 2 && value <= EC_Q) c->opt_ec_qcap = value;
    else if (key == 6) { c->grad_valid = false; if (value == 2) c->brick_max_valid = false; }  // drop the cached gradient-field table (a refinement rebuilds it)
    else if (key == 17 && value >= 1) c->grow_kill_launches = value;
    else if (key == 19) c->opt_self_exchange = value != 0;
    else if (key == 24) c->opt_async_comm = value != 0;
    else return fail(XB_E_ARG, "xb_set_option: unknown key %d", key);
    return XB_OK;
}
int xb_growth_stats(xb_ctx *c, int64_t *retries, int64_t *kill_launches) {
    if (!c) return fail(XB_E_ARG, "null ctx");
    if (retries) *retries = c->stat_grow_retries;
    if (kill_launches) *kill_launches = c->grow_kill_launches;
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
int xb_brick_labels(xb_ctx *c, int32_t *out, int64_t capacity, int64_t dims[3]) {
    if (!c || !c->has_grid) return fail(XB_E_STATE, "xb_brick_labels: no grid");
    if (!c->blab) return fail(XB_E_STATE, "xb_brick_labels: no trapping regions on this context (no neargrid assignment yet)");
    const int64_t nbr = (int64_t)c->nbk[0] * c->nbk[1] * c->nbk[2];
    if (dims) { dims[0] = c->nbk[0]; dims[1] = c->nbk[1]; dims[2] = c->nbk[2]; }
    if (!out) return XB_OK;
    if (capacity < nbr) return fail(XB_E_ARG, "xb_brick_labels: capacity too small");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(out, c->blab, nbr * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}
#ifdef XB_DEBUG_COUNT
// the lean walkers / the retraces note their step counts per start voxel in a buffer of their own (tools/walk_lengths.py)
int xb_debug_steps(xb_ctx *c, int on, signed char *host_out) {
    static signed char *buf = nullptr;
    if (host_out && buf) {
        if (hipMemcpy(host_out, buf, (size_t)c->N, hipMemcpyDeviceToHost) != hipSuccess) return XB_E_HIP;
    }
    if (on && !buf && hipMalloc(&buf, (size_t)c->N) != hipSuccess) return XB_E_HIP;
    if (on && hipMemset(buf, 0, (size_t)c->N) != hipSuccess) return XB_E_HIP;
    signed char *p = on ? buf : nullptr;
    return hipMemcpyToSymbol(HIP_SYMBOL(xb_dbg_steps), &p, sizeof p) == hipSuccess ? XB_OK : XB_E_HIP;
}
int xb_debug_counts(unsigned long long *out, int reset) {
    hipDeviceSynchronize();
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(xb_dbg), sizeof(unsigned long long) * 65536) != hipSuccess) return XB_E_HIP;
    if (reset) {
        static unsigned long long z[65536];
        if (hipMemcpyToSymbol(HIP_SYMBOL(xb_dbg), z, sizeof z) != hipSuccess) return XB_E_HIP;
    }
    return XB_OK;
}
#endif
int xb_enable_timing(xb_ctx *c, int on) {
    if (!c) return fail(XB_E_ARG, "null ctx");
    // 0: off; 1: every timer; otherwise bit k + 1 switches timer k on (an event pair costs ~10 us of an idle stream between
    // dependent kernels: a benchmark times its step with the dominant kernel's timer alone)
    c->timing = on == 0 ? 0u : (on == 1 ? 0xFFFFu : ((unsigned)on >> 1));
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
