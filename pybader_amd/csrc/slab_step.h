// slab_step.h -- the multi-GPU slab step with its control flow on the device (host side; included by bader_hip.hip).
//
// The host-driven slab calls (xb_table_build / xb_table_finish / xb_assign_trace / xb_assign_finish / xb_edge_find /
// xb_refine_trace) wait for the card about fifteen times per assignment + refinement: every list length, every counter and
// every small table goes through the host before the next launch is sized.  Here each of those values stays on the device --
// the kernels stride over device-side counts as the one-GPU path does (k_fused.h) -- and the three exchanges between the
// ranks are collectives on device buffers ("blocks"), ordered on the context's stream:
//
//   xb_slab_assign_masks    pass A over the owned planes (k_brick_masks): brick move masks, single maxima, potentials
//       blocks 0-3          every rank's chunk of the three brick arrays + its tie flag     -> all ranks
//   xb_slab_assign_trace    region growth (replicated: the brick arrays are tiny), walk lists, records of the table window,
//                           the persistent trace of the owned bricks, the local maxima table packed into block 4
//       block 4             (maximum, smallest owned voxel reaching it) rows of every rank  -> all ranks
//   xb_slab_assign_finish   merge (min over ranks) + numbering + relabel + per-brick uniformity; ONE host wait
//       label halo planes   xb_comm_exchange_planes
//   xb_slab_refine_pass     edge sweep + retraces + walker export; its counters go to block 5
//       block 5             sum over ranks
//   xb_slab_refine_counts   ONE host wait: local and summed counters (+ the exported walkers when there are any)
//
// It replaces thread_handlers.py:28-58 (bader_calc's block split) and 154-232 (refine) exactly as the host-driven calls do;
// the kernels are the same ones.  Whatever it cannot take (vacuum, slabs that are not whole bricks, more than XB_TAB_ROWS
// maxima on a rank, trajectories for the exact slow kernel) it reports (status 2) and the scheduler falls back to them.

#define XB_TAB_ROWS 1024                    // maxima a rank reports through block 4 (more: status 2)
#define XB_TAB_INTS (2 + 2 * XB_TAB_ROWS)   // [rows or -1, unused, (maximum, first voxel) ...]
#define XB_SLAB_RANKS_MAX 64
#define XB_XCNT 8                           // counters of a refinement pass (int64): edges, changed, escaped, walkers, overflows
// blocks 6 / 7: the walkers of a refinement pass (retraces that left their rank's valid planes, k_edges.h) and the results of
// the ones carried on, one part per rank: [walkers, results, 0, 0][cap walkers][cap (start voxel, label) pairs].
// The exchanges are blind (no count goes through the host), so the parts travel at fixed sizes: the pass itself may export
// `cap` walkers (2 % of the edge voxels of an eighth of 512^3 is 13 000), the later rounds -- walkers that crossed a whole
// slab -- cap / 8; what does not fit stays parked and is finished by the host-driven path queries.
// (the number of escapes follows the area of a plane: cap = ny nz / 8, at least 32768; a later round an eighth of it)
__host__ __device__ __forceinline__ size_t walk_part(int cap) { return 16 + (size_t)cap * (sizeof(Walker) + 8); }
static inline int walk_cap_for(const Grid &g) { return (int)std::min<long long>(std::max<long long>(32768, (long long)g.nyz / 8), 1 << 20); }

// ---- kernels of the exchanges ------------------------------------------------------------------------------------------
__global__ void k_slab_flag(const int *fs, int *flags, int rank) { flags[rank] = fs[FS_TIES] != 0; }
__global__ void k_slab_any_flag(const int *flags, int nranks, int *fs) {
    int any = 0;
    for (int r = 0; r < nranks; r++) any |= flags[r];
    fs[FS_TIES] = any;
}
// the local maxima table -> this rank's part of block 4 (rows = -1: it cannot be merged on the device)
// (FS_N_OVF counted the trajectories that left the table window and were redone from rho: from here on it says, as on one
// GPU, how many need the exact slow kernel -- the numbering declines then, on every rank: the table says so)
__global__ void k_slab_pack_table(int *tab, const int *__restrict__ max_list, const int *__restrict__ first, int *fs,
                                  const int *slow_count) {
    const int n = fs[FS_N_MAX];
    const bool bad = n > XB_TAB_ROWS || *slow_count > 0 || fs[FS_GROW_RETRY];
    if (threadIdx.x == 0) { tab[0] = bad ? -1 : n; tab[1] = 0; fs[FS_N_REDO] = fs[FS_N_OVF]; fs[FS_N_OVF] = *slow_count; }
    if (bad) return;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int m = max_list[i];
        tab[2 + 2 * i] = m;
        tab[3 + 2 * i] = first[m];
    }
}
// first[m] := the smallest voxel index of ANY rank reaching maximum m (one block per rank's table)
__global__ void k_slab_merge_min(const int *__restrict__ tabs, int *first) {
    const int *tab = tabs + (size_t)blockIdx.x * XB_TAB_INTS;
    const int n = tab[0];
    for (int i = threadIdx.x; i < n; i += blockDim.x) atomicMin(&first[tab[2 + 2 * i]], tab[3 + 2 * i]);
}
// max_list := the distinct maxima of all tables (a row counts when no earlier row, in rank order, names the same maximum);
// a table that could not be packed makes the numbering decline (FS_N_MAX beyond XB_SORT_MAX)
__global__ __launch_bounds__(1024) void k_slab_merge_list(const int *__restrict__ tabs, int nranks, int *max_list, int max_cap, int *fs) {
    __shared__ int s_bad, s_n;
    if (threadIdx.x == 0) { s_bad = 0; s_n = 0; }
    __syncthreads();
    for (int r = threadIdx.x; r < nranks; r += blockDim.x)
        if (tabs[(size_t)r * XB_TAB_INTS] < 0) s_bad = 1;
    __syncthreads();
    if (s_bad) { if (threadIdx.x == 0) fs[FS_N_MAX] = XB_SORT_MAX + 1; return; }
    for (int r = 0; r < nranks; r++) {
        const int *tab = tabs + (size_t)r * XB_TAB_INTS;
        const int n = tab[0];
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int m = tab[2 + 2 * i];
            bool dup = false;
            for (int q = 0; q < r && !dup; q++) {
                const int *tq = tabs + (size_t)q * XB_TAB_INTS;
                for (int j = 0; j < tq[0]; j++)
                    if (tq[2 + 2 * j] == m) { dup = true; break; }
            }
            if (!dup) {   // (a rank's own rows are distinct: note_maximum lists a maximum once)
                const int k = atomicAdd(&s_n, 1);
                if (k < max_cap) max_list[k] = m;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) fs[FS_N_MAX] = s_n;
}
// counters[18..22): voxels relabelled by applied walker results, stuck results, walkers / results lost to a full part, one
// byte per round: this rank carried walkers on in it (summed over the ranks: how many did -- the scheduler sizes the next
// pass's rounds by it).
// blk: the block the last round gathered (its walkers are still travelling: xcnt[3], the same number on every rank)
__global__ void k_slab_pack_counts(long long *xcnt, const int *counters, const char *blk, int nranks, int rank, int cap) {
    long long open = 0, mine = 0;
    for (int r = 0; blk && r < nranks; r++) {
        const int n = min(reinterpret_cast<const int *>(blk + r * walk_part(cap))[0], cap);
        open += n;
        if (r == rank) mine = n;
    }
    xcnt[0] = counters[5]; xcnt[1] = (long long)counters[2] + counters[18]; xcnt[2] = counters[3]; xcnt[3] = open; xcnt[4] = counters[1];
    xcnt[5] = (long long)counters[19] + counters[20]; xcnt[6] = mine; xcnt[7] = counters[21];
}
// a part that ran full: the surplus is lost (those retraces stay parked and are resolved by the path queries)
__global__ void k_slab_walk_clamp(int *hdr, int *lost, int cap, int rcap) {
    if (hdr[0] > cap) { atomicAdd(lost, hdr[0] - cap); hdr[0] = cap; }
    if (hdr[1] > rcap) { atomicAdd(lost, hdr[1] - rcap); hdr[1] = rcap; }
}
// the results of every rank's part (blockIdx.y), applied by the owner of the start voxel (k_walkers_apply)
__global__ void k_slab_walk_apply(GridL g, const char *blk, int own0, int own1, int *labels, int8_t *known, int *changed, int *stuck, int cap,
                                  int *next_hdr, int *n_in) {
    if (next_hdr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 5) {   // (what the round after this kernel counts into)
        if (threadIdx.x < 4) next_hdr[threadIdx.x] = 0;
        else *n_in = 0;
    }
    const char *part = blk + blockIdx.y * walk_part(cap);
    const int n = min(reinterpret_cast<const int *>(part)[1], cap);
    const int *res = reinterpret_cast<const int *>(part + 16 + (size_t)cap * sizeof(Walker));
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const int v = res[2 * t], nv = res[2 * t + 1];
        const int x = v / g.nyz;
        if (x < own0 || x >= own1) continue;
        if (nv == XB_WALKER_STUCK) { atomicAdd(stuck, 1); continue; }
        if (nv != labels[v]) { labels[v] = nv; known[v] = -2; atomicAdd(changed, 1); }
        else known[v] = -1;
    }
}
// the walkers of every rank's part that arrive on an owned plane -> in[0 .. *n_in)
__global__ void k_slab_walk_collect(GridL g, const char *blk, int own0, int own1, Walker *in, int *n_in, int *lost, int cap) {
    const char *part = blk + blockIdx.y * walk_part(cap);
    const int n = min(reinterpret_cast<const int *>(part)[0], cap);
    const Walker *w = reinterpret_cast<const Walker *>(part + 16);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const int qx = w[t].lq / g.nyz;
        if (qx < own0 || qx >= own1) continue;
        const int k = atomicAdd(n_in, 1);
        if (k < cap) in[k] = w[t];
        else atomicAdd(lost, 1);
    }
}
__global__ void k_slab_walk_clamp_in(int *n_in, int *rounds, int round, int cap) {
    if (*n_in > cap) *n_in = cap;
    if (*n_in > 0 && round < 4) *rounds |= 1 << (8 * round);
}

// ---- host side ---------------------------------------------------------------------------------------------------------
static int slab_need_xbuf(xb_ctx *c) {
    if (c->xbuf) return XB_OK;
    const size_t bytes = 1024 + (size_t)XB_SLAB_RANKS_MAX * XB_TAB_INTS * sizeof(int);
    HIPCHK(hipMalloc(&c->xbuf, bytes));
    HIPCHK(hipMemsetAsync(c->xbuf, 0, bytes, c->stream));
    return XB_OK;
}
static inline int *slab_flags(xb_ctx *c) { return (int *)c->xbuf; }                                // 64 ints
static inline long long *slab_counts(xb_ctx *c) { return (long long *)((char *)c->xbuf + 512); }   // local [0, 8), summed [8, 16)
static inline int *slab_tables(xb_ctx *c) { return (int *)((char *)c->xbuf + 1024); }
static int slab_need_wbuf(xb_ctx *c) {
    const int n = std::max(c->slab_nranks, 1), cap = walk_cap_for(c->g);
    if (c->wbuf[0] && c->wbuf_ranks >= n && c->wcap == cap) return XB_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int k = 0; k < 2; k++) { (void)hipFree(c->wbuf[k]); c->wbuf[k] = nullptr; }
    (void)hipFree(c->wk_in); c->wk_in = nullptr;
    for (int k = 0; k < 2; k++) {
        HIPCHK(hipMalloc(&c->wbuf[k], (size_t)n * walk_part(cap)));
        HIPCHK(hipMemsetAsync(c->wbuf[k], 0, (size_t)n * walk_part(cap), c->stream));
    }
    HIPCHK(hipMalloc(&c->wk_in, (size_t)cap * sizeof(Walker) + 16));
    c->wbuf_ranks = n;
    c->wcap = cap;
    return XB_OK;
}

static bool slab_step_ok(const xb_ctx *c, int nranks) {
    const Grid &g = c->g;
    const long long nbr = c->N / (BRK * BRK * BRK);
    return slab_sparse_ok(c) && !c->has_vacuum && nranks >= 2 && nranks <= XB_SLAB_RANKS_MAX && g.nz % 4 == 0 &&
           c->list_cap >= 7 * nbr && c->halo >= 3;
}

extern "C" {

int xb_host_waits(int64_t *n) {
    if (n) *n = xb_waits;
    return XB_OK;
}

int xb_slab_supported(xb_ctx *c, int nranks, int64_t *ok) {
    NEED_GRID_RAW("xb_slab_supported");
    if (ok) *ok = slab_step_ok(c, nranks) ? 1 : 0;
    return XB_OK;
}

int xb_slab_assign_masks(xb_ctx *c, int rank, int nranks) {
    NEED_GRID_RAW("xb_slab_assign_masks");
    if (rank < 0 || rank >= nranks) return fail(XB_E_ARG, "xb_slab_assign_masks: bad rank");
    if (!slab_step_ok(c, nranks)) return fail(XB_E_STATE, "xb_slab_assign_masks: this slab cannot take the device-driven step (see xb_slab_supported)");
    if (int rc = need_grad(c)) return rc;
    if (int rc = slab_need_xbuf(c)) return rc;
    Grid &g = c->g;
    const int nb1 = g.ny / BRK, nb2 = g.nz / BRK, nbr = (g.nx / BRK) * nb1 * nb2;
    if (int rc = ensure_brick_bytes(c, nbr)) return rc;
    c->slab_rank = rank; c->slab_nranks = nranks;
    int *fs = c->fs, *bmask = c->list + nbr, *bmaxv = c->list + 4 * nbr, *bpot = c->list + 5 * nbr;
    HIPCHK(hipMemsetAsync(fs, 0, FS_TOTAL * sizeof(int), c->stream));
    g.main_ties = 1;   // methods.py:324
    {
        ScopedTimer t4(c, 4);
        ScopedTimer t5(c, 5);
        const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
        dim3 grid((g.nz + GT_Z - 1) / GT_Z, (g.ny + GT_Y - 1) / GT_Y, (g.x1 - g.x0) / GT_X);
        GridS gs;
        int mirror = 0;
        double mu_scale = 0.;
        const bool sym = sym_grid(g, gs);
        if (sym && c->opt_mirror) mirror_prefilter(g, mirror, mu_scale);
        const bool diag = c->opt_mask_diag && g.T[1] == 0. && g.T[2] == 0. && g.T[3] == 0. && g.T[5] == 0. && g.T[6] == 0. && g.T[7] == 0.;
        if (sym && diag) k_brick_masks<GridS, 1, true><<<grid, TPB, 0, c->stream>>>(gs, c->rho, small, bmask, bmaxv, fs + FS_TIES, g.x0, mu_scale, mirror, bpot);
        else if (sym) k_brick_masks<GridS, 1, false><<<grid, TPB, 0, c->stream>>>(gs, c->rho, small, bmask, bmaxv, fs + FS_TIES, g.x0, mu_scale, mirror, bpot);
        else k_brick_masks<Grid, 1, false><<<grid, TPB, 0, c->stream>>>(g, c->rho, small, bmask, bmaxv, fs + FS_TIES, g.x0, 0., 0, bpot);
        k_slab_flag<<<1, 1, 0, c->stream>>>(fs, slab_flags(c), rank);
    }
    HIPCHK(hipGetLastError());
    c->slab_sparse = true;
    c->window_seeds.clear();
    c->grad_valid = true;     // (records follow in xb_slab_assign_trace)
    c->grad_cover = 1;
    c->grad_rule = 1;
    c->blab = nullptr;
    c->n_boxes = 0; c->box_voxels = 0;
    c->table_stage = 1;
    c->slab_stage = 1;
    return XB_OK;
}

// block `which` of the step's exchanges: where it lies on the device, its size, and the part this rank contributes
int xb_slab_block(xb_ctx *c, int which, void **dev_ptr, int64_t *bytes_total, int64_t *own_offset, int64_t *own_bytes) {
    NEED_GRID_RAW("xb_slab_block");
    if (!c->xbuf || c->slab_nranks < 1) return fail(XB_E_STATE, "xb_slab_block: call xb_slab_assign_masks first");
    const Grid &g = c->g;
    const int64_t nbr = c->N / 512, per_plane = (int64_t)(g.ny / 8) * (g.nz / 8);
    void *p = nullptr;
    int64_t total = 0, off = 0, own = 0;
    if (which >= 0 && which <= 2) {
        p = c->list + (which == 0 ? nbr : (which == 1 ? 4 * nbr : 5 * nbr));
        total = nbr * 4; off = (g.x0 / 8) * per_plane * 4; own = ((g.x1 - g.x0) / 8) * per_plane * 4;
    } else if (which == 3) { p = slab_flags(c); total = c->slab_nranks * 4; off = c->slab_rank * 4; own = 4; }
    else if (which == 4) { p = slab_tables(c); total = (int64_t)c->slab_nranks * XB_TAB_INTS * 4; off = (int64_t)c->slab_rank * XB_TAB_INTS * 4; own = XB_TAB_INTS * 4; }
    else if (which == 5) { p = slab_counts(c); total = 2 * XB_XCNT * 8; off = 0; own = XB_XCNT * 8; }
    else if (which == 6 || which == 7) {
        if (int rc = slab_need_wbuf(c)) return rc;
        p = c->wbuf[which - 6]; total = (int64_t)c->slab_nranks * walk_part(c->wcap); off = (int64_t)c->slab_rank * walk_part(c->wcap); own = walk_part(c->wcap);
    }
    else return fail(XB_E_ARG, "xb_slab_block: unknown block %d", which);
    if (dev_ptr) *dev_ptr = p;
    if (bytes_total) *bytes_total = total;
    if (own_offset) *own_offset = off;
    if (own_bytes) *own_bytes = own;
    return XB_OK;
}
// what travels of a part of blocks 6 / 7: [0] bytes of a part, [1] header + walkers of the pass itself (round 0), [2] header +
// walkers of a later round, [3] offset and [4] bytes of the results
// Round 5: what TRAVELS of a part is sized by what the scheduler expects, not by what the part can hold.  The blocks are gathered
// blind (no count goes through the host), so every rank sends a fixed prefix of its part: `walk_send` walkers in the pass's own
// gather (the capacity -- ny nz / 8 walkers of 80 bytes, 2.6 MB per rank at 512^3 -- unless xb_slab_walk_send says less: the
// scheduler knows from the previous pass how many walkers all ranks exported together, which bounds every rank's share), an
// eighth of the capacity at most in the later rounds, and as many results as that many walkers on every rank can leave behind.
// A rank that exports more than it may send loses the surplus exactly as it does when the part is full: those retraces stay
// parked and the host-driven path queries finish them.
static inline int walk_send_of(const xb_ctx *c, int cap) { return c->walk_send > 0 ? std::min(c->walk_send, cap) : cap; }
static inline int walk_later_of(const xb_ctx *c, int cap) { return std::min(cap / 8, walk_send_of(c, cap)); }
static inline int walk_results_of(const xb_ctx *c, int cap) {
    return (int)std::min<long long>(cap, (long long)walk_send_of(c, cap) * std::max(c->slab_nranks, 1));
}
int xb_slab_walk_send(xb_ctx *c, int64_t walkers) {
    NEED_GRID_RAW("xb_slab_walk_send");
    c->walk_send = walkers <= 0 ? 0 : (int)std::min<int64_t>(std::max<int64_t>(walkers, 1024), 1 << 20);
    return XB_OK;
}
int xb_slab_walk_layout(xb_ctx *c, int64_t out[5]) {
    NEED_GRID_RAW("xb_slab_walk_layout");
    if (!out) return fail(XB_E_ARG, "xb_slab_walk_layout: null argument");
    const int cap = walk_cap_for(c->g);
    out[0] = (int64_t)walk_part(cap);
    out[1] = 16 + (int64_t)walk_send_of(c, cap) * sizeof(Walker);
    out[2] = 16 + (int64_t)walk_later_of(c, cap) * sizeof(Walker);
    out[3] = 16 + (int64_t)cap * sizeof(Walker);
    out[4] = (int64_t)walk_results_of(c, cap) * 8;
    return XB_OK;
}
// host-staged transports: bytes [off, off + bytes) of a block to / from the host (waits)
int xb_slab_block_copy(xb_ctx *c, int which, int to_device, void *host, int64_t off, int64_t bytes) {
    void *p = nullptr;
    int64_t total = 0;
    if (int rc = xb_slab_block(c, which, &p, &total, nullptr, nullptr)) return rc;
    if (!host || off < 0 || bytes < 0 || off + bytes > total) return fail(XB_E_ARG, "xb_slab_block_copy: bad range");
    if (!bytes) return XB_OK;
    if (to_device) HIPCHK(hipMemcpyAsync((char *)p + off, host, bytes, hipMemcpyHostToDevice, c->stream));
    else HIPCHK(hipMemcpyAsync(host, (char *)p + off, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}

int xb_slab_assign_trace(xb_ctx *c) {
    NEED_GRID_RAW("xb_slab_assign_trace");
    if (c->slab_stage != 1 || !c->grad_valid) return fail(XB_E_STATE, "xb_slab_assign_trace: call xb_slab_assign_masks first");
    Grid &g = c->g;
    const GridL gl = light(g);
    const int nb0 = g.nx / BRK, nb1 = g.ny / BRK, nb2 = g.nz / BRK, nbr = nb0 * nb1 * nb2;
    int *fs = c->fs;
    int *seed = c->list, *bmask = c->list + nbr, *buf0 = c->list + 2 * nbr, *buf1 = c->list + 3 * nbr, *bmaxv = c->list + 4 * nbr,
        *bpot = c->list + 5 * nbr, *reclist = c->list + 6 * nbr;
    int *walk = bmaxv;   // (the single maxima are consumed by the seeding)
    int *box_max = c->boxbuf + BB_REGMAX, *box_first = c->boxbuf + BB_REGFIRST;
    c->box_max_tab = box_max;
    c->labels_zero_pending = false;   // every owned label is written, none is read (no vacuum; the halo planes are the peers')
    HIPCHK(hipMemsetAsync(c->counters, 0, 16 * sizeof(int), c->stream));
    if (!c->first_clean) {  // a previous assignment did not finish: `first` may hold stale minima
        k_fill<int><<<4096, TPB, 0, c->stream>>>(c->first, XB_INT_MAX, c->N);
        HIPCHK(hipGetLastError());
    }
    c->first_clean = false;
    c->regions_pending = false;
    c->buni_valid = false; c->regions_labels = false;
    c->list_valid = false; c->chg_n = -1;
    const bool chase = true;   // provisional labels by one chase along the brick potentials
    {
        ScopedTimer t4(c, 4);
        k_slab_any_flag<<<1, 1, 0, c->stream>>>(slab_flags(c), c->slab_nranks, fs);
        // every rank holds every brick's mask / maximum / potential now: the same seeding + growth as on one GPU (replicated)
        k_seed_bricks<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, bmask, bmaxv, fs, seed, buf0, box_max, box_first);
        if (chase) {
            k_grow_parent<<<(nbr + TPB - 1) / TPB, TPB, 0, c->stream>>>(nb0, nb1, nb2, bmask, bpot, seed, buf1);
            k_grow_chase<<<(nbr + TPB - 1) / TPB, TPB, 0, c->stream>>>(nbr, buf1, seed, buf0, 4 * (nb0 + nb1 + nb2) + 64, fs);
        } else
            k_seed_finish<<<1, 1, 0, c->stream>>>(fs);
        const int long_schedule = 2 * ((std::max(std::max(nb0, nb1), nb2) + BG - 1) / BG) + 12;
        const int launches = chase ? std::min(long_schedule, c->grow_kill_launches) : long_schedule;
        const dim3 ggrid((nb2 + BG - 1) / BG, (nb1 + BG - 1) / BG, (nb0 + BG - 1) / BG);
        for (int l = 0; l < launches; l++)
            k_brick_grow_dev<<<ggrid, BG * BG * BG, 0, c->stream>>>(nb0, nb1, nb2, bmask, seed, buf0, buf1, fs, BG, 0);
        k_grow_finish<<<64, TPB, 0, c->stream>>>(nbr, seed, buf0, buf1, fs, c->blab_buf, box_first, bmask, c->brick_rec, 0, chase && launches < long_schedule ? 1 : 0);
        HIPCHK(hipGetLastError());
    }
    c->blab = c->blab_buf;
    c->walk = walk;
    c->nbk[0] = nb0; c->nbk[1] = nb1; c->nbk[2] = nb2;
    {
        ScopedTimer t0(c, 0);
        // the bricks of the table window (it may wrap round the grid) outside the regions get their records; the owned ones
        // among them are traced
        const int per_plane = nb1 * nb2, w0 = g.wx0 / BRK, wn = g.wlen / BRK, run1 = std::min(wn, nb0 - w0);
        const unsigned lgrid = (nbr + 16 * TPB - 1) / (16 * TPB);
        k_brick_walk_list<<<lgrid, TPB, 0, c->stream>>>(nbr, w0 * per_plane, (w0 + run1) * per_plane, c->blab, reclist, fs + FS_N_RECL, fs + FS_GROW_RETRY);
        if (wn > run1)
            k_brick_walk_list<<<lgrid, TPB, 0, c->stream>>>(nbr, 0, (wn - run1) * per_plane, c->blab, reclist, fs + FS_N_RECL, fs + FS_GROW_RETRY);
        k_brick_walk_list<<<lgrid, TPB, 0, c->stream>>>(nbr, (g.x0 / BRK) * per_plane, (g.x1 / BRK) * per_plane, c->blab, walk, fs + FS_N_WALK, fs + FS_GROW_RETRY);
        {
            ScopedTimer t7(c, 7);
            const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
            GridS gs;
            if (sym_grid(g, gs))
                k_brick_records<GridS><<<4096, TPB, 0, c->stream>>>(gs, c->rho, c->grad, reclist, fs + FS_N_RECL, nbr, nb1, nb2, c->brick_rec, small);
            else
                k_brick_records<Grid><<<4096, TPB, 0, c->stream>>>(g, c->rho, c->grad, reclist, fs + FS_N_RECL, nbr, nb1, nb2, c->brick_rec, small);
        }
        k_note_certain_bricks<<<(nbr + 255) / 256, 256, 0, c->stream>>>(gl, nb0, nb1, nb2, (g.x0 / BRK) * per_plane, (g.x1 / BRK) * per_plane, c->blab,
                                                                      box_max, c->first, c->max_list, fs + FS_N_MAX, c->max_cap, fs + FS_GROW_RETRY);
        c->regions_pending = true;
        {
            // the persistent trace (per-XCD cursors over the list, its length on the device).  A trajectory that leaves the
            // table window lands on a list (in `stage`) and is redone by the kernel that derives missing records from rho.
            ScopedTimer tw(c, 6);
            const int maxsteps = 8 * (g.nx + g.ny + g.nz) + 64;
            int *redo = (int *)c->stage;
            const int redo_cap = (int)std::min<size_t>(c->stage_bytes / sizeof(int), 0x7fffffffu);
            // (the lean walker in workgroups of eight waves, one brick per pull, its records through LDS -- as on one GPU --
            // when the index products fit 24 bits; 32-bit table offsets up to 2^27 window voxels)
            const bool lean = gl.use24 && c->opt_lean;
            const int groups = std::max(1, c->trace_waves / XB_TRACE_WAVES);
            if (lean && (long long)g.wlen * g.nyz <= (1LL << 27))
                k_ng_trace_g<2, 4, true><<<groups, XB_WAVE * XB_TRACE_WAVES, 0, c->stream>>>(gl, c->grad, box_max, c->blab, nb1, nb2, walk, fs, c->labels, c->first,
                                                                               c->max_list, c->max_cap, redo, redo_cap, maxsteps, 0, 8, 1);
            else if (lean)
                k_ng_trace_g<2, 3, true><<<groups, XB_WAVE * XB_TRACE_WAVES, 0, c->stream>>>(gl, c->grad, box_max, c->blab, nb1, nb2, walk, fs, c->labels, c->first,
                                                                               c->max_list, c->max_cap, redo, redo_cap, maxsteps, 0, 8, 1);
            else
                k_ng_trace_g<2, 0><<<groups, XB_WAVE * XB_TRACE_WAVES, 0, c->stream>>>(gl, c->grad, box_max, c->blab, nb1, nb2, walk, fs, c->labels, c->first,
                                                                         c->max_list, c->max_cap, redo, redo_cap, maxsteps, 0, 8, 1);
            k_ng_trace_list<2><<<512, TPB, 0, c->stream>>>(gl, c->grad, box_max, c->blab, nb1, nb2, redo, fs + FS_N_OVF, c->labels, c->first, c->max_list,
                                                          fs + FS_N_MAX, c->max_cap, c->ovf_list, c->counters + 1, c->ovf_cap, maxsteps, c->rho,
                                                          c->dist_dev, 0);
        }
        k_slab_pack_table<<<1, 256, 0, c->stream>>>(slab_tables(c) + (size_t)c->slab_rank * XB_TAB_INTS, c->max_list, c->first, fs, c->counters + 1);
    }
    HIPCHK(hipGetLastError());
    g.main_ties = 0;
    c->slab_stage = 2;
    return XB_OK;
}

// status: 0 done; 1 the region growth wants its long schedule: repeat the step from xb_slab_assign_masks (every rank sees the
// same verdict: the growth is replicated); 2 this assignment is not for the device-driven step (a rank has more than
// XB_TAB_ROWS maxima or trajectories for the exact slow kernel, more than XB_SORT_MAX maxima in all): use the host-driven calls
int xb_slab_assign_finish(xb_ctx *c, int64_t *n_maxima, int64_t *status) {
    NEED_GRID_RAW("xb_slab_assign_finish");
    if (c->slab_stage != 2) return fail(XB_E_STATE, "xb_slab_assign_finish: call xb_slab_assign_trace first");
    c->slab_stage = 0;
    Grid &g = c->g;
    const GridL gl = light(g);
    const int nb1 = g.ny / BRK, nb2 = g.nz / BRK, nbr = (g.nx / BRK) * nb1 * nb2;
    int *fs = c->fs, *walk = c->list + 4 * nbr, *box_max = c->boxbuf + BB_REGMAX;
    int *buni = reinterpret_cast<int *>(c->st);
    const int *tabs = slab_tables(c);
    k_slab_merge_min<<<c->slab_nranks, 256, 0, c->stream>>>(tabs, c->first);
    k_slab_merge_list<<<1, 1024, 0, c->stream>>>(tabs, c->slab_nranks, c->max_list, c->max_cap, fs);
    k_number_maxima<<<1, 1024, 0, c->stream>>>(fs, c->first, c->max_list, c->max_cap, c->max_aux);
    k_relabel_regions_brick<4><<<dim3((g.nz / 4 + 63) / 64, nb1, (g.x1 - g.x0 + 3) / 4), TPB, 0, c->stream>>>(gl, c->labels, c->first, c->blab, nb1, nb2, box_max,
                                                                                                          fs, fs + FS_SORT_OK);
    // per-brick uniformity for the edge sweep: the regions' bricks are uniform on every rank, the owned walk-list bricks are
    // scanned, every other brick counts as mixed -- right whatever the peers' halo planes bring
    k_fill<int><<<(nbr + 4 * TPB - 1) / (4 * TPB), TPB, 0, c->stream>>>(buni, XB_MIXED, nbr);
    k_buni_after_relabel<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, c->blab, box_max, c->first, buni, fs + FS_SORT_OK, nullptr, nullptr, nullptr);
    k_label_uniform_list<<<2048, TPB, 0, c->stream>>>(gl, c->labels, nb1, nb2, walk, 0, fs + FS_N_WALK, fs + FS_SORT_OK, buni);
    k_reset_first<<<8, 256, 0, c->stream>>>(c->first, c->max_aux, 0, fs + FS_N_MAX, fs + FS_SORT_OK);
    HIPCHK(hipGetLastError());
    // the ONE host wait of the assignment: state block + the sorted maxima
    HIPCHK(hipMemcpyAsync(c->host_ints, fs, FS_COUNT * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->host_ints + FS_COUNT, c->max_aux, XB_SORT_MAX * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const int *h = c->host_ints;
    if (n_maxima) *n_maxima = 0;
    c->regions_pending = false;
    if (h[FS_GROW_RETRY]) {
        c->grow_kill_launches = 1 << 20;
        c->stat_grow_retries++;
        c->grad_valid = false;
        if (status) *status = 1;
        return XB_OK;
    }
    if (!h[FS_SORT_OK]) {
        c->grad_valid = false;
        if (status) *status = 2;
        return XB_OK;
    }
    c->grad_rule = h[FS_TIES] ? 1 : 2;   // the records serve both tie rules only when NO rank's planes hold a tie voxel
    c->window_ties = h[FS_TIES] != 0;
    c->n_boxes = h[FS_N_BOXES];
    c->box_voxels = (long long)h[FS_N_CERTAIN] * BRK * BRK * BRK;
    c->n_walk = h[FS_N_WALK];
    if (c->opt_dbg & 16)
        fprintf(stderr, "[slab %d] bricks traced %d, with records %d, trajectories redone beyond the table window %d, regions %d\n", c->slab_rank,
                h[FS_N_WALK], h[FS_N_RECL], h[FS_N_REDO], h[FS_N_BOXES]);
    const int nmax = h[FS_N_MAX];
    c->maxima_sorted.assign(h + FS_COUNT, h + FS_COUNT + nmax);
    c->label_wire = label_wire_for(nmax);
    c->buni_valid = true;
    c->buni_halo_safe = true;
    c->regions_labels = true;
    c->first_clean = true;
    c->table_stage = 2;
    if (n_maxima) *n_maxima = nmax;
    if (status) *status = 0;
    return XB_OK;
}

// edge sweep + retraces of one refinement iteration on this slab, nothing read back: the pass's counters go to block 5
int xb_slab_refine_pass(xb_ctx *c) {
    NEED_GRID("xb_slab_refine_pass");
    if (int rc = slab_need_xbuf(c)) return rc;
    const Grid &g = c->g;
    if (g.vlen >= g.nx) return fail(XB_E_STATE, "xb_slab_refine_pass: not a slab");
    if (c->slab_nranks < 2) return fail(XB_E_STATE, "xb_slab_refine_pass: call xb_slab_assign_masks first");
    if (int rc = slab_need_wbuf(c)) return rc;
    if (int rc = ensure_grad(c, false, false, false)) return rc;
    bool dilate_owned = false;
    if (int rc = edge_find_launch(c, &dilate_owned)) return rc;
    const GridL gl = light(g);
    if (dilate_owned) {
        ScopedTimer t(c, 2);
        k_edge_dilate_list<<<2048, TPB, 0, c->stream>>>(gl, c->known, c->list, 0, c->counters + 5);
    }
    c->g.main_ties = 0;
    c->list_valid = false; c->chg_n = -1;
    c->buni_valid = false;
    c->walk_n_out = 0; c->walk_n_res = 0; c->walk_out_dev = nullptr;
    c->walk_host.clear(); c->res_host.clear();
    HIPCHK(hipMemsetAsync(c->counters, 0, 5 * sizeof(int), c->stream));           // overflows, changed, escaped ...
    HIPCHK(hipMemsetAsync(c->counters + 6, 0, 18 * sizeof(int), c->stream));      // ... deferred [15], walker statistics [18..22) (the sweep's own counts [6], [22], [23] are spent)
    {
        ScopedTimer t(c, 3);
        const int maxsteps = 8 * (g.nx + g.ny + g.nz) + 64;
        const unsigned char *brec = c->grad_cover == 1 ? c->brick_rec : nullptr;
        const int regions_ok = brec && c->regions_labels && !c->has_vacuum ? 1 : 0;
        const int *slab_regions = (table_windowed(c) && c->blab && c->regions_labels && !c->has_vacuum && (c->grad_rule == 2 || c->slab_sparse) &&
                                   g.nx % 8 == 0 && g.ny % 8 == 0 && g.nz % 8 == 0) ? c->blab : nullptr;
        // the deferred retraces in `stage` (at most one per owned voxel); the walkers go straight to this rank's part of block 6
        int *defer = (int *)c->stage;
        char *part = (char *)c->wbuf[0] + (size_t)c->slab_rank * walk_part(c->wcap);
        HIPCHK(hipMemsetAsync(part, 0, 16, c->stream));
        WalkerIO wio{};
        wio.out = (Walker *)(part + 16); wio.out_count = (int *)part;
        wio.out_cap = walk_send_of(c, c->wcap);
        const unsigned grid = (unsigned)std::min<long long>(nblocks((long long)(g.x1 - g.x0) * g.nyz / 16), 1 << 20);
        k_refine_trace<2, false, false, true><<<grid, TPB, 0, c->stream>>>(gl, c->grad, c->labels, c->known, c->list, 0, c->counters + 5, c->counters + 2,
                                                              c->counters + 3, c->ovf_list, c->counters + 1, c->ovf_cap, maxsteps, c->rho, c->dist_dev,
                                                              brec, defer, c->counters + 15, regions_ok, slab_regions, wio);
        k_refine_trace<2, true><<<512, TPB, 0, c->stream>>>(gl, c->grad, c->labels, c->known, defer, 0, c->counters + 15, c->counters + 2,
                                                            c->counters + 3, c->ovf_list, c->counters + 1, c->ovf_cap, maxsteps, c->rho, c->dist_dev,
                                                            brec, nullptr, nullptr, 0, slab_regions, wio);
    }
    k_slab_walk_clamp<<<1, 1, 0, c->stream>>>((int *)((char *)c->wbuf[0] + (size_t)c->slab_rank * walk_part(c->wcap)), c->counters + 20, walk_send_of(c, c->wcap),
                                              walk_results_of(c, c->wcap));
    HIPCHK(hipGetLastError());
    c->slab_stage = 3;
    c->walk_last = -1;
    c->walk_round = 0;
    return XB_OK;
}

// One round of the walkers' journey, after block 6 + src was gathered: the results in it are applied by the owners of their
// start voxels; unless `last`, the walkers in it that arrive on this rank's planes are carried on (the RESUME retrace) and
// what that leaves -- walkers exported again, results -- goes to this rank's part of the other block.  `last` also packs
// the pass's counters into block 5.  Nothing is read back.
int xb_slab_walkers_round(xb_ctx *c, int src, int last) {
    NEED_GRID("xb_slab_walkers_round");
    if (c->slab_stage != 3 || (src != 0 && src != 1)) return fail(XB_E_STATE, "xb_slab_walkers_round: call xb_slab_refine_pass first");
    const Grid &g = c->g;
    const GridL gl = light(g);
    const char *blk = (const char *)c->wbuf[src];
    const dim3 pgrid(8, c->slab_nranks);
    const int cap = c->wcap;
    char *part = (char *)c->wbuf[1 - src] + (size_t)c->slab_rank * walk_part(cap);
    int *n_in = (int *)((char *)c->wk_in + (size_t)cap * sizeof(Walker));
    k_slab_walk_apply<<<pgrid, 256, 0, c->stream>>>(gl, blk, g.x0, g.x1, c->labels, c->known, c->counters + 18, c->counters + 19, cap,
                                                    last ? nullptr : (int *)part, n_in);
    if (!last) {
        k_slab_walk_collect<<<pgrid, 256, 0, c->stream>>>(gl, blk, g.x0, g.x1, (Walker *)c->wk_in, n_in, c->counters + 20, cap);
        k_slab_walk_clamp_in<<<1, 1, 0, c->stream>>>(n_in, c->counters + 21, c->walk_round++, cap);
        WalkerIO wio{};
        wio.in = (const Walker *)c->wk_in;
        wio.out = (Walker *)(part + 16); wio.out_count = (int *)part; wio.out_cap = walk_later_of(c, cap);
        wio.res = (int *)(part + 16 + (size_t)cap * sizeof(Walker)); wio.res_count = (int *)part + 1;
        wio.own0 = g.x0; wio.own1 = g.x1;
        const unsigned char *brec = c->grad_cover == 1 ? c->brick_rec : nullptr;
        const int *slab_regions = (table_windowed(c) && c->blab && c->regions_labels && !c->has_vacuum && (c->grad_rule == 2 || c->slab_sparse) &&
                                   g.nx % 8 == 0 && g.ny % 8 == 0 && g.nz % 8 == 0) ? c->blab : nullptr;
        const int maxsteps = 8 * (g.nx + g.ny + g.nz) + 64;
        k_refine_trace<2, true, true><<<std::min(cap / TPB, 512), TPB, 0, c->stream>>>(gl, c->grad, c->labels, c->known, nullptr, 0, n_in, c->counters + 2,
                                                                              c->counters + 3, c->ovf_list, c->counters + 1, c->ovf_cap, maxsteps, c->rho,
                                                                              c->dist_dev, brec, nullptr, nullptr, 0, slab_regions, wio);
        k_slab_walk_clamp<<<1, 1, 0, c->stream>>>((int *)part, c->counters + 20, walk_later_of(c, cap), walk_results_of(c, cap));
    } else {
        k_slab_pack_counts<<<1, 1, 0, c->stream>>>(slab_counts(c), c->counters, blk, c->slab_nranks, c->slab_rank, cap);
        c->walk_last = src;
    }
    HIPCHK(hipGetLastError());
    c->list_valid = false; c->chg_n = -1; c->buni_valid = false;
    return XB_OK;
}

// ONE host wait: this rank's counters and their sums over the ranks (block 5 after its all-reduce): edges, changed,
// escaped, exported walkers, retraces for the exact slow kernel.  Exported walkers are fetched here (xb_walkers_fetch);
// retraces for the slow kernel are run here, and local[1], local[2] then hold the counts after them (the caller sums again).
int xb_slab_refine_counts(xb_ctx *c, int64_t *local, int64_t *global) {
    NEED_GRID("xb_slab_refine_counts");
    if (c->slab_stage != 3) return fail(XB_E_STATE, "xb_slab_refine_counts: call xb_slab_refine_pass first");
    c->slab_stage = 0;
    long long *h = reinterpret_cast<long long *>(c->host_ints);
    HIPCHK(hipMemcpyAsync(h, slab_counts(c), 2 * XB_XCNT * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    long long loc[XB_XCNT], glo[XB_XCNT];
    for (int i = 0; i < XB_XCNT; i++) { loc[i] = h[i]; glo[i] = h[XB_XCNT + i]; }
    c->list_n = (int)loc[0];
    if (c->opt_dbg & 16) {
        HIPCHK(hipMemcpyAsync(c->host_ints + 64, c->counters + 15, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        fprintf(stderr, "[slab %d] edges %lld, retraces redone from rho %d, exported %lld\n", c->slab_rank, loc[0], c->host_ints[64], loc[2]);
    }
    c->walk_n_out = 0; c->walk_n_res = 0;
    if (loc[6] > 0 && c->walk_last >= 0) {   // walkers still travelling after the rounds: the scheduler's host loop takes them on
        c->walk_n_out = (int)loc[6];
        c->walk_host.resize((size_t)c->walk_n_out * (sizeof(Walker) / 8));
        const char *part = (const char *)c->wbuf[c->walk_last] + (size_t)c->slab_rank * walk_part(c->wcap);
        if (int rc = download_pinned(c, c->walk_host.data(), part + 16, (size_t)c->walk_n_out * sizeof(Walker))) return rc;
    }
    const int novf = (int)loc[4];
    if (novf > c->ovf_cap) return fail(XB_E_LIMIT, "%d retraces need the slow path (cap %d)", novf, c->ovf_cap);
    c->stat_ovf_refine += novf;
    if (novf > 0) {
        if (int rc = run_slow(c, novf, 1)) return rc;
        HIPCHK(hipMemcpyAsync(c->host_ints, c->counters + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(c->host_ints + 2, c->counters + 18, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        loc[1] = (long long)c->host_ints[0] + c->host_ints[2]; loc[2] = c->host_ints[1];
    }
    for (int i = 0; i < XB_XCNT; i++) { if (local) local[i] = loc[i]; if (global) global[i] = glo[i]; }
    return XB_OK;
}

// ---- the blocks over RCCL: ordered on the context's stream, no host wait -------------------------------------------------
// blocks 0-4: every rank's part to all ranks; first / count (bytes, per rank): where each rank's part lies (slabs may differ
// in size).  One broadcast per rank inside one group.
int xb_comm_allgather_block(xb_ctx *c, int which, const int64_t *first, const int64_t *count) {
    if (int rc = comm_need(c, "xb_comm_allgather_block")) return rc;
    void *p = nullptr;
    int64_t total = 0;
    if (int rc = xb_slab_block(c, which, &p, &total, nullptr, nullptr)) return rc;
    if (which < 0 || which == 5 || which > 7) return fail(XB_E_ARG, "xb_comm_allgather_block: block %d is not gathered", which);
    for (int r = 0; r < c->comm->size; r++)
        if (first[r] < 0 || count[r] < 0 || first[r] + count[r] > total) return fail(XB_E_ARG, "xb_comm_allgather_block: bad part of rank %d", r);
    NCCLCHK(xbcomm::g_api.GroupStart());
    GroupErr ge;
    for (int r = 0; r < c->comm->size; r++)
        if (count[r])
            ge.see(xbcomm::g_api.Broadcast((char *)p + first[r], (char *)p + first[r], (size_t)count[r], xbcomm::ncclInt8, r, c->comm->comm, c->stream), "ncclBroadcast");
    NCCL_GROUP_END(ge, "xb_comm_allgather_block");
    return XB_OK;
}
// block 5: summed[0..8) := sum over ranks of local[0..8)
int xb_comm_allreduce_block(xb_ctx *c) {
    if (int rc = comm_need(c, "xb_comm_allreduce_block")) return rc;
    if (!c->xbuf) return fail(XB_E_STATE, "xb_comm_allreduce_block: no block");
    long long *p = slab_counts(c);
    NCCLCHK(xbcomm::g_api.AllReduce(p, p + XB_XCNT, (size_t)XB_XCNT, xbcomm::ncclInt64, xbcomm::ncclSum, c->comm->comm, c->stream));
    return XB_OK;
}

}  // extern "C"
