// host_assign.h -- host side, part 3: the assignments.  xb_assign_trace / xb_assign_finish (host-driven: slabs without the
// device-driven step, ongrid, grids that are not whole bricks), assign_neargrid_fused (one GPU, control flow on the
// device, one host wait) and xb_assign.

// The gradient-field table outside an assignment (ensure_grad): what a refinement (or a host-driven trace) needs -- the records of
// the bricks near label boundaries (one GPU, any grid of at least 16 voxels per axis), the same bricks again under the other tie
// rule, or -- where no trapping regions are built -- the records of every brick of the table window.  All of it is pass B
// (k_brick_records).
static int read_counter(xb_ctx *c, int idx, int *out);
static GridL light(const Grid &g);

// layout of the small device int buffer of the region growth (c->boxbuf): maximum / first brick of up to XB_REGIONS_MAX
// regions (k_seed_bricks)
enum { BB_TOTAL = 1 << 20, BB_REGMAX = 1 << 16, BB_REGFIRST = 1 << 17 };

static bool table_windowed(const xb_ctx *c) { return c->g.wlen < c->g.nx; }

// per-brick arrays that outlive an assignment: blab_buf (nbr ints: region label per brick) and brick_rec (nbr bytes)
static int ensure_brick_bytes(xb_ctx *c, int nbr) {
    if (c->blab_alloc < nbr) {
        hipFree(c->blab_buf); c->blab_buf = nullptr; c->blab_alloc = 0; c->brick_rec = nullptr; c->brick_max_valid = false;
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
        const int nb1r = (g.ny + BRK - 1) / BRK, nb2r = (g.nz + BRK - 1) / BRK, nbr = ((g.nx + BRK - 1) / BRK) * nb1r * nb2r;
        const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
        ScopedTimer t(c, 4);
        GridS gs;
        if (sym_grid(g, gs))
            k_brick_records<GridS><<<4096, TPB, 0, c->stream>>>(gs, c->rho, c->grad, nullptr, nullptr, nbr, nb1r, nb2r, c->brick_rec, small);
        else
            k_brick_records<Grid><<<4096, TPB, 0, c->stream>>>(g, c->rho, c->grad, nullptr, nullptr, nbr, nb1r, nb2r, c->brick_rec, small);
        HIPCHK(hipGetLastError());
        c->grad_rule = main_rule ? 1 : 0;
        return XB_OK;
    }
    if (!force && !boxes && !table_windowed(c) && g.x0 == 0 && g.x1 == g.nx && g.nx >= 16 && g.ny >= 16 && g.nz >= 16) {
        // a refinement without a table from an assignment (ongrid, uploaded labels): retraces only run near label
        // boundaries, so only the bricks whose 27-brick surroundings are not of one label get records (k_masks.h);
        // a retrace that walks on through a brick without records is redone by the from-rho kernel
        const int nb0 = (g.nx + BRK - 1) / BRK, nb1 = (g.ny + BRK - 1) / BRK, nb2 = (g.nz + BRK - 1) / BRK, nbr = nb0 * nb1 * nb2;
        if (int rc = ensure_brick_bytes(c, nbr)) return rc;
        const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
        ScopedTimer t(c, 4);
        int *buni = reinterpret_cast<int *>(c->st);
        if (!c->buni_valid) k_label_uniform_list<<<4096, TPB, 0, c->stream>>>(light(g), c->labels, nb1, nb2, nullptr, nbr, nullptr, nullptr, buni, 0, nbr);
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
    // No trapping regions to lean on (a grid below 16 voxels on an axis, a slab that cuts bricks or whose grid is not made of
    // whole bricks, option 1 = 0): the record of EVERY voxel of the table window, by pass B over all its bricks, and the
    // trajectories are traced in full.  (Round 1-3 kept k_grad_field and closed seed cubes around at most 1023 maxima for
    // these cases: a second pipeline to keep exact, retired in round 4.)
    {
        const int nb0 = (g.nx + BRK - 1) / BRK, nb1 = (g.ny + BRK - 1) / BRK, nb2 = (g.nz + BRK - 1) / BRK, nbr = nb0 * nb1 * nb2;
        if (int rc = ensure_brick_bytes(c, nbr)) return rc;
        const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
        ScopedTimer t(c, 4);
        ScopedTimer tk(c, 5);
        // (a window is brick aligned: xb_set_table_window; bit 1 "may hold a maximum" everywhere: nothing is known about them)
        const int wb0 = table_windowed(c) ? g.wx0 / BRK : 0, wnb = table_windowed(c) ? g.wlen / BRK : nb0;
        k_flag_window_bricks<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nb0, nb1 * nb2, wb0, wnb, (unsigned char)3, c->brick_rec);
        GridS gs;
        if (sym_grid(g, gs))
            k_brick_records<GridS><<<4096, TPB, 0, c->stream>>>(gs, c->rho, c->grad, nullptr, nullptr, nbr, nb1, nb2, c->brick_rec, small);
        else
            k_brick_records<Grid><<<4096, TPB, 0, c->stream>>>(g, c->rho, c->grad, nullptr, nullptr, nbr, nb1, nb2, c->brick_rec, small);
        HIPCHK(hipGetLastError());
    }
    c->grad_valid = true;
    c->grad_cover = 1;
    c->grad_rule = main_rule ? 1 : 0;   // (no tie count on this route: a refinement under the other rule rebuilds the records)
    c->window_ties = true;
    c->n_boxes = 0;
    c->box_voxels = 0;
    c->blab = nullptr;
    c->regions_labels = false;
    c->window_seeds.clear();
    c->table_stage = 2;
    (void)boxes;
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

#define XB_MID_K 32   // voxels of the exact path window of the middle tier (k_ng_trace_list / k_refine_trace over the lean kernels' undecided walkers)
// The exact slow kernel over ovf_list[0..n): whole path in scratch, membership by scanning it (methods.py:411 / refinement.py:200).
// Round 5: in TIERS -- tier 1 gives every walker 64 path voxels (interleaved storage, up to 2 M walkers per launch), the ones whose
// path is longer are listed and go on to 2048, then 32768 voxels.  A density with a noisy vacuum hands MILLIONS of walkers over
// (512^3: 5.8 M); at 2048 walkers x 32768 voxels per launch that took 2800 launches, and the list had a hard cap before.
// (round 5, second half) n_dev: the list's length lives on the device and n is only its bound; stage_free: `stage` is nobody's at
// the moment (a fused assignment, or a fused refinement pass behind its wait) -- the scratch is carved from it behind the list the
// caller may have put at its front, instead of two allocations per call.  A list of up to 64 K walkers runs its first two tiers
// BLIND: both launches are sized by the bound, the second reads the first one's retry count on the device, and the host waits
// once -- for the count of what is left for the third tier (none, almost always) and the error flag.  (216 atoms at 512^3: 11
// walkers per assignment and 10 per refinement pass reach this function; it used to cost each of them 3 waits, 2 allocations
// and 2 frees.)
static int run_slow(xb_ctx *c, int n, int refine, int *max_count = nullptr, int *changed = nullptr, int *escaped = nullptr, const int *list_in = nullptr,
                    const int *n_dev = nullptr, bool stage_free = false) {
    if (!max_count) max_count = c->counters + 0;
    if (!changed) changed = c->counters + 2;
    if (!escaped) escaped = c->counters + 3;
    if (n <= 0) return XB_OK;
    const size_t budget = (size_t)128 << 20;   // ints of path scratch per launch (512 MB)
    // (+ a last tier of 2^20, below; debug switch 64: tiers of 3 / 5 / 8 voxels, so that a test reaches every one of them)
    const int tiers[3] = {(c->opt_dbg & 64) ? 3 : 64, (c->opt_dbg & 64) ? 5 : 2048, (c->opt_dbg & 64) ? 8 : 1 << 15};
    const bool blind = (size_t)n * tiers[1] <= budget;
    const size_t have = std::max<size_t>(std::min<size_t>(budget, (size_t)n * tiers[0]), blind ? (size_t)n * tiers[1] : 0) + (size_t)(1 << 15) * 64;
    const size_t n_lists = 2 * (size_t)n + 8;
    DevBuf<int> path_buf, lists_buf;
    int *path = nullptr, *lists = nullptr;
    {
        const size_t front = (list_in && list_in >= (const int *)c->stage && list_in < (const int *)c->stage + c->stage_bytes / sizeof(int))
                                 ? (size_t)(list_in - (const int *)c->stage) + (size_t)n : 0;
        if (stage_free && c->stage && c->stage_bytes / sizeof(int) >= front + have + n_lists) {
            lists = (int *)c->stage + front;
            path = lists + n_lists;
        } else {
            HIPCHK(path_buf.alloc(have));
            HIPCHK(lists_buf.alloc(n_lists));
            path = path_buf.p; lists = lists_buf.p;
        }
    }
    int *cnt = lists + 2 * (size_t)n;     // [0], [1]: lengths of the two retry lists; [2]: err
    HIPCHK(hipMemsetAsync(cnt, 0, 8 * sizeof(int), c->stream));
    const int *cur = list_in ? list_in : c->ovf_list;
    const int *cur_n_dev = n_dev;
    int n_cur = n;
    if (!blind && n_dev) {   // (a long list is sized by its true length)
        HIPCHK(hipMemcpyAsync(c->host_ints, n_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        n_cur = std::min(n, c->host_ints[0]);
        cur_n_dev = nullptr;
    }
    for (int tier = 0; tier < 3 && n_cur > 0; tier++) {
        const int lmax = tiers[tier];
        int *next = lists + (size_t)(tier & 1) * n, *next_cnt = cnt + (tier & 1);
        const int chunk = (blind && tier < 2) ? n_cur : (int)std::max<size_t>(64, std::min<size_t>(have / lmax, (size_t)n_cur) & ~(size_t)63);
        for (int o = 0; o < n_cur; o += chunk) {   // (blind: one chunk -- the bound times the tier's path length fits)
            const int m = std::min(chunk, n_cur - o);
            k_trace_slow<<<(m + 63) / 64, 64, 0, c->stream>>>(c->g, c->rho, c->labels, c->known, c->known, cur + o, m, path, lmax, refine,
                                                             c->first, c->max_list, max_count, c->max_cap, changed, escaped, cnt + 2, nullptr,
                                                             c->has_vacuum ? 1 : 0, m, 1, next, next_cnt, cur_n_dev);
        }
        HIPCHK(hipGetLastError());
        cur = next;
        if (blind && tier == 0) { cur_n_dev = next_cnt; continue; }   // (tier 1 right behind: its list's length stays on the device)
        HIPCHK(hipMemcpyAsync(c->host_ints, next_cnt, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        n_cur = c->host_ints[0];
        cur_n_dev = nullptr;
        if (n_cur == 0) return XB_OK;
        HIPCHK(hipMemsetAsync(cnt + ((tier + 1) & 1), 0, sizeof(int), c->stream));   // the list the next tier fills
    }
    // Round 6, the last tier: what even 32768 path voxels did not hold (nothing on any density seen so far; rounds 1-5 failed the
    // call here, the reference has no such limit) gets 2^20 voxels per walker in a buffer of its own, a wave of walkers at a time.
    // Only this tier can fail -- loudly.
    {
        const int lmax = 1 << 20;
        DevBuf<int> long_buf;
        HIPCHK(long_buf.alloc((size_t)64 * lmax));
        for (int o = 0; o < n_cur; o += 64) {
            const int m = std::min(64, n_cur - o);
            k_trace_slow<<<1, 64, 0, c->stream>>>(c->g, c->rho, c->labels, c->known, c->known, cur + o, m, long_buf.p, lmax, refine,
                                                 c->first, c->max_list, max_count, c->max_cap, changed, escaped, cnt + 2, nullptr,
                                                 c->has_vacuum ? 1 : 0, m, 1, nullptr, nullptr, nullptr);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(c->host_ints, cnt + 2, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));   // (the scratch may go afterwards)
        if (c->host_ints[0]) return fail(XB_E_LIMIT, "trajectory longer than %d voxels", lmax);
    }
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
    c->regions_neargrid = method == XB_METHOD_NEARGRID;
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
        box_max = c->box_max_tab ? c->box_max_tab : c->boxbuf + BB_REGMAX;
        {
            ScopedTimer t(c, 0);
            const int opt = 3;      // (k_ng_trace: one wave per 4x4x4 cube of start voxels, XCD-aware block order)
            const int tpb = 64;     // one wave per block: a finished wave frees its slot at once
            const bool slab_bricks = (g.x0 % 8 == 0) && (g.x1 % 8 == 0);
            if (c->blab && slab_bricks) {
                // trapping regions known per brick: fill them in one sweep, trace only the rest
                const int nbr = c->nbk[0] * c->nbk[1] * c->nbk[2];
                int *walk = c->list + 4 * nbr;  // a free slice of `list` (seed, masks and the two growth buffers come first)
                c->walk = walk;
                // (the persistent kernel pays off from ~10^5 list items on: 0.24 vs 0.29 ms with the plain launch for an eighth of
                // 512^3, 6.9 vs 7.7 ms for half of 1024^3)
                fast_slab = table_windowed(c) && c->slab_sparse &&
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
                    k_ng_trace_g<2, 0><<<std::max(1, c->trace_waves / XB_TRACE_WAVES), XB_WAVE * XB_TRACE_WAVES, 0, c->stream>>>(light(g), c->grad, box_max, c->blab, c->nbk[1], c->nbk[2], walk,
                                                                                  c->fs, c->labels, c->first, c->max_list, c->max_cap, redo,
                                                                                  redo_cap, maxsteps, c->has_vacuum ? 1 : 0, 8, 1);
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
        // Host-driven ongrid (slabs, vacuum, tiny grids, option 7 = 0): the pointer of every voxel by k_og_masks -- its brick
        // outputs go to scratch, nobody reads them -- then pointer jumping until every chain has reached its root.  (The
        // trapping regions of the pointer field are built by assign_ongrid_fused, one GPU without vacuum.)
        c->blab = nullptr;
        c->n_boxes = 0;
        c->box_voxels = 0;
        {
            ScopedTimer t(c, 1);
            const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
            const int nbr = ((g.nx + BRK - 1) / BRK) * ((g.ny + BRK - 1) / BRK) * ((g.nz + BRK - 1) / BRK);
            if (3LL * nbr > c->list_cap) return fail(XB_E_LIMIT, "xb_assign: scratch too small for the brick arrays of the pointer pass");
            dim3 grid((g.nz + GT_Z - 1) / GT_Z, (g.ny + GT_Y - 1) / GT_Y, (g.nx + GT_X - 1) / GT_X);
            int *bm = c->list, *bmv = c->list + nbr, *bpt = c->list + 2 * nbr;
            GridS gs;
            if (sym_grid(g, gs)) k_og_masks<GridS, true><<<grid, TPB, 0, c->stream>>>(gs, c->rho, c->labels, small, c->has_vacuum ? 1 : 0, bm, bmv, bpt);
            else k_og_masks<Grid, true><<<grid, TPB, 0, c->stream>>>(g, c->rho, c->labels, small, c->has_vacuum ? 1 : 0, bm, bmv, bpt);
        }
        HIPCHK(hipGetLastError());
        box_max = c->boxbuf + BB_REGMAX;
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
    c->label_wire = label_wire_for(n_global);
    const Grid &g = c->g;
    const long long own = (long long)(g.x1 - g.x0) * g.nyz;
    if (n_global) {
        if (int rc = upload_pinned(c, c->max_aux, c->maxima_sorted.data(), n_global * sizeof(int))) return rc;
        k_set_rank<<<(unsigned)((n_global + 255) / 256), 256, 0, c->stream>>>(c->first, c->max_aux, (int)n_global);
        HIPCHK(hipGetLastError());
    }
    c->buni_valid = false; c->regions_labels = false;
    if (c->regions_pending && c->blab) {
        // (one brick-label lookup per 8 rows; 16-byte stores when the rows are aligned.  The launch covers 4-plane groups from x0 on)
        if (g.nz % 4 == 0)
            k_relabel_regions_brick<4><<<dim3((g.nz / 4 + 63) / 64, c->nbk[1], (g.x1 - g.x0 + 3) / 4), TPB, 0, c->stream>>>(
                light(g), c->labels, c->first, c->blab, c->nbk[1], c->nbk[2], (c->box_max_tab ? c->box_max_tab : c->boxbuf + BB_REGMAX),
                nullptr, nullptr, c->n_boxes);
        else
            k_relabel_regions_brick<1><<<dim3((g.nz + 63) / 64, c->nbk[1], (g.x1 - g.x0 + 3) / 4), TPB, 0, c->stream>>>(
                light(g), c->labels, c->first, c->blab, c->nbk[1], c->nbk[2], (c->box_max_tab ? c->box_max_tab : c->boxbuf + BB_REGMAX),
                nullptr, nullptr, c->n_boxes);
        if (g.x1 - g.x0 == g.nx) {  // one slab: the per-brick label uniformity edge_find wants comes for free
            const int nbr = c->nbk[0] * c->nbk[1] * c->nbk[2];
            int *buni = reinterpret_cast<int *>(c->st);
            k_buni_after_relabel<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, c->blab, (c->box_max_tab ? c->box_max_tab : c->boxbuf + BB_REGMAX), c->first, buni, nullptr, nullptr, nullptr, nullptr);
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
            k_fill<int><<<(nbr + 4 * TPB - 1) / (4 * TPB), TPB, 0, c->stream>>>(buni, XB_MIXED, nbr);
            k_buni_after_relabel<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, c->blab, (c->box_max_tab ? c->box_max_tab : c->boxbuf + BB_REGMAX), c->first, buni, nullptr, nullptr, nullptr, nullptr);
            if (c->n_walk)
                k_label_uniform_list<<<(c->n_walk + 3) / 4, TPB, 0, c->stream>>>(light(g), c->labels, c->nbk[1], c->nbk[2],
                                                                                c->walk, c->n_walk, nullptr, nullptr, buni);
            c->buni_valid = true;
            c->buni_halo_safe = true;
        }
    } else
        k_relabel<<<nblocks(own), TPB, 0, c->stream>>>(g, c->labels, c->first, nullptr);
    c->regions_labels = c->regions_pending && c->blab && !c->has_vacuum && c->regions_neargrid;   // certain bricks carry their (neargrid) region's label now
    c->regions_pending = false;
    HIPCHK(hipGetLastError());
    if (n_global) {  // leave `first` clean (INT_MAX everywhere) for the next assignment
        k_reset_first<<<(unsigned)((n_global + 255) / 256), 256, 0, c->stream>>>(c->first, c->max_aux, (int)n_global, nullptr, nullptr);
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
    // (round 4: any grid of at least 16 voxels per axis -- the brick lattice is ceil(n / 8), k_brick_masks PART)
    return c->opt_boxes && c->opt_bricks && g.x0 == 0 && g.x1 == g.nx && !table_windowed(c) &&
           g.nx >= 16 && g.ny >= 16 && g.nz >= 16;
}
static int assign_neargrid_tail(xb_ctx *c, int64_t *n_maxima);
static int fused_numbering_launch(xb_ctx *c);

// numbering + relabel on the device (skipped by their gate when the numbering has to be done on the host), the per-brick
// uniformity for the edge sweep, `first` left clean, and the state block + the sorted maxima on their way to the host
static int fused_relabel_launch(xb_ctx *c);
static int fused_numbering_launch(xb_ctx *c) {
    k_number_maxima<<<1, 1024, 0, c->stream>>>(c->fs, c->first, c->max_list, c->max_cap, c->max_aux, c->fs + FS_TOTAL);
    return fused_relabel_launch(c);
}
// More maxima than k_number_maxima sorts (XB_SORT_MAX): the bitmap numbering of k_fused.h (k_rank_*), then the same relabel
// launches, and the sorted list to the host.  `nmax` maxima are noted, every walker has arrived.  Scratch: N / 4 bytes of `stage`
// (nobody's after an assignment's trace) or an allocation of its own.
static int number_maxima_big(xb_ctx *c, int nmax) {
    const long long n_words = (c->N + 31) / 32, n_blocks = (n_words + 1023) / 1024;
    const size_t need = (size_t)(2 * n_words + n_blocks) * sizeof(int);
    DevBuf<int> own;
    int *scratch = (int *)c->stage;
    if (!c->stage || c->stage_bytes < need) {
        HIPCHK(own.alloc((size_t)(2 * n_words + n_blocks)));
        scratch = own.p;
    }
    unsigned *bits = reinterpret_cast<unsigned *>(scratch);
    int *wprefix = scratch + n_words, *bsum = scratch + 2 * n_words;
    HIPCHK(hipMemsetAsync(bits, 0, (size_t)n_words * sizeof(int), c->stream));
    k_rank_mark<<<(nmax + 255) / 256, 256, 0, c->stream>>>(c->first, c->max_list, nmax, bits);
    k_rank_scan<<<(unsigned)n_blocks, 1024, 0, c->stream>>>(bits, (int)n_words, wprefix, bsum);
    k_rank_blocks<<<1, 1024, 0, c->stream>>>(bsum, (int)n_blocks);
    k_rank_assign<<<(nmax + 255) / 256, 256, 0, c->stream>>>(c->first, c->max_list, nmax, bits, wprefix, bsum, c->max_aux, c->fs);
    HIPCHK(hipGetLastError());
    if (int rc = fused_relabel_launch(c)) return rc;
    c->maxima_sorted.resize(nmax);
    HIPCHK(hipMemcpyAsync(c->maxima_sorted.data(), c->max_aux, (size_t)nmax * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));   // (the scratch may go afterwards)
    return XB_OK;
}
static int fused_relabel_launch(xb_ctx *c) {
    Grid &g = c->g;
    int *fs = c->fs;
    const GridL gl = light(g);
    const int nb1 = c->nbk[1], nb2 = c->nbk[2], nbr = c->nbk[0] * nb1 * nb2;
    int *walk = c->walk, *box_max = c->box_max_tab, *bres = c->bres_last;
    int *buni = reinterpret_cast<int *>(c->st);
    if (c->regions_pending) {
        if (g.nz % 4 == 0)
            k_relabel_regions_brick<4><<<dim3((g.nz / 4 + 63) / 64, nb1, (g.nx + 3) / 4), TPB, 0, c->stream>>>(gl, c->labels, c->first, c->blab, nb1, nb2,
                                                                                                        box_max, fs, fs + FS_SORT_OK);
        else
            k_relabel_regions_brick<1><<<dim3((g.nz + 63) / 64, nb1, (g.nx + 3) / 4), TPB, 0, c->stream>>>(gl, c->labels, c->first, c->blab, nb1, nb2,
                                                                                                     box_max, fs, fs + FS_SORT_OK);
        if (bres) k_buni_after_relabel<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, c->blab, box_max, c->first, buni, fs + FS_SORT_OK, walk, fs + FS_N_WALK, bres);
        else {
            k_buni_after_relabel<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, c->blab, box_max, c->first, buni, fs + FS_SORT_OK, nullptr, nullptr, nullptr);
            k_label_uniform_list<<<2048, TPB, 0, c->stream>>>(gl, c->labels, nb1, nb2, walk, 0, fs + FS_N_WALK, fs + FS_SORT_OK, buni);
        }
    } else
        k_relabel<<<nblocks(c->N), TPB, 0, c->stream>>>(g, c->labels, c->first, fs + FS_SORT_OK);
    k_reset_first<<<8, 256, 0, c->stream>>>(c->first, c->max_aux, 0, fs + FS_N_MAX, fs + FS_SORT_OK);
    HIPCHK(hipGetLastError());
    return XB_OK;
}

static int assign_neargrid_fused(xb_ctx *c, int64_t *n_maxima) {
    if (int rc = need_grad(c)) return rc;
    Grid &g = c->g;
    const GridL gl0 = light(g);
    const int nb0 = (g.nx + BRK - 1) / BRK, nb1 = (g.ny + BRK - 1) / BRK, nb2 = (g.nz + BRK - 1) / BRK, nbr = nb0 * nb1 * nb2;
    const bool part = g.nx % BRK || g.ny % BRK || g.nz % BRK;   // the grid cuts its last bricks
    if (int rc = ensure_brick_bytes(c, nbr)) return rc;
    int *fs = c->fs;
    // scratch carved from `list` (free during an assignment): seed labels, brick masks, two label buffers, walk list
    int *seed = c->list, *bmask = c->list + nbr, *buf0 = c->list + 2 * nbr, *buf1 = c->list + 3 * nbr, *walk = c->list + 4 * nbr;
    int *box_max = c->boxbuf + BB_REGMAX, *box_first = c->boxbuf + BB_REGFIRST;
    int *bmaxv = walk;   // (free until the walk list is made)
    int *bpot = c->list + 5 * nbr;   // brick potentials of the region growth (k_grow_parent); buf1 doubles as the parent array
    const bool chase = true;   // provisional labels by one chase along the brick potentials
    int *bres = nullptr;   // per walk-list brick: the one maximum all its voxels ended on (k_ng_trace_g), for the edge sweep's uniformity
    c->box_max_tab = box_max;
    // (debug switch 32: wait after every stage and say so -- finds the kernel that does not come back)
    auto stage_done = [&](const char *what) {
        if (c->opt_dbg & 32) {
            const hipError_t e = (hipStreamSynchronize)(c->stream);
            int h16[16] = {0};
            (void)hipMemcpy(h16, fs, sizeof h16, hipMemcpyDeviceToHost);
            fprintf(stderr, "[assign] %s: %s  seeds %d regions %d certain %d walk %d maxima %d ovf %d err %d\n", what, hipGetErrorString(e), h16[FS_N_SEEDS],
                    h16[FS_N_BOXES], h16[FS_N_CERTAIN], h16[FS_N_WALK], h16[FS_N_MAX], h16[FS_N_OVF], h16[FS_ERR]);
            fflush(stderr);
        }
    };
    HIPCHK(hipMemsetAsync(fs, 0, FS_TOTAL * sizeof(int), c->stream));
    if (!c->first_clean) {  // a previous assignment did not finish: `first` may hold stale minima
        k_fill<int><<<4096, TPB, 0, c->stream>>>(c->first, XB_INT_MAX, c->N);
        HIPCHK(hipGetLastError());
    }
    c->first_clean = false;
    c->regions_pending = false;
    c->buni_valid = false; c->regions_labels = false;
    c->list_valid = false; c->chg_n = -1;
    g.main_ties = 1;   // methods.neargrid's tie test (methods.py:324)
    const GridL gl = light(g);
    (void)gl0;
    {   // brick masks + seeds
        ScopedTimer t4(c, 4);
        {
            ScopedTimer t5(c, 5);
            const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
            dim3 grid((g.nz + GT_Z - 1) / GT_Z, (g.ny + GT_Y - 1) / GT_Y, (g.nx + GT_X - 1) / GT_X);
            GridS gs;
            const bool sym = sym_grid(g, gs);
            {
                // the assignment's tie rule (methods.py:324) is the template argument
                int mirror = 0;
                double mu_scale = 0.;
                if (sym && c->opt_mirror) mirror_prefilter(g, mirror, mu_scale);
                // (an orthogonal lattice has a diagonal T_grad: exact zeros off the diagonal)
                const bool diag = c->opt_mask_diag && g.T[1] == 0. && g.T[2] == 0. && g.T[3] == 0. && g.T[5] == 0. && g.T[6] == 0. && g.T[7] == 0.;
                if (part) {
                    if (sym && diag) k_brick_masks<GridS, 1, true, true><<<grid, TPB, 0, c->stream>>>(gs, c->rho, small, bmask, bmaxv, fs + FS_TIES, 0, mu_scale, mirror, bpot);
                    else if (sym) k_brick_masks<GridS, 1, false, true><<<grid, TPB, 0, c->stream>>>(gs, c->rho, small, bmask, bmaxv, fs + FS_TIES, 0, mu_scale, mirror, bpot);
                    else k_brick_masks<Grid, 1, false, true><<<grid, TPB, 0, c->stream>>>(g, c->rho, small, bmask, bmaxv, fs + FS_TIES, 0, 0., 0, bpot);
                } else
                if (sym && diag) k_brick_masks<GridS, 1, true><<<grid, TPB, 0, c->stream>>>(gs, c->rho, small, bmask, bmaxv, fs + FS_TIES, 0, mu_scale, mirror, bpot);
                else if (sym) k_brick_masks<GridS, 1, false><<<grid, TPB, 0, c->stream>>>(gs, c->rho, small, bmask, bmaxv, fs + FS_TIES, 0, mu_scale, mirror, bpot);
                else k_brick_masks<Grid, 1, false><<<grid, TPB, 0, c->stream>>>(g, c->rho, small, bmask, bmaxv, fs + FS_TIES, 0, 0., 0, bpot);
            }
        }
        stage_done("brick masks");
        c->grad_valid = true;
        c->grad_rule = 1;
        c->grad_cover = 1;
        {
            // seeds: the bricks that hold exactly one maximum (no cubes, no cap on the number of maxima); they are not
            // fixed: the kill iteration certifies them like every other brick
            k_seed_bricks<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, bmask, bmaxv, fs, seed, buf0, box_max, box_first);
            if (chase) {   // provisional labels by one chase along the brick potentials instead of ~6 propagation launches
                k_grow_parent<<<(nbr + TPB - 1) / TPB, TPB, 0, c->stream>>>(nb0, nb1, nb2, bmask, bpot, seed, buf1);
                k_grow_chase<<<(nbr + TPB - 1) / TPB, TPB, 0, c->stream>>>(nbr, buf1, seed, buf0, 4 * (nb0 + nb1 + nb2) + 64, fs);
            } else
                k_seed_finish<<<1, 1, 0, c->stream>>>(fs);
        }
        // the worst-case schedule; after a chase only the kill iteration is left, which dies out within a few bricks of the
        // dividing surfaces: a short schedule first, and a repeat of the whole assignment with the long one (FS_GROW_RETRY)
        // for the rare density whose cascade runs deeper
        const int long_schedule = 2 * ((std::max(std::max(nb0, nb1), nb2) + BG - 1) / BG) + 12;
        const int launches = chase ? std::min(long_schedule, c->grow_kill_launches) : long_schedule;
        const dim3 ggrid((nb2 + BG - 1) / BG, (nb1 + BG - 1) / BG, (nb0 + BG - 1) / BG);
        for (int l = 0; l < launches; l++)   // each returns at once when the growth has finished (phase on the device)
            k_brick_grow_dev<<<ggrid, BG * BG * BG, 0, c->stream>>>(nb0, nb1, nb2, bmask, seed, buf0, buf1, fs, BG, 0);
        k_grow_finish<<<64, TPB, 0, c->stream>>>(nbr, seed, buf0, buf1, fs, c->blab_buf, box_first, bmask, c->brick_rec, 0, chase && launches < long_schedule ? 1 : 0);
        HIPCHK(hipGetLastError());
        stage_done("region growth");
    }
    c->blab = c->blab_buf;
    c->regions_neargrid = true;
    c->walk = walk;
    c->nbk[0] = nb0; c->nbk[1] = nb1; c->nbk[2] = nb2;
    const long long own = c->N;
    {   // region fill / notes, then the walkers of the uncertain bricks
        ScopedTimer t0(c, 0);
        {
            int bits = 0;
            while ((1 << bits) < std::max(std::max(nb0, nb1), nb2)) bits++;
            const unsigned n_codes = 1u << (3 * bits);
            // (with a vacuum tolerance the bricks that lie below it altogether stay off the list: k_brick_walk_list_morton)
            const bool vac = c->has_vacuum && c->vac_by_tol;
            k_brick_walk_list_morton<<<(n_codes + 16 * TPB - 1) / (16 * TPB), TPB, 0, c->stream>>>(nb0, nb1, nb2, n_codes, c->blab, walk,
                                                                                                  fs + FS_N_WALK, fs + FS_GROW_RETRY,
                                                                                                  vac ? bpot : nullptr, c->vac_tol);
        }
        {   // pass B: records for the bricks of the walk list only
            ScopedTimer t7(c, 7);
            const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
            GridS gs;
            if (sym_grid(g, gs))
                k_brick_records<GridS><<<4096, TPB, 0, c->stream>>>(gs, c->rho, c->grad, walk, fs + FS_N_WALK, nbr, nb1, nb2, c->brick_rec, small);
            else
                k_brick_records<Grid><<<4096, TPB, 0, c->stream>>>(g, c->rho, c->grad, walk, fs + FS_N_WALK, nbr, nb1, nb2, c->brick_rec, small);
        }
        stage_done("walk list + records");
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
#define XB_TRACE_ARGS gl, c->grad, box_max, c->blab, nb1, nb2, walk, fs, c->labels, c->first, c->max_list, c->max_cap, c->ovf_list, c->ovf_cap, \
                      maxsteps, c->has_vacuum ? 1 : 0
            // persistent workgroups of XB_TRACE_WAVES waves, one brick per pull (per-XCD cursors over the Morton-ordered walk list)
            const int groups = std::max(1, c->trace_waves / XB_TRACE_WAVES);
            if (lean) {   // the lean walker, the own brick's records in LDS
                // (without vacuum the walkers also leave, per brick, whether all its voxels ended on one maximum: bres)
                if (!c->has_vacuum) bres = c->list + 6 * nbr;
                if (part) {
                    if (lean == 2) k_ng_trace_g<2, 4, false, true><<<groups, XB_WAVE * XB_TRACE_WAVES, 0, c->stream>>>(XB_TRACE_ARGS, 8, 1, bres);
                    else k_ng_trace_g<2, 3, false, true><<<groups, XB_WAVE * XB_TRACE_WAVES, 0, c->stream>>>(XB_TRACE_ARGS, 8, 1, bres);
                } else if (lean == 2) k_ng_trace_g<2, 4><<<groups, XB_WAVE * XB_TRACE_WAVES, 0, c->stream>>>(XB_TRACE_ARGS, 8, 1, bres);
                else k_ng_trace_g<2, 3><<<groups, XB_WAVE * XB_TRACE_WAVES, 0, c->stream>>>(XB_TRACE_ARGS, 8, 1, bres);
            } else   // the generic walker (option 14 = 0: the tests' cross-check; planes or rows beyond 2^24 voxels); it tests every start voxel
                k_ng_trace_g<2, 0><<<groups, XB_WAVE * XB_TRACE_WAVES, 0, c->stream>>>(XB_TRACE_ARGS, 8, 1);
#undef XB_TRACE_ARGS
        }
        HIPCHK(hipGetLastError());
        stage_done("trace");
    }
    c->bres_last = bres;
    if (int rc = fused_numbering_launch(c)) return rc;
    stage_done("numbering + relabel");
    // the ONE host wait of the assignment: state block + the sorted maxima
    HIPCHK(hipMemcpyAsync(c->host_ints, fs, (FS_TOTAL + XB_SORT_MAX) * sizeof(int), hipMemcpyDeviceToHost, c->stream));   // state block + sorted maxima: one transfer
    if (c->defer_wait) {
        // xb_assign_refine (round 5): the refinement's first iteration is queued right behind this assignment and ONE wait serves
        // both -- the card no longer idles while the host comes back from this wait and goes into the next call (61 us of a 3.4 ms
        // step).  The state the refinement reads is set as the usual outcome leaves it (no tie voxel, nothing for the exact slow
        // path, numbering on the device); assign_neargrid_complete checks the transferred block afterwards and says when it was not so.
        g.main_ties = 0;
        c->grad_rule = 2;
        c->regions_pending = false;
        c->buni_valid = !c->has_vacuum;
        c->regions_labels = !c->has_vacuum;
        c->pending_assign = true;
        return XB_OK;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return assign_neargrid_tail(c, n_maxima);
}

// What the host does once the state block of assign_neargrid_fused has arrived (behind its own wait, or -- xb_assign_refine -- behind
// the wait of the refinement iteration that was queued after it and found the assignment not to have ended the usual way).
static int assign_neargrid_tail(xb_ctx *c, int64_t *n_maxima) {
    Grid &g = c->g;
    int *fs = c->fs;
    const int nb1 = c->nbk[1], nb2 = c->nbk[2];
    int *walk = c->walk, *box_max = c->box_max_tab;
    const int *h = c->host_ints;
    g.main_ties = 0;
    const GridL gl = light(g);
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
    if (nmax > c->max_cap) return fail(XB_E_LIMIT, "%d maxima exceed the table capacity %d", nmax, c->max_cap);
    c->stat_ovf_assign += novf;
    if (h[FS_SORT_OK] && novf == 0) {
        c->maxima_sorted.assign(h + FS_TOTAL, h + FS_TOTAL + nmax);
        c->label_wire = label_wire_for(nmax);
        c->regions_pending = false;
        c->buni_valid = !c->has_vacuum;   // k_buni_after_relabel + k_label_uniform_list ran
        c->regions_labels = !c->has_vacuum;
        c->first_clean = true;
        if (n_maxima) *n_maxima = nmax;
        return XB_OK;
    }
    // rare: trajectories for the exact slow kernel and/or more maxima than the device sort takes
    if (novf > 0) {
        g.main_ties = 1;
        // Round 5, a MIDDLE TIER in front of the exact slow kernel: the listed walkers once more on the table, with an exact path
        // window of XB_MID_K = 32 voxels instead of the lean walker's two (k_ng_trace_list: the generic walker) -- the running-maximum
        // test fails wherever a trajectory dips below a density it passed a few steps ago, which a wider window mostly absorbs
        // (216 atoms at 512^3: 19 K walkers listed; a window of eight left 11 for the slow kernel -- 0.35 ms of serial path scans per step --, 32 leave none).  What it cannot decide either goes on to the
        // slow kernel's tiers.  The second list lives in `stage` (free during an assignment).
        const GridL glt = light(g);
        const int maxsteps = 8 * (g.nx + g.ny + g.nz) + 64;
        int *list2 = (int *)c->stage;
        const int cap2 = (int)std::min<size_t>(c->stage_bytes / sizeof(int), 0x7fffffffu);
        auto two_tiers = [&](int n_listed) -> int {
            c->host_ints[3100] = n_listed;
            HIPCHK(hipMemcpyAsync(c->counters + 15, c->host_ints + 3100, sizeof(int), hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipMemsetAsync(c->counters + 1, 0, sizeof(int), c->stream));
            k_ng_trace_list<XB_MID_K><<<512, TPB, 0, c->stream>>>(glt, c->grad, box_max, c->blab, nb1, nb2, c->ovf_list, c->counters + 15, c->labels, c->first,
                                                          c->max_list, fs + FS_N_MAX, c->max_cap, list2, c->counters + 1, cap2, maxsteps, c->rho, c->dist_dev,
                                                          c->has_vacuum ? 1 : 0);
            HIPCHK(hipGetLastError());
            // (what the wider window leaves is a subset of what it was given: the list's length stays on the device, n_listed bounds it)
            if (c->opt_dbg & 4) {
                int m2 = 0;
                if (int rc2 = read_counter(c, 1, &m2)) return rc2;
                fprintf(stderr, "[assign] %d walkers listed, %d left for the exact slow kernel after the %d-voxel window\n", n_listed, m2, XB_MID_K);
            }
            if (n_listed > cap2) return fail(XB_E_LIMIT, "%d walkers for the exact slow path exceed its list (%d)", n_listed, cap2);
            return run_slow(c, n_listed, 0, fs + FS_N_MAX, nullptr, nullptr, list2, c->counters + 1, true);
        };
        int rc = two_tiers(std::min(novf, c->ovf_cap));
        // more walkers than the list holds (a density that is noise almost everywhere): the unlisted ones still carry -2 in the
        // walk-list bricks -- list and run them a list's worth at a time (round 5: this used to fail the call)
        for (int left = novf - c->ovf_cap; !rc && left > 0;) {
            HIPCHK(hipMemsetAsync(c->counters + 1, 0, sizeof(int), c->stream));
            k_list_unfinished<<<4096, TPB, 0, c->stream>>>(gl, walk, fs + FS_N_WALK, nb1, nb2, c->labels, c->ovf_list, c->counters + 1, c->ovf_cap);
            HIPCHK(hipGetLastError());
            int m = 0;
            if ((rc = read_counter(c, 1, &m))) break;
            if (m == 0) break;
            rc = two_tiers(std::min(m, c->ovf_cap));
            left = m - c->ovf_cap;
        }
        g.main_ties = 0;
        if (rc) return rc;
        // every walker has arrived: the numbering the device declined (k_number_maxima: walkers outstanding) can run there after all
        // -- relabel, the bricks' uniformity from the trace's notes and the reset of `first` with it (round 5: the host used to
        // fetch the table, sort it, send the ranks back and relabel with the generic kernels; five waits more, and the edge sweep
        // of the refinement that follows had to find the uniform bricks by reading them)
        HIPCHK(hipMemsetAsync(fs + FS_N_OVF, 0, sizeof(int), c->stream));
        if (int rc2 = fused_numbering_launch(c)) return rc2;
        HIPCHK(hipMemcpyAsync(c->host_ints, fs, (FS_TOTAL + XB_SORT_MAX) * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        nmax = h[FS_N_MAX];
        if (nmax > c->max_cap) return fail(XB_E_LIMIT, "%d maxima exceed the table capacity %d", nmax, c->max_cap);
        if (h[FS_SORT_OK]) {
            c->maxima_sorted.assign(h + FS_TOTAL, h + FS_TOTAL + nmax);
            c->label_wire = label_wire_for(nmax);
            c->regions_pending = false;
            c->buni_valid = !c->has_vacuum;
            c->regions_labels = !c->has_vacuum;
            c->first_clean = true;
            if (n_maxima) *n_maxima = nmax;
            return XB_OK;
        }
    }
    // more maxima than the LDS sort takes (and every walker in): the bitmap numbering, on the device as well (round 6; the host
    // used to fetch the table, sort it and send the ranks back: 15 of the 66 ms of a noisy vacuum's step)
    if (int rc = number_maxima_big(c, nmax)) return rc;
    c->label_wire = label_wire_for(nmax);
    c->regions_pending = false;
    c->buni_valid = !c->has_vacuum;
    c->regions_labels = !c->has_vacuum;
    c->first_clean = true;
    if (n_maxima) *n_maxima = nmax;
    return XB_OK;
}


// The deferred half of assign_neargrid_fused (xb_assign_refine): the transferred state block is on the host now.  Returns true when
// the assignment ended the usual way -- what the queued refinement iteration took for granted -- and finishes its bookkeeping.
static bool assign_neargrid_complete(xb_ctx *c, int64_t *n_maxima) {
    c->pending_assign = false;
    const int *h = c->host_ints;
    const int nmax = h[FS_N_MAX];
    if (h[FS_GROW_RETRY] || h[FS_TIES] != 0 || h[FS_N_OVF] != 0 || !h[FS_SORT_OK] || nmax > c->max_cap || nmax > XB_SORT_MAX) return false;
    c->n_boxes = h[FS_N_BOXES];
    c->box_voxels = (long long)h[FS_N_CERTAIN] * BRK * BRK * BRK;
    c->n_walk = h[FS_N_WALK];
    c->maxima_sorted.assign(h + FS_TOTAL, h + FS_TOTAL + nmax);
    c->label_wire = label_wire_for(nmax);
    c->first_clean = true;
    if (n_maxima) *n_maxima = nmax;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// The single-GPU ongrid assignment with the control flow on the device (round 4): k_og_masks (pointers + brick masks, maxima,
// potentials) -> the neargrid path's region growth (seed bricks, chase, kill launches, verdict) -> walk list -> pointer chase
// of the uncertain bricks -> numbering and relabel on the device.  One host wait.  Without vacuum only (a chain that steps
// onto a vacuum voxel ends there, methods.py:166-168: a brick with vacuum voxels is no trapping region of its maximum).
// ---------------------------------------------------------------------------------------------------------------
static int assign_ongrid_fused(xb_ctx *c, int64_t *n_maxima) {
    Grid &g = c->g;
    const int nb0 = (g.nx + BRK - 1) / BRK, nb1 = (g.ny + BRK - 1) / BRK, nb2 = (g.nz + BRK - 1) / BRK, nbr = nb0 * nb1 * nb2;
    const bool part = g.nx % BRK || g.ny % BRK || g.nz % BRK;
    if (int rc = ensure_brick_bytes(c, nbr)) return rc;
    int *fs = c->fs;
    int *seed = c->list, *bmask = c->list + nbr, *buf0 = c->list + 2 * nbr, *buf1 = c->list + 3 * nbr, *walk = c->list + 4 * nbr;
    int *bmaxv = walk, *bpot = c->list + 5 * nbr;
    int *box_max = c->boxbuf + BB_REGMAX, *box_first = c->boxbuf + BB_REGFIRST;
    c->box_max_tab = box_max;
    HIPCHK(hipMemsetAsync(fs, 0, FS_TOTAL * sizeof(int), c->stream));
    if (!c->first_clean) {
        k_fill<int><<<4096, TPB, 0, c->stream>>>(c->first, XB_INT_MAX, c->N);
        HIPCHK(hipGetLastError());
    }
    c->first_clean = false;
    c->regions_pending = false;
    c->buni_valid = false; c->regions_labels = false;
    c->list_valid = false; c->chg_n = -1;
    c->zero_outside[0] = -1;
    const GridL gl = light(g);
    {
        ScopedTimer t(c, 1);
        const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
        dim3 grid((g.nz + GT_Z - 1) / GT_Z, (g.ny + GT_Y - 1) / GT_Y, (g.nx + GT_X - 1) / GT_X);
        GridS gs;
        if (sym_grid(g, gs)) {
            if (part) k_og_masks<GridS, true><<<grid, TPB, 0, c->stream>>>(gs, c->rho, c->labels, small, 0, bmask, bmaxv, bpot);
            else k_og_masks<GridS, false><<<grid, TPB, 0, c->stream>>>(gs, c->rho, c->labels, small, 0, bmask, bmaxv, bpot);
        } else {
            if (part) k_og_masks<Grid, true><<<grid, TPB, 0, c->stream>>>(g, c->rho, c->labels, small, 0, bmask, bmaxv, bpot);
            else k_og_masks<Grid, false><<<grid, TPB, 0, c->stream>>>(g, c->rho, c->labels, small, 0, bmask, bmaxv, bpot);
        }
    }
    {
        ScopedTimer t4(c, 4);
        k_seed_bricks<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, bmask, bmaxv, fs, seed, buf0, box_max, box_first);
        k_grow_parent<<<(nbr + TPB - 1) / TPB, TPB, 0, c->stream>>>(nb0, nb1, nb2, bmask, bpot, seed, buf1);
        k_grow_chase<<<(nbr + TPB - 1) / TPB, TPB, 0, c->stream>>>(nbr, buf1, seed, buf0, 4 * (nb0 + nb1 + nb2) + 64, fs);
        const int long_schedule = 2 * ((std::max(std::max(nb0, nb1), nb2) + BG - 1) / BG) + 12;
        const int launches = std::min(long_schedule, c->grow_kill_launches);
        const dim3 ggrid((nb2 + BG - 1) / BG, (nb1 + BG - 1) / BG, (nb0 + BG - 1) / BG);
        for (int l = 0; l < launches; l++)
            k_brick_grow_dev<<<ggrid, BG * BG * BG, 0, c->stream>>>(nb0, nb1, nb2, bmask, seed, buf0, buf1, fs, BG, 0);
        k_grow_finish<<<64, TPB, 0, c->stream>>>(nbr, seed, buf0, buf1, fs, c->blab_buf, box_first, bmask, c->brick_rec, 0, launches < long_schedule ? 1 : 0);
        HIPCHK(hipGetLastError());
    }
    c->grad_valid = false;          // (brick_rec is rewritten: bit 1 = holds a maximum, no records)
    c->grad_cover = 0;
    c->brick_max_valid = true;
    c->blab = c->blab_buf;
    c->regions_neargrid = false;   // (closed under the pointer moves only: the refinement's retraces must not stop on them)
    c->walk = walk;
    c->nbk[0] = nb0; c->nbk[1] = nb1; c->nbk[2] = nb2;
    {
        ScopedTimer t0(c, 0);
        int bits = 0;
        while ((1 << bits) < std::max(std::max(nb0, nb1), nb2)) bits++;
        const unsigned n_codes = 1u << (3 * bits);
        k_brick_walk_list_morton<<<(n_codes + 16 * TPB - 1) / (16 * TPB), TPB, 0, c->stream>>>(nb0, nb1, nb2, n_codes, c->blab, walk,
                                                                                              fs + FS_N_WALK, fs + FS_GROW_RETRY);
        k_note_regions<<<1, XB_BOXES_MAX, 0, c->stream>>>(gl, nb1, nb2, fs, box_first, box_max, c->first, c->max_list, fs + FS_N_MAX, c->max_cap);
        c->regions_pending = true;
        k_og_walk_dev<<<16384, XB_WAVE, 0, c->stream>>>(gl, box_max, c->blab, nb1, nb2, walk, fs, c->labels, c->first, c->max_list, c->max_cap,
                                                        1 << 22);
        HIPCHK(hipGetLastError());
    }
    k_number_maxima<<<1, 1024, 0, c->stream>>>(fs, c->first, c->max_list, c->max_cap, c->max_aux, fs + FS_TOTAL);
    int *buni = reinterpret_cast<int *>(c->st);
    if (g.nz % 4 == 0)
        k_relabel_regions_brick<4><<<dim3((g.nz / 4 + 63) / 64, nb1, (g.nx + 3) / 4), TPB, 0, c->stream>>>(gl, c->labels, c->first, c->blab, nb1, nb2,
                                                                                                    box_max, fs, fs + FS_SORT_OK);
    else
        k_relabel_regions_brick<1><<<dim3((g.nz + 63) / 64, nb1, (g.nx + 3) / 4), TPB, 0, c->stream>>>(gl, c->labels, c->first, c->blab, nb1, nb2,
                                                                                                 box_max, fs, fs + FS_SORT_OK);
    k_buni_after_relabel<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, c->blab, box_max, c->first, buni, fs + FS_SORT_OK, nullptr, nullptr, nullptr);
    k_label_uniform_list<<<2048, TPB, 0, c->stream>>>(gl, c->labels, nb1, nb2, walk, 0, fs + FS_N_WALK, fs + FS_SORT_OK, buni);
    k_reset_first<<<8, 256, 0, c->stream>>>(c->first, c->max_aux, 0, fs + FS_N_MAX, fs + FS_SORT_OK);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->host_ints, fs, (FS_TOTAL + XB_SORT_MAX) * sizeof(int), hipMemcpyDeviceToHost, c->stream));   // state block + sorted maxima: one transfer
    HIPCHK(hipStreamSynchronize(c->stream));
    const int *h = c->host_ints;
    if (h[FS_GROW_RETRY]) {   // the short kill schedule did not reach the fixpoint: once more, with the worst-case one from now on
        c->grow_kill_launches = 1 << 20;
        c->stat_grow_retries++;
        return assign_ongrid_fused(c, n_maxima);
    }
    if (h[FS_ERR] & 1) return fail(XB_E_STATE, "ongrid pointer chase did not terminate");
    c->n_boxes = h[FS_N_BOXES];
    c->box_voxels = (long long)h[FS_N_CERTAIN] * BRK * BRK * BRK;
    c->n_walk = h[FS_N_WALK];
    const int nmax = h[FS_N_MAX];
    if (nmax > c->max_cap) return fail(XB_E_LIMIT, "%d maxima exceed the table capacity %d", nmax, c->max_cap);
    if (h[FS_SORT_OK]) c->maxima_sorted.assign(h + FS_TOTAL, h + FS_TOTAL + nmax);
    else {   // more maxima than the LDS sort takes: the bitmap numbering + the same relabel launches (round 6)
        c->bres_last = nullptr;
        c->regions_pending = true;
        if (int rc = number_maxima_big(c, nmax)) return rc;
    }
    c->label_wire = label_wire_for(nmax);
    c->regions_pending = false;
    c->buni_valid = true;
    c->first_clean = true;
    if (n_maxima) *n_maxima = nmax;
    return XB_OK;
}

// maxima table -> host, sort by first voxel, rank + relabel (the tail of the host-driven path: slabs, vacuum ongrid, small grids)
static int sort_and_finish(xb_ctx *c, int64_t n, int64_t *n_maxima);
int xb_assign(xb_ctx *c, int method, int64_t *n_maxima) {
    NEED_GRID_RAW("xb_assign");
    if (method == XB_METHOD_NEARGRID && fused_ok(c)) {
        if (c->has_vacuum) { if (int rc = settle_labels(c)) return rc; }
        else c->labels_zero_pending = false;   // every label is overwritten, none is read
        return assign_neargrid_fused(c, n_maxima);
    }
    if (method == XB_METHOD_ONGRID && !c->has_vacuum) c->labels_zero_pending = false;   // the pointer pass writes every label
    if (method == XB_METHOD_ONGRID && !c->has_vacuum && fused_ok(c)) return assign_ongrid_fused(c, n_maxima);
    int64_t n = 0;
    if (int rc = xb_assign_trace(c, method, &n)) return rc;
    return sort_and_finish(c, n, n_maxima);
}
static int sort_and_finish(xb_ctx *c, int64_t n, int64_t *n_maxima) {
    // numbering: rank of the smallest voxel index reaching each maximum (thread_handlers.py:59-65
    // numbers maxima in the order the C-order scan discovers them)
    std::vector<int> order(n);
    for (int i = 0; i < n; i++) order[i] = i;
    if (n <= 4096)
        std::sort(order.begin(), order.end(), [&](int a, int b) { return c->local_first[a] < c->local_first[b]; });
    else {
        // a density whose noise makes millions of maxima (round 5: 1.6 M at 512^3 -- the comparison sort through the index took
        // ~100 ms of a 145 ms assignment): three counting passes of 11 bits over (key, index) pairs; the keys are voxel indices,
        // 31 bits, and distinct (a voxel reaches one maximum)
        std::vector<unsigned long long> a(n), b(n);
        for (int64_t i = 0; i < n; i++) a[i] = ((unsigned long long)(unsigned)c->local_first[i] << 32) | (unsigned)i;
        for (int pass = 0; pass < 3; pass++) {
            const int shift = 32 + 11 * pass;
            std::vector<int64_t> cnt(2049, 0);
            for (int64_t i = 0; i < n; i++) cnt[((a[i] >> shift) & 2047) + 1]++;
            for (int k = 0; k < 2048; k++) cnt[k + 1] += cnt[k];
            for (int64_t i = 0; i < n; i++) b[cnt[(a[i] >> shift) & 2047]++] = a[i];
            a.swap(b);
        }
        for (int64_t i = 0; i < n; i++) order[i] = (int)(a[i] & 0xffffffffu);
    }
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
