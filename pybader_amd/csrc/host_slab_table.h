// host_slab_table.h -- host side, part 6: the table window of a slab and the host-driven slab table calls (xb_table_build /
// xb_table_finish: the fallback of slab_step.h), raw device pointers and plane copies for the transports.


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
    return c->opt_boxes && c->opt_bricks && table_windowed(c) && g.nx % BRK == 0 && g.ny % BRK == 0 && g.nz % BRK == 0 &&
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
        k_seed_bricks<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, bmask, bmaxv, fs, seed, buf0, box_max, box_first);
        k_seed_finish<<<1, 1, 0, c->stream>>>(fs);
        const int launches = 2 * ((std::max(std::max(nb0, nb1), nb2) + BG - 1) / BG) + 12;
        const dim3 ggrid((nb2 + BG - 1) / BG, (nb1 + BG - 1) / BG, (nb0 + BG - 1) / BG);
        for (int l = 0; l < launches; l++)
            k_brick_grow_dev<<<ggrid, BG * BG * BG, 0, c->stream>>>(nb0, nb1, nb2, bmask, seed, buf0, buf1, fs, BG, 0);
        k_grow_finish<<<64, TPB, 0, c->stream>>>(nbr, seed, buf0, buf1, fs, c->blab_buf, box_first, bmask, c->brick_rec, 0, 0);
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
    // (no trapping regions on this route: the records of every voxel of the window are there, the slab is traced in full)
    (void)seeds; (void)n_seeds;
    c->table_stage = 2;
    c->table_prebuilt = true;
    return XB_OK;
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
    // (planes from outside: any label may arrive -- but the halo's wire width must stay a collectively agreed value, so it is
    // widened only when a label of these planes does not fit it: never for planes of peers that ran the same assignment)
    if (to_device && which == 0) {
        c->zero_outside[0] = -1;
        c->label_wire = std::max(c->label_wire, labels_fit_wire((const int32_t *)host, (xb - xa) * c->g.nyz));
    }
    if (to_device) { c->list_valid = false; c->chg_n = -1; c->has_vacuum = c->has_vacuum || c->g.x1 - c->g.x0 == c->g.nx;
                     c->buni_valid = c->buni_valid && c->buni_halo_safe && c->g.x1 - c->g.x0 < c->g.nx;
                     if (c->g.x1 - c->g.x0 == c->g.nx) c->regions_labels = false; }
    if (to_device) HIPCHK(hipMemcpyAsync(dev + off, host, bytes, hipMemcpyHostToDevice, c->stream));
    else HIPCHK(hipMemcpyAsync(host, dev + off, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return XB_OK;
}

// The narrowest signed width (1 / 2 / 4 bytes) that holds every resident label: what a label halo travels in
// (xb_comm_exchange_planes).  It changes in calls that every rank of a slab run makes with the same arguments (numbering,
// uploads, vacuum_assign, volume_assign); `widen_to` (1, 2, 4; 0: only ask) raises it so that a scheduler can make the ranks
// agree after a per-rank event (max over the ranks) -- send and receive sizes must match.
int xb_label_wire(xb_ctx *c, int widen_to, int *wire_out) {
    if (!c) return fail(XB_E_ARG, "xb_label_wire: null context");
    if (widen_to != 0 && widen_to != 1 && widen_to != 2 && widen_to != 4) return fail(XB_E_ARG, "xb_label_wire: width must be 0, 1, 2 or 4");
    c->label_wire = std::max(c->label_wire, widen_to);
    if (wire_out) *wire_out = c->label_wire;
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
