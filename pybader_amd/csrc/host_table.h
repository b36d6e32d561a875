// host_table.h -- host side, part 2: the gradient-field table outside an assignment.  ensure_grad builds what a
// refinement (or a host-driven trace) needs: the records of the bricks near label boundaries (one GPU, any grid of at least
// 16 voxels per axis), the same bricks again under the other tie rule, or -- where no trapping regions are built -- the records
// of every brick of the table window.  All of it is pass B (k_brick_records); round 1's k_grad_field, its seed cubes and the
// 1023-maxima cap that came with them are gone (round 4).

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
