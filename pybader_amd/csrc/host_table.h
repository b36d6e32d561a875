// host_table.h -- host side, part 2: the gradient-field table and the trapping regions that are NOT built by the brick
// passes A / B of the fused pipeline.
//
// This is where the ROUND-1 ROUTE lives (quarantined here): ensure_grad's full pass (k_grad_field: a 32-byte record for
// every voxel of the window) and table_regions (closed seed cubes around the maxima, k_box_*, then brick growth from
// them).  It still serves (i) grids that are not made of whole 8^3 bricks, (ii) slabs that cut bricks / the slab path with
// the sparse passes switched off, (iii) the ongrid assignment's regions (table_regions with ranges_from_rho), (iv) option
// 12 = 0 (tests compare the two routes).  Everything else -- every BASELINE configuration -- goes through k_brick_masks /
// k_brick_records (assign_neargrid_fused, slab_step.h); ensure_grad's two sparse branches (records for flagged bricks
// only) belong to that pipeline.

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
    if (!force && !boxes && c->opt_sparse && !table_windowed(c) && g.x0 == 0 && g.x1 == g.nx && g.nx >= 16 && g.ny >= 16 && g.nz >= 16) {
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
