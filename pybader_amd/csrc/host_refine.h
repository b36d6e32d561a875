// host_refine.h -- host side, part 4: the refinement.  Edge sweep, retraces (+ escaped-path queries and walkers of the
// slab path), edge_check (dependency counters), the fused one-GPU iteration and xb_refine.

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
        // per-brick label uniformity first: grids of whole bricks (slabs), any grid of at least 16 voxels per axis on one slab
        // (the brick lattice is ceil(n / 8); a brick the grid cuts counts its voxels inside the grid)
        if ((g.nx % 8 == 0 && g.ny % 8 == 0 && g.nz % 8 == 0) || (whole && g.nx >= 16 && g.ny >= 16 && g.nz >= 16)) {
            buni = reinterpret_cast<int *>(c->st);             // N bytes >= 2 nbr ints; edge_check reuses st later
            const int nb0 = (g.nx + 7) / 8, nb1 = (g.ny + 7) / 8, nb2 = (g.nz + 7) / 8;
            const int nbr = nb0 * nb1 * nb2, per_plane = nb1 * nb2;
            if (!c->buni_valid) {
                // a slab only scans the bricks its sweep can look at (the swept planes +- one brick)
                int b_off = 0, count = nbr;
                if (!all && np + 32 < g.nx) {
                    const int p0 = ((xa - 8) % g.nx + g.nx) % g.nx;
                    b_off = (p0 / 8) * per_plane;
                    count = ((np + 8 + 7 + (p0 % 8)) / 8 + 1) * per_plane;
                }
                k_label_uniform_list<<<(unsigned)std::min(4096, (count + 3) / 4), TPB, 0, c->stream>>>(gl, c->labels, nb1, nb2, nullptr, count, nullptr, nullptr, buni, b_off, nbr);
                c->buni_halo_safe = false;
            }
            k_buni3<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nb0, nb1, nb2, buni, buni + nbr);
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
            k_edge_flag_listed<<<std::min(n_own, 2048), TPB, 0, c->stream>>>(gl, c->rho, c->labels, c->known, c->list, c->counters + 5, small, G, brec,
                                                            c->has_vacuum ? 0 : 1, tiles_own, c->counters + 22);
            k_edge_flag_listed<<<std::min(n_halo, 2048), TPB, 0, c->stream>>>(ga, c->rho, c->labels, c->known, halo_list, c->counters + 6, small, G, brec,
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
    c->chg_n = -1;   // `known` is rewritten: the changed-voxel list of an earlier retrace pass no longer describes it
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
    c->list_valid = false; c->chg_n = -1;
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
        if (e == hipSuccess) k_move_voxels<true><<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(d, (int)n, c->labels, c->known, dlab, dkn);
        c->list_valid = false; c->chg_n = -1;
        c->buni_valid = false; c->regions_labels = false;
        c->zero_outside[0] = -1;
        // The halo's wire width is a value every rank must agree on (ncclSend / ncclRecv sizes): a scatter is a per-rank event
        // (only the ranks with parked retraces scatter), so it does NOT widen it -- the labels it writes are labels
        // already present on the grid.  Only a label the resident width cannot hold widens it (raw API use); a scheduler then
        // agrees on the width with xb_label_wire before its next exchange (slab.py does).
        c->label_wire = std::max(c->label_wire, labels_fit_wire(lab, n));
    } else {
        if (e == hipSuccess) k_move_voxels<false><<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(d, (int)n, c->labels, c->known, dlab, dkn);
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
    c->list_valid = false; c->chg_n = -1;  // the retrace rewrites known
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
    c->list_valid = false; c->chg_n = -1; c->buni_valid = false;
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
    int *plist = nullptr;   // the processed voxels, listed for k_ec_apply (in the first seed list: free once the chase is over)
    int plist_cap = 0;
    {
        // counters + classes for the whole list, round 1 over the whole list, then the dependency chains are
        // chased asynchronously by a small grid of workgroups; queue overflows seed another launch.  Scratch: two
        // seed / overflow lists of N ints in the staging buffer, 16 bits per voxel for the counters (only
        // 'changed' refinement needs them).
        // (the seed / overflow lists: in `stage` when it is grid sized, else in a buffer of their own -- at most every listed
        // voxel is queued at once)
        int cap = (int)std::min<long long>(c->N, 0x7fffffffLL);
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
        plist = buf[0]; plist_cap = cap;
        if (!c->ec_pend) HIPCHK(hipMalloc(&c->ec_pend, 8 * (size_t)c->N + 16));
        if (!c->ec_pflag) {
            HIPCHK(hipMalloc(&c->ec_pflag, (size_t)c->N + 16));
            HIPCHK(hipMemsetAsync(c->ec_pflag, 0, (size_t)c->N + 16, c->stream));
        }
        ec_word *pend_w = c->ec_pend;
        // counters: 6 / 24 the seed list and the overflow list of a chase pass (alternating), 25 undecided voxels, 7 new edges,
        // 27 processed voxels, 26 the next pass's list -- zeroed together; the host reads 24 once (did the LDS queues spill?) and
        // the rest with the results
        HIPCHK(hipMemsetAsync(c->counters + 6, 0, 2 * sizeof(int), c->stream));
        HIPCHK(hipMemsetAsync(c->counters + 24, 0, 4 * sizeof(int), c->stream));
        if (cls) k_ec_init_cls<<<nblocks(n), TPB, 0, c->stream>>>(g, c->known, c->list, n, cls, pend_w);
        else k_ec_init<<<(nblocks(n) + 7) & ~7u, TPB, 0, c->stream>>>(g, c->rho, c->labels, c->known, c->list, n, pend_w);
        k_ec_first<<<(unsigned)std::min<long long>(nblocks(n), 4096), TPB, 0, c->stream>>>(g, c->known, pend_w, c->list, n, buf[0],
                                                                                         c->counters + 6, cap);
        HIPCHK(hipGetLastError());
        const int groups = std::max(1, std::min(256, c->opt_ec_groups));
        // (round 5: the workgroups share out a long front through mailboxes -- k_ec_chase)
        int *share = nullptr;
        if (c->opt_ec_share && groups >= 2) {
            if (!c->ec_share) HIPCHK(hipMalloc(&c->ec_share, ec_share_bytes(256)));
            share = c->ec_share;
        }
        int n_seeds = 1;
        for (int pass = 0; n_seeds > 0; pass++) {
            if (pass > 256) return fail(XB_E_LIMIT, "xb_edge_check: queue overflow passes did not drain");
            int *cnt_in = c->counters + ((pass & 1) ? 24 : 6), *cnt_out = c->counters + ((pass & 1) ? 6 : 24);
            if (pass) HIPCHK(hipMemsetAsync(cnt_out, 0, sizeof(int), c->stream));
            if (share) {
                HIPCHK(hipMemsetAsync(share, 0, ec_share_bytes(groups), c->stream));
            }
            k_ec_chase<<<groups, EC_CHASE_THREADS, 0, c->stream>>>(g, c->known, pend_w, buf[pass & 1], cnt_in, buf[1 - (pass & 1)], cnt_out,
                                                                   cap, c->opt_ec_qcap, share);
            HIPCHK(hipGetLastError());
            // (with the first pass's overflow count also k_ec_first's own seed count: k_ec_chase clamps what it reads to the
            // list's capacity, so a list that was too small must fail HERE, not later as "undecided" -- ADVICE r4)
            if (pass == 0) HIPCHK(hipMemcpyAsync(c->host_ints + 1, c->counters + 6, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            if (share) HIPCHK(hipMemcpyAsync(c->host_ints + 2, share + 32, sizeof(int), hipMemcpyDeviceToHost, c->stream));   // (the sharing's error flag)
            if (int rc = read_counter(c, (pass & 1) ? 6 : 24, &n_seeds)) return rc;
            if (n_seeds > cap || (pass == 0 && c->host_ints[1] > cap)) return fail(XB_E_LIMIT, "xb_edge_check: seed list too small");
            if (share && c->host_ints[2]) return fail(XB_E_STATE, "xb_edge_check: a reserved mailbox slot of the chase was never filled");
            if (c->opt_dbg & 4) {
                fprintf(stderr, "edge_check pass %d: %d overflowed (%d groups)\n", pass, n_seeds, groups);
#ifdef XB_EC_PROBE
                    {
                        static unsigned long long hd[256 * 15];
                        HIPCHK(hipMemcpyFromSymbol(hd, HIP_SYMBOL(xb_dbg), sizeof hd, 1024 * sizeof(unsigned long long)));
                        int best = 0;
                        unsigned long long bt = 0;
                        for (int w = 0; w < groups; w++) {
                            unsigned long long a = 0;
                            for (int k = 0; k < 5; k++) a += hd[w * 15 + k];
                            if (a > bt) { bt = a; best = w; }
                        }
                        unsigned long long tot[15] = {0};
                        for (int w = 0; w < groups; w++) for (int k = 0; k < 15; k++) tot[k] += hd[w * 15 + k];
                        const char *nm[5] = {"<=8", "<=32", ">32", "shedding", "long"};
                        for (int k = 0; k < 5; k++)
                            fprintf(stderr, "  rounds %-8s busiest wg %3d (%.3f ms of rounds): %6llu rounds %8.3f us each (%7.1f entries)   all: %8llu rounds %8.3f us each (%7.1f entries)\n", nm[k], best, bt / 1e5,
                                    hd[best * 15 + 5 + k], hd[best * 15 + 5 + k] ? hd[best * 15 + k] / 100.0 / hd[best * 15 + 5 + k] : 0.,
                                    hd[best * 15 + 5 + k] ? (double)hd[best * 15 + 10 + k] / hd[best * 15 + 5 + k] : 0.,
                                    tot[5 + k], tot[5 + k] ? tot[k] / 100.0 / tot[5 + k] : 0., tot[5 + k] ? (double)tot[10 + k] / tot[5 + k] : 0.);
                    }
#endif
                if (share) {
                    int h[8] = {0};
                    HIPCHK(hipMemcpy(h, share + 32, sizeof h, hipMemcpyDeviceToHost));
                    fprintf(stderr, "edge_check sharing: %d entries shed, %d received, %d rounds in the busiest workgroup, %d working rounds in all, error %d\n", h[1], h[3], h[2], h[4], h[0]);
                }
            }
        }
    }
    HIPCHK(hipMemsetAsync(c->counters + 25, 0, sizeof(int), c->stream));
    k_ec_collect<<<nblocks(n), TPB, 0, c->stream>>>(c->known, c->list, n, c->st, c->counters + 25);
    if (near_np < g.nx) k_ec_keep_near<<<nblocks(n), TPB, 0, c->stream>>>(g, c->list, n, c->st, near_xa, near_np);
    HIPCHK(hipMemsetAsync(c->counters64, 0, 2 * sizeof(unsigned long long), c->stream));
    const int new_cap = (int)std::min<long long>(c->list_cap - n, 0x7fffffffLL);   // the rest of `list` behind the compacted edges
    HIPCHK(hipMemsetAsync(c->counters + 7, 0, sizeof(int), c->stream));
    HIPCHK(hipMemsetAsync(c->counters + 26, 0, 2 * sizeof(int), c->stream));
    k_ec_mark<<<nblocks(n), TPB, 0, c->stream>>>(c->list, n, c->st, c->ec_pflag, (int8_t)1, plist, c->counters + 27, plist_cap);
    k_ec_apply<<<(unsigned)std::min<long long>(nblocks(27LL * n), 4096), TPB, 0, c->stream>>>(g, c->rho, c->labels, c->known, plist, c->counters + 27,
                                                  c->counters64 + 1, c->list + n, c->counters + 7, new_cap, c->ec_pflag);
    k_ec_mark<<<nblocks(n), TPB, 0, c->stream>>>(c->list, n, c->st, c->ec_pflag, (int8_t)0, nullptr, nullptr, 0);   // (the flags are zero again)
    k_ec_restore<<<nblocks(n), TPB, 0, c->stream>>>(c->known, c->list, n);
    {   // -1 ring around the new edges (-3): from their list (its length stays on the device) -- a list that may not fit (a
        // grid most of whose voxels changed) is read back, and a full-grid sweep takes over if it did not
        if (new_cap >= std::min<long long>(27LL * n, c->N))
            k_edge_dilate_list<<<(unsigned)std::min<long long>(nblocks(n), 4096), TPB, 0, c->stream>>>(light(g), c->known, c->list + n, 0, c->counters + 7);
        else {
            int n_new = 0;
            if (int rc = read_counter(c, 7, &n_new)) return rc;
            if (n_new > new_cap) k_edge_dilate<<<nblocks(c->N), TPB, 0, c->stream>>>(g, c->known, 0, g.nx, -3);
            else if (n_new) k_edge_dilate_list<<<nblocks(n_new), TPB, 0, c->stream>>>(light(g), c->known, c->list + n, n_new, nullptr);
        }
    }
    // (the sweep also lists the voxels flagged -2 afterwards, the retrace list of the next pass, into `list` itself: its
    // compacted entries and the new-edge list behind them have served)
    const int fin_cap = (int)std::min<long long>(c->list_cap, 0x7fffffffLL);
    k_ec_finish<<<(unsigned)std::min<long long>(nblocks((c->N + 15) / 16), 2048), TPB, 0, c->stream>>>(c->known, c->N, c->counters64,
                                                                                                      count_lo, count_hi, c->list, c->counters + 26, fin_cap);
    HIPCHK(hipGetLastError());
    unsigned long long r[2];
    HIPCHK(hipMemcpyAsync(r, c->counters64, sizeof r, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->host_ints, c->counters + 25, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));   // undecided, list length
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->host_ints[0]) return fail(XB_E_STATE, "xb_edge_check: %d edge voxels left undecided", c->host_ints[0]);
    if (c->host_ints[1] <= fin_cap) { c->list_n = c->host_ints[1]; c->list_valid = true; }   // (else: the next pass compacts the flags itself)
    if (edges) *edges = (int64_t)r[0];
    if (checked) *checked = (int64_t)(r[1] + r[0]);  // refinement.py:479 + 504
    return XB_OK;
}

int xb_edge_check(xb_ctx *c, int64_t *checked, int64_t *edges) {
    NEED_GRID("xb_edge_check");
    const Grid &g = c->g;
    const int chg_n = c->chg_n;   // (read before the invalidations below)
    c->list_valid = false; c->chg_n = -1;
    c->buni_valid = false;
    if (g.x1 - g.x0 != g.nx) return fail(XB_E_STATE, "xb_edge_check: one slab only; slabs use xb_edge_check_local + xb_edge_check_global");
    int n = 0;
    bool listed = false;
    if (chg_n >= 0) {   // the last retrace pass listed its relabelled voxels: exactly the known == -2 ones
        n = chg_n;
        if (n) {
            HIPCHK(hipMemcpyAsync(c->list, (int *)c->stage + c->N, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(c->host_ints + 8, c->fs + FS_N_CHGLIST, sizeof(int), hipMemcpyDeviceToHost, c->stream));   // (checked below)
            listed = true;
        }
    } else if (int rc = compact(c, -2, &n)) return rc;
    if (int rc = edge_check_resolve(c, n, nullptr, 0, g.nx, 0, c->N, checked, edges)) return rc;
    // (the list was launched after the retrace pass's wait, its length taken from the pass's count of relabelled voxels: the
    // kernel's own count came back with the first wait above and must agree)
    if (listed && c->host_ints[8] != n) return fail(XB_E_STATE, "xb_edge_check: %d voxels listed for %d relabelled ones", c->host_ints[8], n);
    return XB_OK;
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
    c->list_valid = false; c->chg_n = -1;
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
    c->list_valid = false; c->chg_n = -1;
    c->buni_valid = false;
    if (checked) *checked = 0;
    if (edges) *edges = 0;
    if (n < 0 || n > c->N) return fail(XB_E_ARG, "xb_edge_check_global: bad list length");
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
#define XB_STEP_REDO 1000   // internal (never crosses the ABI): the assignment queued in front of this refinement did not end the usual way
static int refine_iteration_fused(xb_ctx *c, int64_t *edges, int64_t *changed, bool late_records = false) {
    const Grid &g = c->g;
    int *fs = c->fs;
    c->g.main_ties = 0;
    const GridL gl = light(g);
    static_assert(FS_N_CHGLIST == FS_N_EDGES + 6, "one memset clears the refinement's counters");
    HIPCHK(hipMemsetAsync(fs + FS_N_EDGES, 0, 7 * sizeof(int), c->stream));   // edges, changed, escaped, overflows, deferred, listed tiles, relabelled list
    c->chg_n = -1;
    {
        ScopedTimer t(c, 2);
        const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
        int *buni = nullptr;
        int ntiles_listed = 0;
        if (g.nx >= 16 && g.ny >= 16 && g.nz >= 16) {   // (any such grid: the brick lattice is ceil(n / 8))
            const int nb0 = (g.nx + 7) / 8, nb1 = (g.ny + 7) / 8, nb2 = (g.nz + 7) / 8, nbr = nb0 * nb1 * nb2;
            buni = reinterpret_cast<int *>(c->st);
            if (!c->buni_valid)
                k_label_uniform_list<<<4096, TPB, 0, c->stream>>>(gl, c->labels, nb1, nb2, nullptr, nbr, nullptr, nullptr, buni, 0, nbr);
            k_buni3<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nb0, nb1, nb2, buni, buni + nbr);
            buni += nbr;
        }
        const unsigned char *brec = (c->grad_valid && c->grad_cover == 1) || late_records ? c->brick_rec : nullptr;
        if (buni) {
            // flags preset to "known", then only the tiles that are not of one non-vacuum label with their surroundings
            const int ntiles = ((g.nz + ET_Z - 1) / ET_Z) * ((g.ny + ET_Y - 1) / ET_Y) * ((g.nx + ET_X - 1) / ET_X);
            ntiles_listed = ntiles;
            int *tiles = (int *)c->stage;
            HIPCHK(hipMemsetAsync(c->known, 2, (size_t)c->N, c->stream));
            k_edge_tile_list<<<(ntiles + TPB - 1) / TPB, TPB, 0, c->stream>>>(gl, buni, tiles, fs + FS_N_TILES, 0, (g.nx + ET_X - 1) / ET_X,
                                                                              c->pending_assign ? fs : nullptr);
            k_edge_flag_listed<<<std::min(ntiles, 2048), TPB, 0, c->stream>>>(gl, c->rho, c->labels, c->known, c->list, fs + FS_N_EDGES, small,
                                                           c->grad_valid ? c->grad : nullptr, brec, c->has_vacuum ? 0 : 1, tiles, fs + FS_N_TILES);
        } else {
            dim3 grid((g.nz + ET_Z - 1) / ET_Z, (g.ny + ET_Y - 1) / ET_Y, (g.nx + ET_X - 1) / ET_X);
            k_edge_flag_tiled<<<grid, TPB, 0, c->stream>>>(gl, c->rho, c->labels, c->known, 0, g.nx, c->list, fs + FS_N_EDGES, small, buni,
                                                           c->grad_valid ? c->grad : nullptr, brec, c->has_vacuum ? 0 : 1);
        }
        if (buni && c->opt_tile_dilate && g.nz % ET_Z == 0) k_edge_dilate_tiles<<<std::min(ntiles_listed, 4096), TPB, 0, c->stream>>>(gl, c->known, (const int *)c->stage, fs + FS_N_TILES);
        else k_edge_dilate_list<<<nblocks(c->N / 16), TPB, 0, c->stream>>>(gl, c->known, c->list, 0, fs + FS_N_EDGES);
    }
    c->list_valid = false; c->chg_n = -1;
    c->buni_valid = false;
    if (late_records) {
        // the records of the band bricks (k_masks.h), now that the flags say where the band is
        if (int rc = need_grad(c)) return rc;
        ScopedTimer t(c, 4);
        const int nb0 = (g.nx + 7) / 8, nb1 = (g.ny + 7) / 8, nb2 = (g.nz + 7) / 8, nbr = nb0 * nb1 * nb2;
        const int small = (g.nx < 16 || g.ny < 16 || g.nz < 80);
        unsigned char *band = reinterpret_cast<unsigned char *>(c->blab_buf);   // (the bricks' region labels have served: scratch)
        HIPCHK(hipMemsetAsync(band, 0, (size_t)nbr, c->stream));
        k_flag_band_bricks<<<4096, TPB, 0, c->stream>>>(gl, c->known, (const int *)c->stage, fs + FS_N_TILES, band);
        k_rec_set_bit0<<<(nbr + 255) / 256, 256, 0, c->stream>>>(nbr, c->brick_rec, band);
        GridS gs;
        if (sym_grid(g, gs))
            k_brick_records<GridS><<<4096, TPB, 0, c->stream>>>(gs, c->rho, c->grad, nullptr, nullptr, nbr, nb1, nb2, c->brick_rec, small);
        else
            k_brick_records<Grid><<<4096, TPB, 0, c->stream>>>(g, c->rho, c->grad, nullptr, nullptr, nbr, nb1, nb2, c->brick_rec, small);
        HIPCHK(hipGetLastError());
        c->grad_valid = true;
        c->grad_cover = 1;
        c->grad_rule = 0;
        c->regions_labels = false;
        c->blab = nullptr;
        c->table_stage = 0;
    }
    const int maxsteps = 8 * (g.nx + g.ny + g.nz) + 64;
    const unsigned char *brec = c->grad_cover == 1 ? c->brick_rec : nullptr;
    const int regions_ok = c->grad_cover == 1 && c->regions_labels && !c->has_vacuum ? 1 : 0;
    {
        ScopedTimer t(c, 3);
        int *defer = (int *)c->stage;
        WalkerIO wl{};   // (no walkers on one GPU)
        constexpr int XB_RT_BLOCK = 128;   // (threads per workgroup of the retrace kernel: 64 / 128 / 256 measured 0.512 / 0.495 / 0.514 ms on the whole list)
        k_refine_trace<2, false><<<(unsigned)((c->N / 16 + XB_RT_BLOCK - 1) / XB_RT_BLOCK), XB_RT_BLOCK, 0, c->stream>>>(gl, c->grad, c->labels, c->known, c->list, 0, fs + FS_N_EDGES,
                                                              fs + FS_CHANGED, fs + FS_ESCAPED, c->ovf_list, fs + FS_R_OVF,
                                                              c->ovf_cap, maxsteps, c->rho, c->dist_dev, brec, defer, fs + FS_R_DEFER,
                                                              regions_ok, nullptr, wl);
        if (c->grad_cover == 1 && !regions_ok)   // the few retraces whose walk goes on through a brick without records (count on the
                                                  // device); with regions_ok a retrace stops on such a brick: nothing is deferred
            k_refine_trace<2, true><<<512, TPB, 0, c->stream>>>(gl, c->grad, c->labels, c->known, (int *)c->stage, 0, fs + FS_R_DEFER,
                                                               fs + FS_CHANGED, fs + FS_ESCAPED, c->ovf_list, fs + FS_R_OVF,
                                                               c->ovf_cap, maxsteps, c->rho, c->dist_dev, c->brick_rec, nullptr, nullptr, 0, nullptr, wl);
    }
    HIPCHK(hipGetLastError());
    int *hr = c->host_ints + 3000;   // (its own corner: an assignment's block may still wait to be read at the front, xb_assign_refine)
    HIPCHK(hipMemcpyAsync(hr, fs + FS_N_EDGES, 7 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->pending_assign && !assign_neargrid_complete(c, &c->pending_n_maxima)) return XB_STEP_REDO;   // (nothing below is to be trusted)
    *edges = hr[0];
    *changed = hr[1];
    const int novf = hr[3];
    // The relabelled start voxels (known == -2 now) are listed for the next edge_check in the upper half of `stage` -- launched only
    // when there are any (round 5: the launch came before the wait and returned at once in the usual case of a neargrid
    // assignment, nothing changed: 5 us of every step).  The list is complete when nothing went to the exact slow kernel.
    if (novf == 0 && hr[1] > 0 && c->stage_bytes >= 8 * (size_t)c->N && hr[1] <= (int)std::min<long long>(c->N, 0x7fffffffLL)) {
        k_list_changed<<<1024, TPB, 0, c->stream>>>(c->list, fs + FS_N_EDGES, c->known, (int *)c->stage + c->N, fs + FS_N_CHGLIST,
                                                    (int)std::min<long long>(c->N, 0x7fffffffLL), fs + FS_CHANGED);
        HIPCHK(hipGetLastError());
        c->chg_n = hr[1];   // (every relabelled voxel is an entry of the edge list: the kernel lists exactly that many)
    } else if (novf == 0 && hr[1] == 0)
        c->chg_n = 0;
    c->stat_deferred += hr[4];
    if (hr[2]) return fail(XB_E_STATE, "xb_refine: %d traces left the grid", hr[2]);
    if (novf > c->ovf_cap) return fail(XB_E_LIMIT, "%d retraces need the slow path (cap %d)", novf, c->ovf_cap);
    c->stat_ovf_refine += novf;
    if (novf > 0) {
        // Round 5, the MIDDLE TIER of the assignment (host_assign.h) for the retraces: the listed ones once more on the table with an
        // exact path window of XB_MID_K voxels instead of two -- what the running-maximum test could not decide is mostly a dip below a
        // density passed a few steps ago -- and only what is left goes to the exact slow kernel's tiers (216 atoms at 512^3: 13 K
        // retraces listed per pass).  Not where a retrace may be deferred to the from-rho kernel: its list lives in `stage` as well.
        const int *slow_list = nullptr, *slow_n_dev = nullptr;
        const int n_slow = novf;
        const bool can_defer = c->grad_cover == 1 && !regions_ok;
        if (!can_defer && c->stage_bytes >= sizeof(int) * (size_t)novf) {
            int *list2 = (int *)c->stage;
            HIPCHK(hipMemsetAsync(c->counters + 1, 0, sizeof(int), c->stream));
            k_refine_trace<XB_MID_K, false><<<nblocks(novf), TPB, 0, c->stream>>>(gl, c->grad, c->labels, c->known, c->ovf_list, novf, nullptr, fs + FS_CHANGED,
                                                                          fs + FS_ESCAPED, list2, c->counters + 1, novf, maxsteps,
                                                                          c->rho, c->dist_dev, brec, nullptr, nullptr, regions_ok, nullptr, WalkerIO{});
            HIPCHK(hipGetLastError());
            slow_list = list2;
            slow_n_dev = c->counters + 1;   // (the list's length stays on the device; novf bounds it)
            if (c->opt_dbg & 4) {
                int m2 = 0;
                if (int rc = read_counter(c, 1, &m2)) return rc;
                fprintf(stderr, "[refine] %d retraces listed, %d left for the exact slow kernel after the %d-voxel window\n", novf, m2, XB_MID_K);
            }
        }
        if (int rc = run_slow(c, n_slow, 1, nullptr, fs + FS_CHANGED, fs + FS_ESCAPED, slow_list, slow_n_dev, true)) return rc;
        HIPCHK(hipMemcpyAsync(hr, fs + FS_CHANGED, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        *changed = hr[0];
    }
    return XB_OK;
}

static int refine_impl(xb_ctx *c, int mode, int64_t iters, int64_t *log, int64_t log_capacity, int64_t *n_iters);
int xb_refine(xb_ctx *c, int mode, int64_t iters, int64_t *log, int64_t log_capacity, int64_t *n_iters) {
    const int rc = refine_impl(c, mode, iters, log, log_capacity, n_iters);
    // the changed-voxel list in `stage` is private to this call's iterations: a later xb_edge_check through the C ABI compacts
    // known == -2 itself instead of trusting a list that other calls (downloads through `stage`, xb_edge_find) may have clobbered
    if (c) c->chg_n = -1;
    return rc;
}
static int refine_impl(xb_ctx *c, int mode, int64_t iters, int64_t *log, int64_t log_capacity, int64_t *n_iters) {
    NEED_GRID("xb_refine");
    if (n_iters) *n_iters = 0;
    if (iters == 0) return XB_OK;  // thread_handlers.py:146-147
    int64_t edges = 0, changed = 0, esc = 0, checked = 0;
    const bool fused = c->g.x0 == 0 && c->g.x1 == c->g.nx && !table_windowed(c);
    // After an ongrid assignment (no table yet, the bricks' maxima known, no vacuum) the first iteration builds the records
    // itself, after its edge sweep and only where the band is; otherwise the table comes first (the sweep reads it).
    const bool late = fused && !c->grad_valid && c->brick_max_valid && !c->has_vacuum && c->brick_rec && c->g.nx >= 16 && c->g.ny >= 16 && c->g.nz >= 16;
    if (!late)
        if (int rc = xb_prepare_refine(c)) return rc;
    int64_t k = 0;
    auto put = [&](int64_t e, int64_t ch) {
        if (log && 2 * k + 1 < log_capacity) { log[2 * k] = e; log[2 * k + 1] = ch; }
        k++;
        if (n_iters) *n_iters = k;
    };
    if (fused) {
        if (int rc = refine_iteration_fused(c, &edges, &changed, late)) return rc;
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

// bader_calc + refine in ONE call (round 5): what Bader.__call__ does back to back (interface.py:471-490), with the refinement's first
// iteration queued behind the assignment -- one host wait for both instead of two, and no idle card between them.  Same results
// and the same log as xb_assign followed by xb_refine; whenever the assignment does not end the usual way (a tie voxel, walkers for
// the exact slow path, the growth's long schedule, numbering on the host) or the combination is not the fused one-GPU neargrid path
// without vacuum, the two ordinary calls run instead.
int xb_assign_refine(xb_ctx *c, int method, int mode, int64_t iters, int64_t *n_maxima, int64_t *log, int64_t log_capacity, int64_t *n_iters) {
    NEED_GRID_RAW("xb_assign_refine");
    if (n_iters) *n_iters = 0;
    const bool fused = c->g.x0 == 0 && c->g.x1 == c->g.nx && !table_windowed(c);
    if (method == XB_METHOD_NEARGRID && fused && fused_ok(c) && !c->has_vacuum && iters != 0 && (mode == XB_REFINE_ALL || mode == XB_REFINE_CHANGED)) {
        c->labels_zero_pending = false;   // every label is overwritten, none is read
        c->defer_wait = true;
        int rc = assign_neargrid_fused(c, nullptr);
        c->defer_wait = false;
        if (rc) { c->pending_assign = false; return rc; }
        rc = refine_impl(c, mode, iters, log, log_capacity, n_iters);
        c->chg_n = -1;
        const bool redo = rc == XB_STEP_REDO || c->pending_assign;   // (pending still: the refinement returned before its first wait)
        const bool arrived = rc == XB_STEP_REDO;                     // the assignment's state block is on the host
        c->pending_assign = false;
        if (!redo) {
            if (n_maxima) *n_maxima = c->pending_n_maxima;
            return rc;
        }
        if (n_iters) *n_iters = 0;
        if (arrived) {
            // The queued iteration did nothing: its tile list is gated on the assignment's outcome (k_edge_tile_list), no tile means no
            // edge and no retrace.  So the assignment is where its own wait would have found it -- the host part takes over (walkers
            // for the slow path, numbering on the host, or the whole assignment once more with the long kill schedule), then the
            // refinement as an ordinary call.  (Round 5: this used to repeat the assignment from the start -- 11.2 instead of 5.6 ms
            // per step on a 216-atom cell, whose 19 K undecidable walkers make every step end here.)
            c->grad_rule = 1;
            c->buni_valid = false; c->regions_labels = false;
            c->regions_pending = true;
            c->list_valid = false;
            if (int rc2 = assign_neargrid_tail(c, n_maxima)) return rc2;
            return xb_refine(c, mode, iters, log, log_capacity, n_iters);
        }
        c->grad_valid = false;   // the ordinary way, from the start: the assignment rewrites every label
        c->first_clean = false;
        c->buni_valid = false; c->regions_labels = false;
        if (n_iters) *n_iters = 0;
    }
    if (int rc = xb_assign(c, method, n_maxima)) return rc;
    return xb_refine(c, mode, iters, log, log_capacity, n_iters);
}
