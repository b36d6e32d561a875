// host_sums.h -- host side, part 5: charge / volume sums, volume_assign, atom assignment, surface distance, masks.


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
    long long top = 0;
    for (int64_t i = 0; i < n_swap; i++) { s[i] = (int)swap[i]; top = std::max<long long>(top, std::llabs((long long)swap[i]) + 1); }
    c->label_wire = label_wire_for(top);
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
