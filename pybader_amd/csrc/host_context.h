// host_context.h -- host side of libbader_hip.so, part 1 (included by bader_hip.hip inside its extern "C" block):
// context life cycle, grid set-up and the scratch / table allocations, density and label transfers (pinned double buffers),
// the CHGCAR text parser's driver, the device density generator, vacuum assignment.

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
    HIPCHK(hipMalloc(&c->counters, (1024 + XB_SORT_MAX) * sizeof(int)));   // (counters, the state block fs, and behind it the sorted maxima for the one copy of an assignment's wait)
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
    hipFree(c->st); hipFree(c->stage); hipFree(c->ec_pend); c->ec_pend = nullptr; hipFree(c->ec_share); c->ec_share = nullptr; hipFree(c->ec_pflag); c->ec_pflag = nullptr; hipFree(c->max_list); hipFree(c->max_aux); hipFree(c->ovf_list);
    hipFree(c->blab_buf); c->blab_buf = nullptr; c->blab_alloc = 0; c->labels_zero_pending = false;
    hipFree(c->ec_buf); c->ec_buf = nullptr; c->ec_buf_cap = 0; c->grad_cap = 0; c->list_cap = 0;
    c->brick_rec = nullptr; c->grad_cover = 0;
    c->rho = nullptr; c->grad = nullptr; c->grad_valid = false; c->brick_max_valid = false; c->labels = nullptr; c->known = nullptr; c->first = nullptr; c->list = nullptr;
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
    if (own == g.nx || planes >= g.nx) planes = g.nx;
    const long long list_want = planes == g.nx ? N : std::max<long long>(planes * g.nyz, 8 * (N / 512) + 4096);
    const size_t stage_want = planes == g.nx ? (size_t)N * 8 : std::max<size_t>((size_t)planes * g.nyz * 8, (size_t)64 << 20);
    if (c->list_cap < list_want) {
        HIPCHK(hipStreamSynchronize(c->stream));
        hipFree(c->list); c->list = nullptr; c->list_cap = 0;
        HIPCHK(hipMalloc(&c->list, (size_t)list_want * sizeof(int)));
        c->list_cap = list_want;
        c->list_valid = false; c->chg_n = -1; c->walk = nullptr; c->n_walk = 0;
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
        hipFree(c->grad); c->grad = nullptr; c->grad_cap = 0; c->grad_valid = false; c->brick_max_valid = false;
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
        c->ovf_cap = (int)std::min<long long>(N, std::max<long long>(1 << 22, N / 16));   // (walkers for the exact slow path; beyond it the assignment lists them again, cap by cap)
        HIPCHK(hipMalloc(&c->ovf_list, c->ovf_cap * sizeof(int)));
        c->n_alloc = N;
        c->first_clean = false;
    }
    c->zero_outside[0] = -1;
    Grid &g = c->g;
    if (g.nx != (int)shape[0] || g.ny != (int)shape[1] || g.nz != (int)shape[2]) { c->grad_valid = false; c->brick_max_valid = false; }
    if (dist_mat && !T_grad) return fail(XB_E_ARG, "xb_set_grid: dist_mat without T_grad");
    g.nx = (int)shape[0]; g.ny = (int)shape[1]; g.nz = (int)shape[2];
    g.nyz = g.ny * g.nz;
    g.x0 = (int)x0; g.x1 = (int)x1;
    if (dist_mat) {
        if (memcmp(g.dist, dist_mat, sizeof g.dist) != 0) { c->grad_valid = false; c->brick_max_valid = false; }   // the tabulated ongrid successors depend on it
        memcpy(g.dist, dist_mat, sizeof g.dist);
        HIPCHK(hipMemcpyAsync(c->dist_dev, dist_mat, sizeof g.dist, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->dist_dev + 27, T_grad, sizeof g.T, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (T_grad && memcmp(g.T, T_grad, sizeof g.T) != 0) { memcpy(g.T, T_grad, sizeof g.T); c->grad_valid = false; c->brick_max_valid = false; }
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
    c->label_wire = 4;
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
// page-locked host memory for the caller's result arrays (round 5: the Python layer hands out label arrays that live in such
// buffers and recycles them -- a device-to-host copy lands in them directly, no staging copy, no first-touch page faults)
int xb_host_alloc(int64_t bytes, void **out) {
    if (!out || bytes <= 0) return fail(XB_E_ARG, "xb_host_alloc: bad argument");
    void *p = nullptr;
    HIPCHK(hipHostMalloc(&p, (size_t)bytes));
    *out = p;
    return XB_OK;
}
int xb_host_free(void *p) {
    if (p) HIPCHK(hipHostFree(p));
    return XB_OK;
}
static bool host_pinned(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}
static int staged_d2h(xb_ctx *c, void *dst_host, const void *src_dev, size_t bytes) {   // returns with the data on the host
    if (bytes < (4u << 20) || host_pinned(dst_host)) {
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
    if (c) c->vac_by_tol = false;   // (the -1 labels no longer say "rho <= vac_tol" of the density on the card)
    NEED_GRID("xb_upload_density");
    c->grad_valid = false; c->brick_max_valid = false;
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
    if (c) c->vac_by_tol = false;   // (the -1 labels no longer say "rho <= vac_tol" of the density on the card)
    NEED_GRID("xb_parse_density_text");
    const Grid &g = c->g;
    if (!text || nbytes <= 0) return fail(XB_E_ARG, "xb_parse_density_text: empty text");
    if (nbytes / (TPB * TXT_BYTES) >= (1LL << 31) - 2) return fail(XB_E_LIMIT, "xb_parse_density_text: text too large");
    if (!(divisor == divisor) || divisor == 0.) return fail(XB_E_ARG, "xb_parse_density_text: bad divisor");
    c->grad_valid = false; c->brick_max_valid = false;
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
    if (c) c->vac_by_tol = false;   // (the -1 labels no longer say "rho <= vac_tol" of the density on the card)
    NEED_GRID("xb_synth_density");
    if (n_atoms < 0 || n_atoms > 4096) return fail(XB_E_ARG, "xb_synth_density: bad atom count");
    c->grad_valid = false; c->brick_max_valid = false;
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
    c->list_valid = false; c->chg_n = -1;
    c->has_vacuum = true;
    c->vac_by_tol = false;
    c->buni_valid = false; c->regions_labels = false;
    const size_t sz = dtype_size(dtype);
    if (!sz) return fail(XB_E_ARG, "xb_upload_labels: bad dtype code %d", dtype);
    c->label_wire = sz >= 4 ? 4 : (int)sz;   // what the caller's own dtype holds, the narrowed halo holds
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
    } else if (void *dev_view = nullptr; host_pinned(labels_host) && hipHostGetDevicePointer(&dev_view, labels_host, 0) == hipSuccess && dev_view) {
        // a page-locked destination (pybader_amd's pooled result arrays): the narrowing kernel writes it over the bus itself --
        // no staging buffer, no second transfer (round 5: 0.8 -> ~0.4 ms for the 16 MB of a 256^3 map)
        if (dtype == XB_I8 || dtype == XB_I16) {
            const long long per = 16 / (long long)sz, n16 = c->N / per, done = n16 * per;
            if (dtype == XB_I8) {
                if (n16) k_narrow_vec<int8_t><<<nblocks(n16), TPB, 0, c->stream>>>(c->labels, (int8_t *)dev_view, n16);
                if (done < c->N) k_narrow<int8_t><<<1, TPB, 0, c->stream>>>(c->labels + done, (int8_t *)dev_view + done, c->N - done);
            } else {
                if (n16) k_narrow_vec<int16_t><<<nblocks(n16), TPB, 0, c->stream>>>(c->labels, (int16_t *)dev_view, n16);
                if (done < c->N) k_narrow<int16_t><<<1, TPB, 0, c->stream>>>(c->labels + done, (int16_t *)dev_view + done, c->N - done);
            }
        } else
            k_narrow<long long><<<nblocks(c->N), TPB, 0, c->stream>>>(c->labels, (long long *)dev_view, c->N);
        HIPCHK(hipGetLastError());
    } else {
        c->chg_n = -1;   // (the narrowing goes through `stage`, whose upper half may list the changed voxels)
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
    c->list_valid = false; c->chg_n = -1;
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
    c->label_wire = 1;   // every valid plane holds 0 / -1 from here on
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
        c->vac_by_tol = false;
        if (vac_charge) *vac_charge = 0.;
        if (vac_volume) *vac_volume = 0.;
        return XB_OK;
    }
    c->zero_outside[0] = -1;
    HIPCHK(hipMemsetAsync(c->dsum, 0, sizeof(double), c->stream));
    HIPCHK(hipMemsetAsync(c->counters64, 0, sizeof(unsigned long long), c->stream));
    k_vacuum_assign<<<(unsigned)std::min<long long>(nblocks(c->N), 4096), TPB, 0, c->stream>>>(c->g, c->rho, c->labels, vac_tol, c->dsum, c->counters64);
    HIPCHK(hipGetLastError());
    double s;
    unsigned long long n;
    HIPCHK(hipMemcpyAsync(&s, c->dsum, sizeof s, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&n, c->counters64, sizeof n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->has_vacuum = true;   // (the count covers the owned slab only: stay conservative)
    c->vac_by_tol = true; c->vac_tol = vac_tol;
    if (vac_charge) *vac_charge = s * voxel_volume;  // utils.py:400
    if (vac_volume) *vac_volume = (double)n * voxel_volume;
    return XB_OK;
}

