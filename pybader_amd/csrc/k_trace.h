// k_trace.h -- device kernels of libbader_hip.so: neargrid assignment: region fill, walker trace, exact slow path.
// Included by bader_hip.hip (one translation unit); see bader_kernels.h for the common device code.
#pragma once

// ---------------------------------------------------------------------------------------------
// neargrid assignment: every owned non-vacuum voxel follows its own dr=0 trajectory
// (refinement.py:17-322 stepping rules without the early stop) to the maximum it reaches.
// One lane per voxel, lanes along z (coalesced first loads).  labels: in 0/-1, out = linear index
// of the maximum (-1 vacuum, -2 = handed to the exact slow kernel).
// ---------------------------------------------------------------------------------------------
// Voxels inside a trapping region end at its maximum: fill their labels in one streaming sweep
// (vacuum voxels keep -1; a region whose maximum is vacuum hands out -1, refinement.py:286) and note
// the maxima for the numbering.  The uncertain bricks go to the work list of k_ng_trace.
__global__ __launch_bounds__(TPB) void k_fill_certain(GridL g, const int *__restrict__ blab, int nb1, int nb2,
                                                      const int *__restrict__ box_max, int *labels, int *first,
                                                      int *max_list, int *max_count, int max_cap, const int *skip = nullptr) {
    if (skip && *skip) return;
    const int vbeg = g.x0 * g.nyz, vend = g.x1 * g.nyz;
    const int v = vbeg + blockIdx.x * TPB + threadIdx.x;
    const bool in = v < vend;
    int result = -1;
    bool has = false;
    if (in) {
        const int x = v / g.nyz;
        const int r = v - x * g.nyz;
        const int y = r / g.nz, z = r - y * g.nz;
        const int b = blab[((x >> 3) * nb1 + (y >> 3)) * nb2 + (z >> 3)];
        if (b > 0 && labels[v] != -1) {
            result = box_max[b - 1];
            if (result != v && labels[result] == -1) result = -1;
            labels[v] = result;
            has = result >= 0;
        }
    }
    note_maximum_wave(has, result, in ? v : 0, first, max_list, max_count, max_cap);
}
// Without vacuum every voxel of a certain brick belongs to its maximum and the smallest voxel
// index of a brick is its corner: the numbering needs one note per owned certain brick, and the
// labels themselves are written by k_relabel_regions after the trace.
__global__ void k_note_certain_bricks(GridL g, int nb0, int nb1, int nb2, int b_lo, int b_hi,
                                      const int *__restrict__ blab, const int *__restrict__ box_max, int *first,
                                      int *max_list, int *max_count, int max_cap, const int *skip = nullptr) {
    if (skip && *skip) return;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int l = (b < nb0 * nb1 * nb2 && b >= b_lo && b < b_hi) ? blab[b] : 0;
    const bool has = l > 0;
    const int b0 = b / (nb1 * nb2), b1 = (b / nb2) % nb1, b2 = b % nb2;
    // one atomic per distinct maximum per wave (a handful of maxima own all the bricks)
    note_maximum_wave(has, has ? box_max[l - 1] : 0, ((b0 * 8) * g.ny + b1 * 8) * g.nz + b2 * 8, first, max_list,
                      max_count, max_cap);
}
// labels := rank of the maximum; voxels of certain bricks take it from the brick label, the others from the maximum index the
// trace left in `labels`.
// Brick-shaped version: a thread owns the 8 y-rows of one brick at one (x, z / V): ONE brick-label lookup (through a
// per-block LDS table of the region ranks) for 8 stores of V labels; a block covers 4 x-planes x 64 V voxels of z.
// V = 4: 16-byte stores (z extents that are multiples of 4: rows are 16-byte aligned); V = 1: any z extent -- 4-byte
// stores, a wave writes 256 contiguous bytes of a row (round 4; before, such grids took the per-voxel k_relabel: three
// integer divisions and a brick lookup per voxel, 0.48 instead of 0.14 ms at 500^3).
template <int V>
__global__ __launch_bounds__(TPB) void k_relabel_regions_brick(GridL g, int *labels, const int *__restrict__ rank,
                                                               const int *__restrict__ blab, int nb1, int nb2,
                                                               const int *__restrict__ box_max, const int *__restrict__ fs,
                                                               const int *gate, int n_boxes = -1) {
    static_assert(V == 1 || V == 4, "a label or four of them per store");
    struct alignas(4 * V) Vec { int l[V]; };
    __shared__ int s_rank[XB_BOXES_MAX];
    if (gate && !*gate) return;
    // (n_boxes: the number of regions when the caller knows it -- a slab, whose planes x0..x1 the launch covers)
    for (int i = threadIdx.x; i < min(n_boxes >= 0 ? n_boxes : fs[FS_N_BOXES], XB_BOXES_MAX); i += TPB) s_rank[i] = rank[box_max[i]];
    __syncthreads();
    const int z = V * (blockIdx.x * 64 + (threadIdx.x & 63)), by = blockIdx.y, x = g.x0 + blockIdx.z * 4 + (threadIdx.x >> 6);
    if (z >= g.nz || x >= g.x1) return;
    const int b = blab[((x >> 3) * nb1 + by) * nb2 + (z >> 3)];
    Vec *p = reinterpret_cast<Vec *>(labels + ((size_t)(x * g.ny + by * 8) * g.nz + z));
    const int stride = g.nz / V;   // vectors per row
    const int rows = min(8, g.ny - by * 8);   // (a brick the grid cuts in y: its rows inside the grid)
    if (b > 0) {
        const int l = b <= XB_BOXES_MAX ? s_rank[b - 1] : rank[box_max[b - 1]];
        Vec v;
#pragma unroll
        for (int k = 0; k < V; k++) v.l[k] = l;
#pragma unroll
        for (int r = 0; r < 8; r++)
            if (r < rows) p[(size_t)r * stride] = v;
    } else {
        Vec m[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (r < rows) m[r] = p[(size_t)r * stride];
            else
#pragma unroll
                for (int k = 0; k < V; k++) m[r].l[k] = -1;
        }
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int k = 0; k < V; k++)
                if (m[r].l[k] >= 0) m[r].l[k] = rank[m[r].l[k]];
#pragma unroll
        for (int r = 0; r < 8; r++)
            if (r < rows) p[(size_t)r * stride] = m[r];
    }
}
// 16 bricks per thread, one atomic per block; the list keeps brick order inside a block's range
__global__ __launch_bounds__(TPB) void k_brick_walk_list(int nbr, int b_lo, int b_hi, const int *__restrict__ blab,
                                                         int *walk, int *n_walk, const int *skip = nullptr) {
    if (skip && *skip) return;
    const int base = (blockIdx.x * TPB + threadIdx.x) * 16;   // bricks [b_lo, b_hi) are the owned slab
    unsigned int hits = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int b = base + k;
        if (b < nbr && b >= b_lo && b < b_hi && blab[b] <= 0) hits |= 1u << k;
    }
    int total;
    const int off = block_scan_excl(__popc(hits), total);
    __shared__ int base_s;
    if (threadIdx.x == 0) base_s = total ? atomicAdd(n_walk, total) : 0;
    __syncthreads();
    int w = base_s + off;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if ((hits >> k) & 1u) walk[w++] = base + k;
}

// The same list in MORTON order of the brick coordinates (single GPU): the waves of the persistent trace work on
// consecutive list entries at the same time, and a Z-order run of bricks is a compact 3-D neighbourhood -- its
// trajectories share table lines in the XCD's L2 -- where a run in linear order is a thin row of bricks.
__device__ __forceinline__ int compact3(unsigned x) {   // every third bit of x, packed
    x &= 0x09249249u;
    x = (x ^ (x >> 2)) & 0x030c30c3u;
    x = (x ^ (x >> 4)) & 0x0300f00fu;
    x = (x ^ (x >> 8)) & 0xff0000ffu;
    x = (x ^ (x >> 16)) & 0x000003ffu;
    return (int)x;
}
// Round 6, VACUUM BRICKS: with a vacuum tolerance a brick whose largest density lies below it holds no voxel to assign -- and on
// a noisy vacuum (what a tolerance is for) no region certifies it either, so it used to get 512 records and eight wave-loads
// of walkers that leave at once (1.4 + 0.4 ms of a 5.7 ms step at 512^3).  `bpot` (pass A's brick potentials: the brick's
// largest density as a float in integer order) says which bricks these are; they stay off the list and their brick label
// becomes XB_NOREC: "no records here" -- a walker that steps into one (downhill across the tolerance: not excluded) is handed
// to the exact slow kernel, which works from rho.  The float is compared with a margin that covers its rounding.
#define XB_NOREC (-2147483647 - 1)
__device__ __forceinline__ bool brick_is_vacuum(const int *__restrict__ bpot, int b, double vac_tol) {
    if (!bpot) return false;
    const int p = bpot[b];
    const double f = (double)__int_as_float(p >= 0 ? p : p ^ 0x7fffffff);
    return f + fabs(f) * 1e-6 < vac_tol;
}
__global__ __launch_bounds__(TPB) void k_brick_walk_list_morton(int nb0, int nb1, int nb2, unsigned n_codes,
                                                                int *blab, int *walk, int *n_walk, const int *skip,
                                                                const int *__restrict__ bpot = nullptr, double vac_tol = 0.) {
    if (skip && *skip) return;   // (the region growth asks for a repeat: no list, nothing to trace)
    const unsigned base = (blockIdx.x * TPB + threadIdx.x) * 16u;   // 16 consecutive Morton codes per thread
    unsigned int hits = 0;
    int bidx[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const unsigned code = base + k;
        const int b2 = compact3(code), b1 = compact3(code >> 1), b0 = compact3(code >> 2);
        const bool in = code < n_codes && b0 < nb0 && b1 < nb1 && b2 < nb2;
        bidx[k] = in ? (b0 * nb1 + b1) * nb2 + b2 : 0;
        if (in && blab[bidx[k]] <= 0) {
            if (brick_is_vacuum(bpot, bidx[k], vac_tol)) blab[bidx[k]] = XB_NOREC;
            else hits |= 1u << k;
        }
    }
    int total;
    const int off = block_scan_excl(__popc(hits), total);
    __shared__ int base_s;
    if (threadIdx.x == 0) base_s = total ? atomicAdd(n_walk, total) : 0;
    __syncthreads();
    int w = base_s + off;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if ((hits >> k) & 1u) walk[w++] = bidx[k];
}

__device__ __forceinline__ void og_offsets(int og, int &ox, int &oy, int &oz) {
    ox = og / 9 - 1; oy = (og / 3) % 3 - 1; oz = og % 3 - 1;
}

// The walkers of one wave: lane -> start voxel (sx,sy,sz); `in_walk`: the start lies in a brick of the walk list
// (uncertain by construction).  WIN: the table covers only a window of the grid (slabs); records outside it are
// derived from rho on the spot.  The single-GPU instantiation leaves that (register hungry) path out.
template <int K, bool WIN>
__device__ __forceinline__ void ng_walk_wave(const GridL &g, const GradRec *__restrict__ G, const int *__restrict__ box_max,
                                             const int *__restrict__ blab, int nb1, int nb2, bool in_walk, int sx, int sy, int sz,
                                             int *labels, int *first, int *max_list, int *max_count, int max_cap,
                                             int *ovf_list, int *ovf_count, int ovf_cap, int maxsteps,
                                             const double *__restrict__ rho, const double *__restrict__ gc,
                                             bool has_vacuum = true) {
    const bool valid = sx < g.x1 && sy < g.ny && sz < g.nz;
    const int v = valid ? (sx * g.ny + sy) * g.nz + sz : 0;
    bool moving = false;
    int result = -1;
    int px = 0, py = 0, pz = 0, lp = 0, steps = 0;
    double dr0 = 0., dr1 = 0., dr2 = 0.;
    GradRec rec = {0., 0., 0., 0.};
    PathWindow<K> w;
    w.init(0, 0.);
    if (valid) {
        // the three start loads travel together (label for the vacuum test, brick label, own record); a start
        // voxel from the work list lies in an uncertain brick by construction
        const int lab0 = has_vacuum ? labels[v] : 0;   // without vacuum `labels` is write-only here
        int b = (blab && !in_walk) ? blab[((sx >> 3) * nb1 + (sy >> 3)) * nb2 + (sz >> 3)] : 0;
        rec = fetch_rec_w(g, G, v);
        if (lab0 != -1) {
            px = sx; py = sy; pz = sz;
            lp = v;
            // trapping regions: brick labels (grids made of whole 8^3 bricks) or box ids in the keys
            if (b <= 0) b = key_box(rec.key);
            if (b > 0) result = box_max[b - 1];  // starts inside a trapping region: ends at its maximum
            else { w.init(v, rec.key); moving = true; }
        }
    }
    while (__any(moving)) {
        if (moving) {
            const int bits = key_bits(rec.key);
            const int code = bits & 63;
            int qx, qy, qz, lq = 0;
            // methods.py:345-363 / refinement.py:132-154: the gradient move (if the voxel has one)
            bool og_move = (code == XB_STAY_CODE);
            if (!og_move) {
                ng_move_t(g, px, py, pz, rec, code, dr0, dr1, dr2, qx, qy, qz);
                lq = lin3f(g, qx, qy, qz);
                og_move = w.contains(lq);  // methods.py:411 / refinement.py:200: already been here on this path
            }
            if (og_move) {  // methods.py:412-447 / refinement.py:201-235: dr = 0 and one ongrid step from p (tabulated)
                const int og = (bits >> 6) & 31;
                if (og == XB_OG_SELF) { result = lp; moving = false; }  // break_flag: p is the maximum
                else {
                    int ox, oy, oz;
                    og_offsets(og, ox, oy, oz);
                    dr0 = dr1 = dr2 = 0.;
                    qx = wrap_u(px + ox, g.nx); qy = wrap_u(py + oy, g.ny); qz = wrap_u(pz + oz, g.nz);
                    lq = lin3f(g, qx, qy, qz);
                }
            }
            if (moving) {
                const bool in_win = plane_in_window(g, qx);  // the table only exists inside the window (slabs)
                GradRec nr = fetch_rec(G, in_win ? rec_slot(g, lq) : 0);  // issued before the brick label: both in flight (outside the window: any valid slot)
                const int bl = blab ? blab[((qx >> 3) * nb1 + (qy >> 3)) * nb2 + (qz >> 3)] : 0;
                if (WIN && !in_win && bl <= 0) nr = make_rec_rho(g, rho, gc, qx, qy, qz);  // outside the window: from rho
                const int b = bl > 0 ? bl : key_box(nr.key);
                if (bl == XB_NOREC) {  // a vacuum brick (k_brick_walk_list_morton): no records -- the exact slow kernel works from rho
                    result = -2;
                    moving = false;
                } else if (b) {  // arrived inside a trapping region (q cannot be an old path voxel: the
                    result = box_max[b - 1];  // trajectory would have stopped there already)
                    moving = false;
                } else if ((!WIN && !in_win) || (!og_move && nr.key <= w.m_old) || ++steps > maxsteps) {
                    result = -2;  // membership undecidable: exact slow kernel
                    moving = false;  // (ongrid moves are appended without a membership test, 305-315)
                } else {
                    w.push(lq, nr.key);
                    px = qx; py = qy; pz = qz; lp = lq; rec = nr;
                }
            }
        }
    }
    // a maximum that is itself vacuum hands its -1 to the start voxel (methods.py:449-452 / refinement.py:286)
    if (has_vacuum && valid && result >= 0 && result != v && labels[result] == -1) result = -1;
    if (valid) labels[v] = result;
    note_maximum_wave(valid && result >= 0, result, v, first, max_list, max_count, max_cap);
    if (valid && result == -2) {
        const int k = atomicAdd(ovf_count, 1);
        if (k < ovf_cap) ovf_list[k] = v;
    }
}
// ---------------------------------------------------------------------------------------------
// The lean walker of the single-GPU persistent trace.  Same trajectory, same exits and the same results as
// ng_walk_wave<2, false> -- tests/test_gpu_parity.py runs both -- for the case that kernel is launched in: the table
// window is the whole grid, trapping regions are brick labels (no box ids in the keys), nx * ny and nz below 2^24.
// What differs is the instruction stream of a step (round 2: ~100 VALU + 48 SALU per wave-step, 87 % of the kernel's
// time in VALU issue):
//   * the gradient move is computed for every moving lane, straight-line; the rare ongrid step (a voxel without a
//     gradient step, or the move lands on the last two path voxels: methods.py:411) sits behind ONE wave-uniform
//     branch and un-does the move from the move's own offsets instead of keeping the old coordinates alive;
//   * the coordinates are updated in place, the new record is loaded into the registers of the old one (no copies
//     at the back-edge), linear and brick indices by 24-bit multiply-adds (the generic form compiles to the
//     quarter-rate v_mad_u64_u32), the periodic wrap is one v_min3_u32 of (q, q + n, q - n);
//   * no window / table-cover tests.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int wrap3(int q, int n) {   // q in [-n, 2n) -> [0, n)
    return (int)min(min((unsigned)q, (unsigned)(q + n)), (unsigned)(q - n));
}
__device__ __forceinline__ int mad24(int a, int b, int c) {   // a * b + c, a and b below 2^24 (one full-rate instruction)
    int r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int lin24(const GridL &g, int x, int y, int z) { return mad24(mad24(x, g.ny, y), g.nz, z); }
// OFF32: every byte offset into the table (32 B per voxel) fits 32 bits (N <= 2^27): scalar base + 32-bit lane offset
template <bool OFF32>
__device__ __forceinline__ GradRec fetch_rec_o(const GradRec *__restrict__ G, int l) {
    if (OFF32) return *reinterpret_cast<const GradRec *>(reinterpret_cast<const char *>(G) + ((unsigned)l << 5));
    return fetch_rec(G, l);
}
// CACHE: the 512 records of the walkers' OWN brick sit in LDS (s_rec, loaded by the workgroup before its eight waves start:
// k_ng_trace_g) -- the start record and every step that stays inside the brick (about 40 % of the fetches) read them there,
// and the own brick, a walk-list brick, needs no brick-label lookup either.  The trace is bound by its L2 requests.
// WINDOW (slabs): the table covers the planes of a window only (rec_slot); a walker that steps out of it without landing in a
// trapping region ends with -2 like an undecidable one -- its start voxel goes to `ovf_list`, here the list of the
// trajectories the caller redoes with records derived from rho (k_ng_trace_list)
// (the cache is a file-scope LDS array, not a pointer argument: handed over as a generic pointer it made the gfx950 backend of
// this ROCm emit `v_cmp_ne_u32 0, src_shared_base` -- "illegal instruction, operand has incorrect register class" -- for
// some shapes of the surrounding code)
// The shape of a workgroup of the group trace: XB_TRACE_WAVES waves share out the eight eighths (brick_sub_voxel) of ONE brick.  What counts
// is eighths per wave (two: a wave that drew a short one takes another instead of waiting at the barrier) and WAVES PER BARRIER
// (the slowest of them sets the pace).  Measured at 512^3 (round 5): 8 waves x 1 brick 1.22 ms, 8 x 2 1.175, 16 x 4 1.29, 8 x 3 1.58
// (48 KB of LDS: three workgroups per compute unit), **4 x 1 1.135**, 4 x 2 1.56, 2 x 1 1.45 (the last two lose occupancy to LDS).
constexpr int XB_TRACE_WAVES = 4;
__shared__ GradRec xb_s_rec[512];   // the records of the brick a workgroup of the group trace walks (16 KB)
template <bool OFF32, bool CACHE, bool WINDOW = false>
__device__ __forceinline__ int ng_walk_lean(const GridL &g, const GradRec *__restrict__ G, const int *__restrict__ box_max,
                                             const int *__restrict__ blab, int nb1, int nb2, int sx, int sy, int sz,
                                             int *labels, int *first, int *max_list, int *max_count, int max_cap,
                                             int *ovf_list, int *ovf_count, int ovf_cap, int maxsteps, bool has_vacuum) {
    // every start voxel is valid: the walk list holds whole bricks, and the lanes of a brick the grid cuts start from its
    // voxels inside the grid (k_ng_trace_g PART)
    const int v = lin24(g, sx, sy, sz);
    const int lab0 = has_vacuum ? labels[v] : 0;   // without vacuum `labels` is write-only here
    const int ox8 = sx & ~7, oy8 = sy & ~7, oz8 = sz & ~7;   // origin of the own brick
    GradRec rec = CACHE ? xb_s_rec[((sx & 7) << 6) | ((sy & 7) << 3) | (sz & 7)] : fetch_rec_o<OFF32>(G, WINDOW ? rec_slot(g, v) : v);
    int result = -1;
    int px = sx, py = sy, pz = sz, steps = 0;
    double dr0 = 0., dr1 = 0., dr2 = 0.;
    // PathWindow<2> by hand: (i0, k0) the current voxel and its key, (i1, k1) the one before, m_old the largest older key.
    // (Round 5, measured: the running-maximum test fails wherever a trajectory dips below a key it passed two or more steps ago --
    // at 512^3 19 K walkers of a 216-atom cell and 5.8 M of a noisy vacuum go to the exact slow path with this window of 2, 7 K /
    // 2.6 M with 3, 2.3 K / 1.2 M with 4; the headline pays 1.5 % / 3 % of the trace for a wider one, so 2 it is and the slow path
    // got its tiers instead: host_assign.h run_slow.)
    int i0 = v, i1 = -1;
    double k0 = rec.key, k1 = -1.7976931348623157e308, m_old = -1.7976931348623157e308;
#ifdef XB_DEBUG_COUNT
    int dbg_lane_steps = 0;
#endif
    // Round 6: the loop in its natural divergent form -- a lane that has arrived leaves the loop and drops out of the execution
    // mask.  The round-3 form `while (ballot(moving)) if (moving) {...}` kept `moving` as a per-lane flag that was materialised and
    // tested at the head of every step, and -- what cost more -- kept every lane's `result` live across the back edge: the load of
    // box_max[] for the lanes that had just arrived was waited for at the head of the NEXT step, a second memory round trip in
    // the chain of every step in which some lane arrived.  Now that load is waited for after the loop (1.14 -> 1.05 ms at 512^3).
    if (lab0 != -1) for (;;) {
        const int bits = key_bits(rec.key);
        // methods.py:345-363: dr += r; corr = rha(dr); q = p + int_grad + corr; dr -= corr (a voxel without a gradient
        // step has code 63 and r = 0: the move below is garbage for it and replaced by the ongrid step)
        const double t0 = dr0 + rec.r0, t1 = dr1 + rec.r1, t2 = dr2 + rec.r2;
        const int id0 = rha_cs(t0), id1 = rha_cs(t1), id2 = rha_cs(t2);
        const int m0 = (bits & 3) + id0 - 1, m1 = ((bits >> 2) & 3) + id1 - 1, m2 = ((bits >> 4) & 3) + id2 - 1;
        px += m0; py += m1; pz += m2;
        dr0 = t0 - (double)id0; dr1 = t1 - (double)id1; dr2 = t2 - (double)id2;
        // the periodic wrap only for the waves that touch the faces of the grid
        if (__builtin_amdgcn_ballot_w64((unsigned)px >= (unsigned)g.nx || (unsigned)py >= (unsigned)g.ny || (unsigned)pz >= (unsigned)g.nz) != 0) {
            px = wrap3(px, g.nx); py = wrap3(py, g.ny); pz = wrap3(pz, g.nz);
        }
        int lq = lin24(g, px, py, pz);
        const bool og_move = (bits & 63) == XB_STAY_CODE || lq == i0 || lq == i1;   // methods.py:411: already on this path
        bool at_max = false;
        if (__builtin_amdgcn_ballot_w64(og_move) != 0) {   // rare: dr = 0 and one ongrid step from p (methods.py:412-447, tabulated)
            if (og_move) {
                const int og = (bits >> 6) & 31;
                int ox, oy, oz;
                og_offsets(og, ox, oy, oz);
                at_max = og == XB_OG_SELF;   // break_flag: p is the maximum
                dr0 = dr1 = dr2 = 0.;
                px = wrap3(wrap3(px - m0, g.nx) + ox, g.nx);
                py = wrap3(wrap3(py - m1, g.ny) + oy, g.ny);
                pz = wrap3(wrap3(pz - m2, g.nz) + oz, g.nz);
                lq = lin24(g, px, py, pz);   // (== i0 at a maximum)
            }
        }
        // both loads in flight together; a lane at its maximum reloads its own record (harmless)
        int bl = 0;
        const bool own = CACHE && (unsigned)((px ^ ox8) | (py ^ oy8) | (pz ^ oz8)) < 8u;
        const bool in_win = !WINDOW || own || plane_in_window(g, px);
        if (own) rec = xb_s_rec[((px & 7) << 6) | ((py & 7) << 3) | (pz & 7)];
        else {
            rec = fetch_rec_o<OFF32>(G, WINDOW ? (in_win ? rec_slot(g, lq) : 0) : lq);   // (outside the window: any valid slot, the value is not used)
            const unsigned bidx = (unsigned)mad24(mad24(px >> 3, nb1, py >> 3), nb2, pz >> 3);
            bl = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(blab) + (bidx << 2));
        }
        steps++;
#ifdef XB_DEBUG_COUNT
        dbg_lane_steps++;
#endif
        // arrived inside a trapping region (q cannot be an old path voxel: the trajectory would have stopped there
        // already); membership undecidable from the window: exact slow kernel (ongrid moves are appended without a
        // membership test, methods.py:513-521)
        const bool arrived = bl > 0 && !at_max;
        const bool undecided = (!og_move && rec.key <= m_old) || steps > maxsteps || (WINDOW && !in_win) || bl == XB_NOREC;   // (a vacuum brick: no records)
        if (arrived) result = box_max[bl - 1];
        else if (at_max) result = i0;
        else if (undecided) result = -2;
        if (arrived || at_max || undecided) break;
        m_old = max_raw(m_old, k1);
        i1 = i0; k1 = k0;
        i0 = lq; k0 = rec.key;
    }
#ifdef XB_DEBUG_COUNT
    if (xb_dbg_steps) xb_dbg_steps[v] = (signed char)min(dbg_lane_steps, 127);   // tools/walk_lengths.py
#endif
    // a maximum that is itself vacuum hands its -1 to the start voxel (methods.py:449-452)
    if (has_vacuum && result >= 0 && result != v && labels[result] == -1) result = -1;
    labels[v] = result;
    note_maximum_wave(result >= 0, result, v, first, max_list, max_count, max_cap);
    if (result == -2) {
        const int k = atomicAdd(ovf_count, 1);
        if (k < ovf_cap) ovf_list[k] = v;
    }
    return result;
}
// lane -> voxel of the eighth `sub` of brick b: 4 x 2 x 8 voxels, z fastest -- a wave's 16 rows of 8 records, two table lines
// each.  Measured at 512^3 (round 6, k_ng_trace_g): 4x4x4 1.052 ms, 2x4x8 1.035, **4x2x8 1.017**, 1x8x8 1.09, 8x1x8 1.06, 2x8x4 1.10,
// 4x8x2 1.20, 8x8x1 1.55 (64 lines per gather instead of 16: the lines a gather instruction touches cost, whole rows are cheapest)
__device__ __forceinline__ void brick_sub_voxel(int b, int sub, int lane, int nb1, int nb2, int &sx, int &sy, int &sz) {
    const int b0 = b / (nb1 * nb2), b1 = (b / nb2) % nb1, b2 = b % nb2;
    sx = b0 * 8 + ((sub >> 2) << 2) + (lane >> 4);
    sy = b1 * 8 + ((sub & 3) << 1) + ((lane >> 3) & 1);
    sz = b2 * 8 + (lane & 7);
}
template <int K, bool WIN>
__global__ __launch_bounds__(TPB) void k_ng_trace(GridL g, const GradRec *__restrict__ G,
                                                  const int *__restrict__ box_max, const int *__restrict__ blab,
                                                  int nb1, int nb2, const int *__restrict__ walk, int n_walk,
                                                  int *labels, int *first,
                                                  int *max_list, int *max_count, int max_cap, int *ovf_list,
                                                  int *ovf_count, int ovf_cap, int maxsteps, int opt,
                                                  const double *__restrict__ rho, const double *__restrict__ gc, int has_vacuum) {
    // XCD-aware block order (opt bit 1): blocks are dealt round-robin over the 8 XCDs, each with
    // its own L2; give XCD k the k-th contiguous eighth of the work so that spatial neighbours --
    // whose trajectories read the same table lines -- share one L2.
    int blk = blockIdx.x;
    if (opt & 2) {
        const int per = gridDim.x >> 3;
        if (blk < (per << 3)) blk = (blk & 7) * per + (blk >> 3);
    }
    const int wpb = blockDim.x / XB_WAVE;  // waves per block (launch-time choice)
    const int wave = blk * wpb + threadIdx.x / XB_WAVE;
    const int lane = threadIdx.x % XB_WAVE;
    int sx, sy, sz;
    if (walk) {     // work list of the 8^3 bricks outside the trapping regions: 8 waves (an eighth each: brick_sub_voxel) per brick
        if ((wave >> 3) >= n_walk) return;
        brick_sub_voxel(walk[wave >> 3], wave & 7, lane, nb1, nb2, sx, sy, sz);
        if (sx < g.x0 || sx >= g.x1) return;  // whole wave: 4 planes of one brick half, slab edges are brick aligned or not owned
    } else if (opt & 1) {  // one wave = one 4x4x4 brick of start voxels (z fastest: 4 lanes per 128-B table line)
        const int bz_n = (g.nz + 3) >> 2, by_n = (g.ny + 3) >> 2;
        const int bx = wave / (by_n * bz_n);
        const int brem = wave - bx * (by_n * bz_n);
        const int by = brem / bz_n, bz = brem - by * bz_n;
        sx = g.x0 + bx * 4 + (lane >> 4); sy = by * 4 + ((lane >> 2) & 3); sz = bz * 4 + (lane & 3);
    } else {        // one wave = a run of 64 voxels along z
        const int rz_n = (g.nz + 63) >> 6;
        const int row = wave / rz_n;
        sz = (wave - row * rz_n) * 64 + lane;
        sx = g.x0 + row / g.ny;
        sy = row - (row / g.ny) * g.ny;
    }
    ng_walk_wave<K, WIN>(g, G, box_max, blab, nb1, nb2, walk != nullptr, sx, sy, sz, labels, first, max_list, max_count,
                         max_cap, ovf_list, ovf_count, ovf_cap, maxsteps, rho, gc, has_vacuum != 0);
}
// Slabs: the start voxels the lean kernel could not finish (their trajectory leaves the table window) once more, with
// the from-rho fallback.  The list and its length live on the device; the grid strides over it.
template <int K>
__global__ __launch_bounds__(TPB) void k_ng_trace_list(GridL g, const GradRec *__restrict__ G, const int *__restrict__ box_max,
                                                       const int *__restrict__ blab, int nb1, int nb2,
                                                       const int *__restrict__ vox, const int *n_dev, int *labels, int *first,
                                                       int *max_list, int *max_count, int max_cap, int *ovf_list,
                                                       int *ovf_count, int ovf_cap, int maxsteps,
                                                       const double *__restrict__ rho, const double *__restrict__ gc, int has_vacuum) {
    const int n = *n_dev;
    for (int base = blockIdx.x * TPB; base < n; base += gridDim.x * TPB) {   // uniform per block
        const int t = base + threadIdx.x;
        int sx = g.x1, sy = 0, sz = 0;     // (a lane without work: not valid)
        if (t < n) {
            const int v = vox[t];
            sx = v / g.nyz;
            const int r = v - sx * g.nyz;
            sy = r / g.nz;
            sz = r - sy * g.nz;
        }
        ng_walk_wave<K, true>(g, G, box_max, blab, nb1, nb2, true, sx, sy, sz, labels, first, max_list, max_count, max_cap,
                              ovf_list, ovf_count, ovf_cap, maxsteps, rho, gc, has_vacuum != 0);
    }
}
// (the XCD a workgroup runs on: each XCD takes the k-th contiguous eighth of the walk list first -- spatial neighbours read
// the same table lines, one L2 each -- and helps the others afterwards)
__device__ __forceinline__ int xcc_id() {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7;
}

// Group form of the persistent trace: a workgroup of XB_TRACE_WAVES waves pulls ONE brick (eight consecutive items) of the Morton
// ordered walk list and its waves take the brick's eight eighths (4x2x8 voxels each) one by one from a counter in LDS, so that the eighths --
// whose walkers converge onto the same voxels within a few steps -- run on ONE compute unit at the same time: their record
// loads meet in that unit's L1 (hit, or merged with the miss in flight) instead of each occupying a miss slot of a
// different unit.  The trace is bound by exactly that: L2 requests x L2 latency / misses in flight per compute unit
// (profiles/r5_final_pmc_sq_512_neargrid.txt: 9.1e7 L2 requests x 370 cycles = 50 requests in flight per compute unit on
// average, TCP_PENDING_STALL 55 % of the kernel; DESIGN.md 4.3 has what round 6 measured around that bound).
// LEAN 0: the generic walker ng_walk_wave (tests every start voxel, box ids in the keys, table windows); LEAN 3 / 4: the lean
// walker (4: 32-bit table offsets), the brick's 512 records copied into LDS (16 KB) before the waves start (see ng_walk_lean).
// CH: items per pull (8: one brick).
template <int K, int LEAN, bool WINDOW = false, bool PART = false>
__global__ __launch_bounds__(XB_WAVE * XB_TRACE_WAVES, 8) void k_ng_trace_g(GridL g, const GradRec *__restrict__ G, const int *__restrict__ box_max,
                                                       const int *__restrict__ blab, int nb1, int nb2,
                                                       const int *__restrict__ walk, int *fs, int *labels, int *first,
                                                       int *max_list, int max_cap, int *ovf_list, int ovf_cap, int maxsteps,
                                                       int has_vacuum, int CH, int xcd_split, int *__restrict__ bres = nullptr) {
    __shared__ int s_base, s_next;
    // CACHE + bres: did every voxel of the brick in work end on ONE maximum?  (the edge sweep's per-brick uniformity without a
    // pass over the labels.)  Per eighth the result of its lane 0, and a flag any lane raises whose result differs from it --
    // plain LDS stores and loads in the wave's program order: no atomics (1024 of them on one address per brick cost 0.2 ms
    // at 512^3), no cross-lane operations.
    __shared__ int s_w[8], s_mixed;
    constexpr bool CACHE = LEAN >= 3;
    int prev_brick = -1;
#ifdef XB_DEBUG_COUNT   // time probes (tools/trace_probe.py): where do the waves of the persistent trace spend their cycles?
    const long long pr_t0 = clock64();
    long long pr_walk = 0, pr_wait = 0, pr_load = 0;
    int pr_bricks = 0;
    const unsigned long long pr_e0 = wall_clock64();
#endif
    const int n_items = fs[FS_N_WALK] * 8;
    const int per = (((n_items + 7) >> 3) + 7) & ~7;   // whole bricks per XCD range
    const int home = xcd_split ? xcc_id() : (blockIdx.x & 7);
    const int lane = threadIdx.x & (XB_WAVE - 1);
    // the verdict of the brick this workgroup walked last (thread 0, after a barrier)
    auto verdict = [&]() {
        if (CACHE && bres && prev_brick >= 0) {
            int r0 = s_mixed ? -1 : s_w[0];
#pragma unroll
            for (int k = 1; k < 8; k++) r0 = (s_w[k] == r0) ? r0 : -1;
            bres[prev_brick] = r0 >= 0 ? r0 : (-2147483647 - 1);
        }
        prev_brick = -1;   // (written: a pull that finds its range empty must not write it again)
    };
    for (int r = 0; r < 8; r++) {
        const int q = (home + r) & 7;
        const int beg = q * per, end = min(beg + per, n_items);
        for (;;) {
#ifdef XB_DEBUG_COUNT
            const long long pr_a = clock64();
#endif
            __syncthreads();   // the previous brick's readers are done with s_base / s_next and the records
#ifdef XB_DEBUG_COUNT
            pr_wait += clock64() - pr_a;
#endif
            if (threadIdx.x == 0) {
                verdict();
                s_mixed = 0;
                s_base = beg + atomicAdd(&fs[FS_CURSOR0 + q * FS_CURSOR_STRIDE], CH);
                s_next = 0;
            }
            __syncthreads();
            const int base = s_base;
            if (base >= end) break;   // uniform over the workgroup
            const int stop = min(base + CH, end);
#ifdef XB_DEBUG_COUNT
            const long long pr_b = clock64();
            pr_bricks++;
#endif
            if (CACHE) {   // (CH == 8, base a multiple of 8: one brick) thread t copies the records of voxels t, t + 256 of the brick
                constexpr int PER = 512 / (XB_WAVE * XB_TRACE_WAVES);
                const int b = walk[base >> 3];
                prev_brick = b;
                const int b0 = b / (nb1 * nb2), b1 = (b / nb2) % nb1, b2 = b % nb2;
                GradRec tmp[PER];
#pragma unroll
                for (int k = 0; k < PER; k++) {
                    const int t = threadIdx.x + k * XB_WAVE * XB_TRACE_WAVES;
                    int cx = b0 * 8 + (t >> 6), cy = b1 * 8 + ((t >> 3) & 7), cz = b2 * 8 + (t & 7);
                    if (PART) { cx = min(cx, g.nx - 1); cy = min(cy, g.ny - 1); cz = min(cz, g.nz - 1); }   // (beyond the grid: any record, nobody reads the slot)
                    const int lt = lin24(g, cx, cy, cz);
                    tmp[k] = fetch_rec(G, WINDOW ? rec_slot(g, lt) : lt);
                }
#pragma unroll
                for (int k = 0; k < PER; k++) xb_s_rec[threadIdx.x + k * XB_WAVE * XB_TRACE_WAVES] = tmp[k];
                __syncthreads();
            }
#ifdef XB_DEBUG_COUNT
            pr_load += clock64() - pr_b;
#endif
            for (;;) {
                int i = 0;
                if (lane == 0) i = atomicAdd(&s_next, 1);
                const int item = base + __builtin_amdgcn_readfirstlane(i);
                if (item >= stop) break;
                int sx, sy, sz;
                brick_sub_voxel(walk[item >> 3], item & 7, lane, nb1, nb2, sx, sy, sz);
                // PART (a brick the grid cuts): a lane whose voxel lies beyond the grid walks from the nearest voxel of the brick
                // inside it -- the same trajectory, the same label and notes as that voxel's own lane leaves, so the duplicates
                // change nothing, and the brick's verdict below still speaks of its voxels inside the grid only
                if (PART) { sx = min(sx, g.nx - 1); sy = min(sy, g.ny - 1); sz = min(sz, g.nz - 1); }
                if (LEAN) {
#ifdef XB_DEBUG_COUNT
                    const long long pr_c = clock64();
#endif
                    const int res = ng_walk_lean<LEAN == 4, CACHE, WINDOW>(g, G, box_max, blab, nb1, nb2, sx, sy, sz, labels, first, max_list, &fs[FS_N_MAX],
                                                                max_cap, ovf_list, &fs[FS_N_OVF], ovf_cap, maxsteps, has_vacuum != 0);
#ifdef XB_DEBUG_COUNT
                    pr_walk += clock64() - pr_c;
#endif
                    if (CACHE && bres) {
                        volatile int *w = s_w;
                        if (lane == 0) w[item - base] = res;
                        if (w[item - base] != res) *(volatile int *)&s_mixed = 1;   // (same wave: the store above is older in its LDS queue)
                    }
                }
                else
                    ng_walk_wave<K, false>(g, G, box_max, blab, nb1, nb2, true, sx, sy, sz, labels, first, max_list, &fs[FS_N_MAX],
                                           max_cap, ovf_list, &fs[FS_N_OVF], ovf_cap, maxsteps, nullptr, nullptr, has_vacuum != 0);
            }
        }
    }
#ifdef XB_DEBUG_COUNT
    // PLAIN stores into a slot per wave (the host adds them up).  Round 5 learnt it the hard way: 50 K atomicAdds on one line as the
    // workgroups finish stretch the end of the kernel by 0.5 ms, and everything measured in that stretch is the probe itself.
    if (lane == 0 && blockIdx.x < 2048) {
        unsigned long long *o = xb_dbg + 64 + 6 * (blockIdx.x * XB_TRACE_WAVES + (threadIdx.x >> 6));
        o[0] = (unsigned long long)pr_walk; o[1] = (unsigned long long)(clock64() - pr_t0); o[2] = (unsigned long long)pr_wait;
        o[3] = (unsigned long long)pr_load; o[4] = pr_e0; o[5] = (unsigned long long)wall_clock64();
        if (threadIdx.x == 0) xb_dbg[64 + 6 * 8192 + blockIdx.x] = (unsigned long long)pr_bricks;
    }
#endif
    if (CACHE && bres) {   // the last brick this workgroup walked
        __syncthreads();
        if (threadIdx.x == 0) verdict();
    }
}

// The voxels of the walk-list bricks that still carry -2 ("handed to the exact slow kernel") -- for a density that hands over more
// walkers than the list holds (round 5: no hard cap any more; the host lists and runs them cap by cap).  A workgroup per brick.
__global__ __launch_bounds__(TPB) void k_list_unfinished(GridL g, const int *__restrict__ walk, const int *n_walk, int nb1, int nb2,
                                                         const int *__restrict__ labels, int *out, int *out_count, int out_cap) {
    __shared__ int s_buf[BlockAppender<2>::CAP], s_n[2];
    BlockAppender<2> app;
    app.init(s_buf, s_n, out, out_count, out_cap);
    const int n = *n_walk;
    for (int e = blockIdx.x; e < n; e += gridDim.x) {   // (uniform per block)
        const int b = walk[e];
        const int b0 = b / (nb1 * nb2), b1 = (b / nb2) % nb1, b2 = b % nb2;
        int v[2];
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int t = threadIdx.x + k * TPB;
            const int x = b0 * 8 + (t >> 6), y = b1 * 8 + ((t >> 3) & 7), z = b2 * 8 + (t & 7);
            if (x < g.nx && y < g.ny && z < g.nz) {
                const int l = (x * g.ny + y) * g.nz + z;
                if (labels[l] == -2) v[cnt++] = l;
            }
        }
        app.add(cnt, [&](int k) { return v[k]; });
    }
    app.finish();
}

// helpers of the remote path queries (slab scheduler)
// packed path of walker t: its start voxel, then the voxels from index first[t] on
__global__ void k_path_pack(const int *__restrict__ path, int lmax, const int *__restrict__ off, const int *__restrict__ len,
                            const int *__restrict__ first, int *__restrict__ packed) {
    const int t = blockIdx.x;
    const int *P = path + (size_t)t * lmax;
    if (threadIdx.x == 0) packed[off[t]] = P[0];
    for (int k = first[t] + threadIdx.x; k < len[t]; k += blockDim.x) packed[off[t] + 1 + k - first[t]] = P[k];
}
// labels / known of listed voxels out of the grid arrays (SCATTER false) or back into them (remote path queries)
template <bool SCATTER>
__global__ void k_move_voxels(const int *__restrict__ idx, int n, int *labels, int8_t *known, int *lab_buf, int8_t *kn_buf) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (SCATTER) { labels[idx[t]] = lab_buf[t]; known[idx[t]] = kn_buf[t]; }
    else { lab_buf[t] = labels[idx[t]]; kn_buf[t] = known[idx[t]]; }
}

// Exact slow path for the (rare) trajectories whose path membership could not be decided from the
// window: the whole path lives in global scratch and is scanned linearly.
// mode 0: assignment (write maximum index, note it); mode 1: refinement retrace; mode 2: only record the
// whole trajectory up to its maximum (path + length), for the slab scheduler's remote path queries.
// Path storage: voxel k of walker t at path[k * sk + t * st] -- (sk, st) = (1, lmax) one walker after the other (the path dumps of
// the slab scheduler read them like that), or (n, 1) interleaved: the lanes of a wave write neighbouring words (round 5: the
// exact path runs in TIERS -- every listed walker with 64 path voxels each in a few large launches, the few that need more with
// 2048 and then 32768 -- `retry`: where a walker whose path does not fit is listed instead of failing the call).
__global__ void k_trace_slow(Grid g, const double *__restrict__ rho, int *labels, const int8_t *known_ro,
                             int8_t *known, const int *list, int n, int *path, int lmax, int refine, int *first,
                             int *max_list, int *max_count, int max_cap, int *changed, int *escaped, int *err,
                             int *lens, int has_vacuum = 1, long long sk = 1, long long st = -1, int *retry = nullptr, int *retry_count = nullptr,
                             const int *n_dev = nullptr) {
    // (n_dev: the list's length lives on the device -- the launch is sized by the bound n, the path storage strides by it)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n || (n_dev && t >= *n_dev)) return;
    const int v = list[t];
    if (st < 0) st = lmax;
    int *P0 = path + (size_t)t * st;
#define P(k) P0[(size_t)(k) * sk]
    int np = 0;
    int px = v / g.nyz;
    int r = v - px * g.nyz;
    int py = r / g.nz, pz = r - py * g.nz, lp = v;
    double c = rho[v], dr0 = 0., dr1 = 0., dr2 = 0.;
    const int vol_num = labels[v];
    P(np++) = v;
    int result = -3;
    for (;;) {
        int qx, qy, qz;
        const bool stay = ng_step(rho, g, px, py, pz, lp, c, dr0, dr1, dr2, qx, qy, qz);
        int lq = lin3(g, qx, qy, qz);
        bool on_path = stay;
        for (int k = np - 1; k >= 0 && !on_path; k--) on_path = (P(k) == lq);
        if (on_path) {
            dr0 = dr1 = dr2 = 0.;
            og_step(rho, g, g.dist, px, py, pz, c, qx, qy, qz);
            lq = lin3(g, qx, qy, qz);
            if (qx == px && qy == py && qz == pz) { result = lp; break; }
        }
        if (refine == 1) {
            if (!plane_valid(g, qx)) { known[v] = -6; atomicAdd(escaped, 1); return; }
            if (known_ro[lq] == 2) { result = lq; break; }
        }
        if (np >= lmax) {
            if (refine == 2) {  // truncated dump: the scheduler asks again with a longer cap
                lens[t] = -np;
                int fo = np;
                for (int k = 1; k < np && fo == np; k++)
                    if (!plane_valid(g, P(k) / g.nyz)) fo = k;
                lens[n + t] = fo;
                return;
            }
            if (retry) retry[atomicAdd(retry_count, 1)] = v;   // (the list takes every walker of this launch)
            else atomicExch(err, 1);
            return;
        }
        P(np++) = lq;
        px = qx; py = qy; pz = qz; lp = lq; c = rho[lq];
    }
    if (refine == 2) {  // length, and where the path first leaves this rank's valid planes (the part before is known
        lens[t] = np;   // to hold no stop voxel: the fast retrace walked it)
        int fo = np;
        for (int k = 1; k < np && fo == np; k++)
            if (!plane_valid(g, P(k) / g.nyz)) fo = k;
        lens[n + t] = fo;
        return;
    }
#undef P
    if (refine) {
        const int nv = labels[result];
        if (nv != vol_num) { labels[v] = nv; known[v] = -2; atomicAdd(changed, 1); }
        else known[v] = -1;
    } else {
        // the vacuum rule (methods.py:449-452) -- only with vacuum: without it `labels` is write-only during an assignment
        // (a deferred labels := 0 was dropped, the regions' labels come later), and the label of the maximum's voxel may
        // be whatever an earlier owner of the memory left there
        if (has_vacuum && result != v && labels[result] == -1) result = -1;
        labels[v] = result;
        if (result >= 0) note_maximum(result, v, first, max_list, max_count, max_cap);
    }
}
