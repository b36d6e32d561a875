// k_fused.h -- device kernels of libbader_hip.so: the control flow of the single-GPU step kept ON the device.
// Included by bader_hip.hip (one translation unit).
//
// Round 1 drove the step from the host: ~25 times per step a counter was copied back and waited for (seed count,
// box radii, brick-growth convergence flags, list lengths, maxima table) -- ~30 us each, ~0.8 ms of a 5.6 ms step.
// Here every such decision is taken by a small kernel that leaves its result in the state block `fs` (device
// ints), the data-dependent launches are either fixed-size grids that stride over a device-side count or
// persistent grids pulling from a device-side cursor, and the host waits ONCE at the end of xb_assign and ONCE per
// refinement iteration.
#pragma once

// state block (ints, device).  Zeroed by one memset at the start of an assignment.
enum {
    FS_N_SEEDS = 0,      // 26-neighbour maxima appended by the table pass (raw count, may exceed the capacity)
    FS_TIES,             // != 0: some voxel's record depends on the tie rule
    FS_N_SEEDS_EFF,      // seeds used for boxes (0 when there are none or too many)
    FS_N_BOXES,          // trapping regions (seed bricks that may grow one)
    FS_GROW_PHASE,       // 0 propagate, 1 kill, 2 done
    FS_GROW_CHANGED,     // a block of the current launch changed a label
    FS_GROW_TICKET,      // blocks of the current launch that have finished
    FS_GROW_CUR,         // which of the two label buffers holds the current labels
    FS_GROW_CONVERGED,   // the kill iteration reached its fixpoint
    FS_N_CERTAIN,        // bricks inside trapping regions
    FS_N_WALK,           // bricks on the walk list
    FS_N_MAX,            // maxima noted by the trace
    FS_N_OVF,            // trajectories handed to the exact slow kernel
    FS_SORT_OK,          // the numbering was done on the device
    FS_ERR,              // loud failures (bit 0: ongrid chase did not terminate)
    FS_N_EDGES,          // refinement: edge list length
    FS_CHANGED, FS_ESCAPED, FS_R_OVF,
    FS_R_DEFER,          // refinement: retraces handed to the from-rho kernel (their walk goes on through a brick without records)
    FS_N_TILES,          // refinement: tiles of the edge sweep that are not of one label with their surroundings
    FS_N_CHGLIST,        // refinement: start voxels the retraces relabelled, as listed for the next edge_check
    FS_GROW_RETRY,       // the scheduled kill launches did not reach the fixpoint: the host repeats the assignment with the long schedule
    FS_N_RECL,           // slabs: bricks of the table window that get records (the walk list holds the owned ones among them)
    FS_N_REDO,           // slabs: trajectories that left the table window and were redone from rho (statistics)
    FS_COUNT = 64,
    // 8 per-XCD work cursors of the persistent trace, one per 128-byte line: device-scope atomics on ONE line
    // serialise at ~88 per microsecond whatever the word (measured: 8 cursors in one line = one cursor)
    FS_CURSOR0 = 64, FS_CURSOR_STRIDE = 32,
    FS_TOTAL = FS_CURSOR0 + 8 * FS_CURSOR_STRIDE
};

#define XB_REGIONS_MAX 65535   // trapping regions seeded by bricks (k_seed_bricks)

#define XB_BOXES_MAX 1024   // regions whose rank sits in the LDS table of the relabel kernels (the others are looked up)
// Growing the trapping regions brick by brick (8x8x8 voxels).  Let U be a union of sets certain for maximum m.  A brick B
// without a 26-neighbour maximum whose every possible move (any dr) from every voxel lands in B itself or in bricks that are
// certain for the SAME m keeps U + B closed, and a trajectory cannot stay in B forever (it only ends on a maximum), so it must
// enter U: B is certain for m as well.  Mutually dependent bricks are certified together by a greatest-fixpoint (kill)
// iteration on provisional labels: a workgroup stages an 8^3 chunk of brick labels + a one-brick periodic halo in LDS and
// iterates on it until nothing changes or `inner` rounds are done; the halo is what the previous launch left.  The schedule
// does not matter for soundness: a provisional label is only a guess, and the kill iteration is monotone (stale neighbour
// labels can only delay a kill), so its fixpoint -- reached when a whole launch changes nothing -- is the same greatest
// fixpoint.  Convergence logic on the device: every launch of the fixed schedule reads the phase (0 propagate, 1 kill, 2 done:
// return at once); the block that finishes last advances the phase when the launch changed nothing and flips the buffer.
__global__ __launch_bounds__(BG * BG * BG) void k_brick_grow_dev(int nb0, int nb1, int nb2, const int *__restrict__ bmask,
                                                                 const int *__restrict__ seed, int *buf0, int *buf1, int *fs,
                                                                 int inner, int seeds_fixed) {
    __shared__ int lab[2][BG + 2][BG + 2][BG + 2];
    __shared__ int s_any;
    const int phase = fs[FS_GROW_PHASE];
    if (phase >= 2) return;
    const int curbuf = fs[FS_GROW_CUR];
    const int *in = curbuf ? buf1 : buf0;
    int *out = curbuf ? buf0 : buf1;
    if (threadIdx.x == 0) s_any = 0;
    const int c0 = blockIdx.z * BG, c1 = blockIdx.y * BG, c2 = blockIdx.x * BG;
    for (int i = threadIdx.x; i < (BG + 2) * (BG + 2) * (BG + 2); i += BG * BG * BG) {
        const int e2 = i % (BG + 2), e1 = (i / (BG + 2)) % (BG + 2), e0 = i / ((BG + 2) * (BG + 2));
        const int l = in[(wrap_any(c0 + e0 - 1, nb0) * nb1 + wrap_any(c1 + e1 - 1, nb1)) * nb2 + wrap_any(c2 + e2 - 1, nb2)];
        lab[0][e0][e1][e2] = l;
        lab[1][e0][e1][e2] = l;
    }
    const int t2 = threadIdx.x % BG, t1 = (threadIdx.x / BG) % BG, t0 = threadIdx.x / (BG * BG);
    const int b0 = c0 + t0, b1 = c1 + t1, b2 = c2 + t2;
    const bool active = b0 < nb0 && b1 < nb1 && b2 < nb2;
    const int b = active ? (b0 * nb1 + b1) * nb2 + b2 : 0;
    const int m = active ? bmask[b] : 0;
    // seeds_fixed: the seeds are closed cubes, trapping regions on their own (they never die).  Otherwise the seeds are
    // the bricks that hold exactly one maximum (k_seed_bricks): such a brick may keep ITS OWN label through the kill
    // iteration, but only under the same closure rule as every other brick
    const int sd = active ? seed[b] : 0;
    const bool fixed = !active || (phase == 0 && (m >> 27)) || (phase == 1 && seeds_fixed && sd != 0);
    const bool own_seed_ok = !seeds_fixed && sd != 0 && !((m >> 28) & 1);
    __syncthreads();
    const int first = lab[0][t0 + 1][t1 + 1][t2 + 1];
    int l = first, cur = 0;
    for (int it = 0; it < inner; it++) {
        int nl = l;
        if (!fixed) {
            if (phase == 0) {
                if (l == 0) {
                    int best = 0;
                    for (unsigned mm = (unsigned)m & 0x7ffffffu; mm; mm &= mm - 1) {   // the bricks it can move into
                        const int k = __ffs(mm) - 1;
                        const int q = lab[cur][t0 + k / 9][t1 + (k / 3) % 3][t2 + k % 3];
                        if (q > 0 && (best == 0 || q < best)) best = q;
                    }
                    nl = best;
                }
            } else if (l > 0) {
                bool ok = !(m >> 27) || (own_seed_ok && sd == l);
                for (unsigned mm = (unsigned)m & 0x7ffffffu; mm && ok; mm &= mm - 1) {
                    const int k = __ffs(mm) - 1;
                    ok = (lab[cur][t0 + k / 9][t1 + (k / 3) % 3][t2 + k % 3] == l);
                }
                if (!ok) nl = 0;
            }
        }
        lab[cur ^ 1][t0 + 1][t1 + 1][t2 + 1] = nl;
        const int any = __syncthreads_or(nl != l);
        l = nl;
        cur ^= 1;
        if (!any) break;
    }
    if (active) out[b] = l;
    if (active && l != first) s_any = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        // ONE atomic carries this block's ticket (low 16 bits) and its change flag (bit 16 up): the block whose add
        // comes last sees every other block's flag in the returned word -- no fence needed, the labels themselves
        // only have to be visible to the NEXT launch
        const int nblk = gridDim.x * gridDim.y * gridDim.z;
        const int old = atomicAdd(&fs[FS_GROW_TICKET], 1 + (s_any ? 0x10000 : 0));
        if ((old & 0xffff) == nblk - 1) {   // last block of this launch
            const bool ch = (old >> 16) != 0 || s_any;
            fs[FS_GROW_TICKET] = 0;
            fs[FS_GROW_CUR] = curbuf ^ 1;       // `out` holds the labels now (identical to `in` if nothing changed)
            if (!ch) {
                fs[FS_GROW_PHASE] = phase + 1;
                if (phase == 1) fs[FS_GROW_CONVERGED] = 1;
            }
        }
    }
}
// Provisional labels of the region growth WITHOUT rounds (round 3; replaces phase 0 of k_brick_grow_dev on the one-GPU
// path: ~6 dependent launches of labels crawling 8 bricks each).  Any guess is sound -- the kill iteration decides what
// survives -- so every brick simply points at ONE brick it can move into, the one with the largest density maximum
// (`bpot`, from pass A) provided that is larger than its own: potentials rise strictly along a chain, so chains cannot
// cycle, and they end where no successor is higher -- in the brick that holds the basin's maximum (a seed: label = its
// region id) or in a dead end (a brick with several maxima, or none higher around: label 0).  k_grow_chase follows the
// chain of every brick at once; the parents are read-only, so no synchronisation is needed.
__global__ __launch_bounds__(TPB) void k_grow_parent(int nb0, int nb1, int nb2, const int *__restrict__ bmask, const int *__restrict__ bpot,
                                                     const int *__restrict__ seed, int *__restrict__ parent) {
    const int b = blockIdx.x * TPB + threadIdx.x;
    if (b >= nb0 * nb1 * nb2) return;
    const int m = bmask[b];
    if (seed[b] != 0) { parent[b] = b; return; }            // exactly one maximum: the root of its region
    if ((m >> 27) & 3) { parent[b] = -1; return; }          // several maxima (or a seed beyond the table): never certified
    const int b0 = b / (nb1 * nb2), b1 = (b / nb2) % nb1, b2 = b % nb2;
    // Round 4: the order along a chain is (potential, is a seed, brick index), lexicographic -- a strict total order, so chains
    // still cannot cycle, but a TIE of the single-precision potentials no longer ends one.  It matters exactly where it hurts
    // most: a maximum that lies on a face of its brick has a neighbour voxel of almost its density in the next brick, the two
    // bricks get the same potential, the neighbour found no higher successor, kept label 0, and the seed brick -- which can
    // move into it -- died in the kill iteration with its whole region (a third of the voxels of a 500^3 grid walked instead
    // of certified; the brick-aligned 512^3 of the headline happens to have no maximum on a brick face).
    int best = -1, bp = bpot[b], bs = 0, bq = b;
    for (unsigned mm = (unsigned)m & 0x7ffffffu; mm; mm &= mm - 1) {   // the bricks it can move into
        const int k = __ffs(mm) - 1;
        const int q = (wrap_any(b0 + k / 9 - 1, nb0) * nb1 + wrap_any(b1 + (k / 3) % 3 - 1, nb1)) * nb2 + wrap_any(b2 + k % 3 - 1, nb2);
        if (q == b) continue;   // (a lattice two bricks wide wraps onto the brick itself)
        const int pq = bpot[q], sq = seed[q] != 0 ? 1 : 0;
        if (pq > bp || (pq == bp && (sq > bs || (sq == bs && q > bq)))) { bp = pq; bs = sq; bq = q; best = q; }
    }
    parent[b] = best;
}
// (fs: thread 0 also sets the growth's state for the kill iteration that follows -- the chase has made the provisional labels;
// a launch of its own for these five words was 5 us of the step)
__global__ __launch_bounds__(TPB) void k_grow_chase(int nbr, const int *__restrict__ parent, const int *__restrict__ seed, int *__restrict__ lab,
                                                    int limit, int *fs) {
    const int b = blockIdx.x * TPB + threadIdx.x;
    if (b == 0) {
        const int n = min(fs[FS_N_SEEDS], XB_REGIONS_MAX);
        fs[FS_N_SEEDS_EFF] = n;
        fs[FS_N_BOXES] = n;
        fs[FS_GROW_PHASE] = n ? 1 : 2;   // nothing to grow without a seed
        fs[FS_GROW_CONVERGED] = 0;
        fs[FS_GROW_CUR] = 0;
    }
    if (b >= nbr) return;
    int p = b, l = 0;
    for (int it = 0; it < limit; it++) {
        const int pp = parent[p];
        if (pp < 0) break;
        if (pp == p) { l = seed[p]; break; }
        p = pp;
    }
    lab[b] = l;
}
// blab := the surviving labels (a FIXPOINT of the kill iteration is closed under every move; without one fall
// back to the seed cubes, which are trapping regions on their own); counts the certain bricks
// box_first[id] (preset to INT_MAX): the smallest certain brick index of box id + 1 -- without vacuum every voxel of
// a certain brick belongs to the region's maximum and the smallest voxel index of a brick is its corner, so the
// numbering needs one note per region, not one per brick.
__global__ __launch_bounds__(TPB) void k_grow_finish(int nbr, const int *__restrict__ seed, const int *__restrict__ buf0,
                                                     const int *__restrict__ buf1, int *fs, int *__restrict__ blab,
                                                     int *box_first, const int *__restrict__ bmask, unsigned char *brick_rec,
                                                     int seeds_fixed, int verdict) {
    __shared__ int s_first[XB_BOXES_MAX];
    // verdict: the scheduled kill launches are over -- did they reach the fixpoint?  If not, everything downstream is skipped
    // (FS_GROW_RETRY) and the host repeats the assignment with the long schedule.  Every block sees the same state.
    const bool retry = verdict ? (fs[FS_GROW_PHASE] == 1 && !fs[FS_GROW_CONVERGED]) : fs[FS_GROW_RETRY] != 0;
    if (verdict && blockIdx.x == 0 && threadIdx.x == 0) fs[FS_GROW_RETRY] = retry ? 1 : 0;
    if (retry) return;
    // without a fixpoint only closed cubes are regions on their own; seed BRICKS alone certify nothing
    const int *src = fs[FS_GROW_CONVERGED] ? (fs[FS_GROW_CUR] ? buf1 : buf0) : seed;
    const int nbx = (fs[FS_GROW_CONVERGED] || seeds_fixed) ? fs[FS_N_BOXES] : 0;
    for (int i = threadIdx.x; i < XB_BOXES_MAX; i += TPB) s_first[i] = XB_INT_MAX;
    __syncthreads();
    int cnt = 0;
    for (int b = blockIdx.x * TPB + threadIdx.x; b < nbr; b += gridDim.x * TPB) {
        const int l = nbx ? src[b] : 0;
        blab[b] = l;
        if (brick_rec) brick_rec[b] = (unsigned char)(((bmask[b] >> 27) & 1) << 1);   // no records yet; bit 1: holds a maximum
        cnt += (l > 0);
        if (l > 0 && l <= XB_BOXES_MAX) { if (s_first[l - 1] > b) atomicMin(&s_first[l - 1], b); }
        else if (l > XB_BOXES_MAX) atomicMin(&box_first[l - 1], b);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < XB_BOXES_MAX; i += TPB)
        if (s_first[i] != XB_INT_MAX) atomicMin(&box_first[i], s_first[i]);
    int total;
    block_scan_excl(cnt, total);
    if (threadIdx.x == 0 && total) atomicAdd(&fs[FS_N_CERTAIN], total);
}
// one note per region (replaces a note per certain brick)
__global__ __launch_bounds__(XB_BOXES_MAX) void k_note_regions(GridL g, int nb1, int nb2, const int *__restrict__ fs, const int *__restrict__ box_first,
                                                     const int *__restrict__ box_max, int *first, int *max_list, int *max_count,
                                                     int max_cap) {
    if (fs[FS_GROW_RETRY]) return;
    for (int t = threadIdx.x; t < fs[FS_N_BOXES]; t += blockDim.x) {
        if (box_first[t] == XB_INT_MAX) continue;
        const int b = box_first[t];
        const int b0 = b / (nb1 * nb2), b1 = (b / nb2) % nb1, b2 = b % nb2;
        note_maximum(box_max[t], ((b0 * 8) * g.ny + b1 * 8) * g.nz + b2 * 8, first, max_list, max_count, max_cap);
    }
}
// Seeds of the brick growth WITHOUT cubes (the route of k_brick_masks): a brick that holds exactly one 26-neighbour
// maximum starts with that maximum's region id.  No cap but the table size: a density with thousands of maxima (noise in
// the vacuum of a real CHGCAR) still gets the regions of its atoms, and a brick with several maxima is simply never
// certified.  Ids are handed out in arrival order (the basin numbering is decided later, by first voxel).
// (round 5: the launch also presets box_first, the regions' smallest certain brick, for k_grow_finish -- a fill of its own before)
__global__ void k_seed_bricks(int nbr, const int *__restrict__ bmask, const int *__restrict__ bmaxv, int *fs, int *seed, int *buf0,
                              int *box_max, int *box_first) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = b; i < XB_REGIONS_MAX; i += gridDim.x * blockDim.x) box_first[i] = XB_INT_MAX;
    if (b >= nbr) return;
    int id = 0;
    if (((bmask[b] >> 27) & 3) == 1) {
        const int k = atomicAdd(&fs[FS_N_SEEDS], 1);
        if (k < XB_REGIONS_MAX) { id = k + 1; box_max[k] = bmaxv[b]; }
    }
    seed[b] = id;
    buf0[b] = id;
}
__global__ void k_seed_finish(int *fs) {
    const int n = min(fs[FS_N_SEEDS], XB_REGIONS_MAX);
    fs[FS_N_SEEDS_EFF] = n;
    fs[FS_N_BOXES] = n;
    fs[FS_GROW_PHASE] = n ? 0 : 2;   // nothing to grow without a seed
    fs[FS_GROW_CONVERGED] = 0;
}
// numbering on the device: maxima sorted by the smallest voxel index that reaches them (thread_handlers.py:59-65
// numbers maxima in scan order); first[m] := rank.  One block, bitonic sort in LDS; more than XB_SORT_MAX maxima
// (noisy data) leave FS_SORT_OK = 0 and the host sorts instead.
#define XB_SORT_MAX 2048
__global__ __launch_bounds__(1024) void k_number_maxima(int *fs, int *first, const int *__restrict__ max_list, int max_cap,
                                                        int *sorted, int *sorted2 = nullptr) {
    __shared__ unsigned long long key[XB_SORT_MAX];
    const int n = fs[FS_N_MAX];
    // trajectories waiting for the exact slow kernel may still discover maxima: number on the host afterwards
    if (n > XB_SORT_MAX || n > max_cap || fs[FS_N_OVF] > 0 || fs[FS_GROW_RETRY]) { if (threadIdx.x == 0) fs[FS_SORT_OK] = 0; return; }
    int len = 64;   // sort the next power of two >= n (a handful of maxima is the common case)
    while (len < n) len <<= 1;
    for (int i = threadIdx.x; i < len; i += 1024)
        key[i] = i < n ? ((unsigned long long)(unsigned)first[max_list[i]] << 32) | (unsigned)max_list[i] : ~0ull;
    __syncthreads();
    for (int k = 2; k <= len; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < len; i += 1024) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long a = key[i], b = key[p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { key[i] = b; key[p] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += 1024) {
        const int m = (int)(key[i] & 0xffffffffu);
        sorted[i] = m;
        if (sorted2) sorted2[i] = m;   // (a copy next to the state block: the host fetches both with one transfer)
        first[m] = i;
    }
    if (threadIdx.x == 0) fs[FS_SORT_OK] = 1;
}

// ---------------------------------------------------------------------------------------------------------------
// Numbering of ANY number of maxima on the device (round 6; k_number_maxima sorts up to XB_SORT_MAX of them in LDS, and a
// density whose noise makes millions -- 1.6 M in the noisy vacuum of bench.py's user leg -- went to the host for a counting
// sort: fetch the table, sort, send the ranks back, 15 ms).  No sort is needed: the rank of a maximum is the number of maxima
// whose FIRST voxel (the smallest voxel index that reaches them) is smaller, and first voxels are distinct -- a voxel reaches
// one maximum.  So: mark the first voxels in a bitmap over the grid, prefix-count the bitmap (a popcount per word, a scan per
// 1024 words, a scan of the block sums) and every maximum reads its rank off its own bit.  N / 4 bytes of scratch.
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_rank_mark(const int *__restrict__ first, const int *__restrict__ max_list, int n, unsigned *bits) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned v = (unsigned)first[max_list[i]];
    atomicOr(&bits[v >> 5], 1u << (v & 31));
}
// exclusive scan of `x` over the 1024 threads of a workgroup; `total`: the sum
__device__ __forceinline__ int scan1024_excl(int x, int &total) {
    __shared__ int wsum[1024 / XB_WAVE];
    const int lane = threadIdx.x % XB_WAVE, w = threadIdx.x / XB_WAVE;
    int incl = x;
#pragma unroll
    for (int o = 1; o < XB_WAVE; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == XB_WAVE - 1) wsum[w] = incl;
    __syncthreads();
    int base = 0;
    total = 0;
#pragma unroll
    for (int q = 0; q < 1024 / XB_WAVE; q++) {
        if (q < w) base += wsum[q];
        total += wsum[q];
    }
    __syncthreads();
    return base + incl - x;
}
// per workgroup 1024 words of the bitmap: wprefix[w] = set bits in the words of the block before w, bsum[block] = its set bits
__global__ __launch_bounds__(1024) void k_rank_scan(const unsigned *__restrict__ bits, int n_words, int *__restrict__ wprefix, int *__restrict__ bsum) {
    const int w = blockIdx.x * 1024 + threadIdx.x;
    const int cnt = w < n_words ? __popc(bits[w]) : 0;
    int total;
    const int excl = scan1024_excl(cnt, total);
    if (w < n_words) wprefix[w] = excl;
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}
// exclusive scan of the block sums in place (one workgroup, 1024 at a time with a carry)
__global__ __launch_bounds__(1024) void k_rank_blocks(int *bsum, int n_blocks) {
    int carry = 0;
    for (int base = 0; base < n_blocks; base += 1024) {
        const int i = base + threadIdx.x;
        const int x = i < n_blocks ? bsum[i] : 0;
        int total;
        const int excl = scan1024_excl(x, total);
        if (i < n_blocks) bsum[i] = carry + excl;
        carry += total;
    }
}
// first[m] := rank of m; sorted[rank] := m; the numbering is on the device now (FS_SORT_OK: the relabel kernels' gate)
__global__ void k_rank_assign(int *first, const int *__restrict__ max_list, int n, const unsigned *__restrict__ bits,
                              const int *__restrict__ wprefix, const int *__restrict__ bsum, int *__restrict__ sorted, int *fs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) fs[FS_SORT_OK] = 1;
    if (i >= n) return;
    const int m = max_list[i];
    const unsigned v = (unsigned)first[m];
    const unsigned w = v >> 5;
    const int r = bsum[w >> 10] + wprefix[w] + __popc(bits[w] & ((1u << (v & 31)) - 1u));
    sorted[r] = m;
    first[m] = r;
}
