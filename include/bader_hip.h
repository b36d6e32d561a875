/*
 * bader_hip.h -- C ABI of libbader_hip.so, the MI355X (gfx950) replacement for the hot path of
 * pybader v0.3.12: neargrid / ongrid steepest-ascent voxel->maximum assignment and the
 * edge-refinement sweep (pybader/methods.py, refinement.py, thread_handlers.py, the njit half of
 * utils.py).  Plain C types only; every function returns 0 on success or a negative XB_E_* code
 * with a message available from xb_last_error().  No C++ exception crosses this boundary.
 *
 * Data layout (the reference's, io/vasp.py:102-103): every grid array is C-order [x][y][z],
 * z fastest.  Density is float64; working labels ("volumes") are int32 on the device; `known`
 * edge flags are int8.  Host buffers are caller-owned numpy arrays; device buffers belong to the
 * context.  Label sentinels: -1 vacuum (utils.py:396-397), >=0 basin number after assignment.
 *
 * Each entry point cites the reference interface it replaces (file:line into pybader/).
 */
#ifndef BADER_HIP_H
#define BADER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xb_ctx xb_ctx;

enum { XB_OK = 0, XB_E_ARG = -1, XB_E_HIP = -2, XB_E_STATE = -3, XB_E_LIMIT = -4, XB_E_COMM = -5,
       XB_E_SHORT = -6 /* xb_parse_density_text: fewer numbers in the text than voxels in the grid */ };

/* label dtype codes accepted at the boundary: the reference narrows/widens labels between
 * int8/16/32/64 (utils.py:15-37 dtype_calc, jits.py:22-38 dtype matrix) */
enum { XB_I8 = 1, XB_I16 = 2, XB_I32 = 4, XB_I64 = 8 };

/* methods.__contains__ (methods.py:12) */
enum { XB_METHOD_ONGRID = 0, XB_METHOD_NEARGRID = 1 };
/* refine_mode[0] (thread_handlers.py:144, 201-205) */
enum { XB_REFINE_ALL = 0, XB_REFINE_CHANGED = 1 };

/* ---- library / context ------------------------------------------------------------------ */
const char *xb_last_error(void);
int xb_device_count(void);                       /* number of visible HIP devices (0 = none) */
int xb_create(int device, xb_ctx **out);         /* one context per GPU; owns a stream + buffers */
void xb_destroy(xb_ctx *c);
int xb_sync(xb_ctx *c);                          /* hipStreamSynchronize on the context's stream */
void *xb_stream(xb_ctx *c);                      /* the hipStream_t every kernel is launched on */

/* ---- grid residency ---------------------------------------------------------------------- */
/* Declares the grid and the two small matrices the reference computes on the host with numpy and
 * passes to every kernel (Bader.distance_matrix interface.py:242-259, Bader.T_grad 285-290):
 * dist_mat[27] row-major [3][3][3] with index 2 == -1; T_grad[9] row-major.  Allocates/reuses
 * device buffers.  x-slab [x0,x1) is the range of axis-0 planes this context owns (multi-GPU
 * slab scheduler; x0=0,x1=nx for one GPU).  Every rank holds the full density.
 * SIZE LIMIT (the reference indexes with int64 throughout, methods.py / refinement.py): voxel indices are int32 on the
 * device, so a grid needs nx*ny*nz < 2^31 - 1 voxels (1024^3 = 2^30 fits; 1290^3 is the largest cube) -- more returns
 * XB_E_LIMIT before anything is allocated.  Every entry point works up to that size. */
int xb_set_grid(xb_ctx *c, const int64_t shape[3], const double dist_mat[27], const double T_grad[9],
                int64_t x0, int64_t x1);
int xb_upload_density(xb_ctx *c, const double *rho_host);           /* H2D, nx*ny*nz float64 */
/* workload generator, bit-identical to pybader_amd/synth.py (bench + tests; not in the reference) */
int xb_synth_density(xb_ctx *c, const double lattice[9], const double *atoms5, int64_t n_atoms,
                     double background);
int xb_download_density(xb_ctx *c, double *rho_host);
/* The density block of a VASP CHGCAR / CHG file (io/vasp.py:90-104, 147-149): `text` holds nx*ny*nz (or more)
 * whitespace separated decimal numbers in Fortran order (x fastest); they are converted exactly as numpy's
 * string -> float64 does (correctly rounded), divided by `divisor` (the cell volume) and stored as the
 * resident density rho[x][y][z].  The text is uploaded as it is and parsed on the device; tokens outside the
 * exact fast path go through strtod on the host (n_host of them).  SURVEY.md 8(f) rank 4. */
int xb_parse_density_text(xb_ctx *c, const char *text, int64_t nbytes, double divisor, int64_t *n_tokens,
                          int64_t *n_host);
/* labels: host <-> device with widening/narrowing on the device (utils.dtype_change, utils.py:255-259) */
int xb_upload_labels(xb_ctx *c, const void *labels_host, int dtype);
int xb_download_labels(xb_ctx *c, void *labels_host, int dtype);
int xb_upload_known(xb_ctx *c, const int8_t *known_host);
int xb_download_known(xb_ctx *c, int8_t *known_host);

/* ---- hot path, device resident ----------------------------------------------------------- */
/* utils.vacuum_assign (utils.py:382-401) via Bader.volumes_init (interface.py:449-469):
 * labels := 0, then -1 where rho <= vac_tol (vac_tol NaN => no vacuum, the vacuum_tol=None case).
 * Returns the vacuum charge (sum(rho)*voxel_volume) and volume. */
int xb_vacuum_assign(xb_ctx *c, double vac_tol, double voxel_volume, double *vac_charge, double *vac_volume);

/* thread_handlers.bader_calc (thread_handlers.py:15-75) with methods.neargrid (methods.py:222-611)
 * or methods.ongrid (methods.py:15-219): labels (0 / -1 on entry) become 0-based basin numbers,
 * numbered by the smallest C-order voxel index of each basin (the order the reference's scan
 * discovers maxima).  neargrid computes every voxel's own dr=0 trajectory -- the order-independent
 * map the reference's refinement converges to.
 * n_maxima: number of basins.  With several slabs the numbering is completed by
 * xb_assign_finish after the per-rank maxima tables were merged. */
int xb_assign(xb_ctx *c, int method, int64_t *n_maxima);
/* bader_max of thread_handlers.py:75: voxel indices int64[n][3] in label order */
int xb_get_maxima(xb_ctx *c, int64_t *maxima_out, int64_t capacity);

/* slab scheduler pieces of xb_assign (multi-GPU): phase 1 traces the owned slab and returns the
 * local table (maximum voxel index, smallest owned voxel index reaching it); the host merges the
 * tables of all ranks (min over ranks), sorts, and hands the global table back to phase 2. */
int xb_assign_trace(xb_ctx *c, int method, int64_t *n_local);
int xb_assign_local_table(xb_ctx *c, int64_t *max_idx, int64_t *first_idx, int64_t capacity);
int xb_assign_finish(xb_ctx *c, const int64_t *max_idx_sorted, int64_t n_global);

/* optional, before the first xb_edge_find of a refinement: build what xb_refine_trace needs (the gradient-field
 * table of the resident density) now, so that edge_find can read "not a maximum" off it (xb_refine does it) */
int xb_prepare_refine(xb_ctx *c);
/* refinement.edge_find (refinement.py:326-405) on a fresh `known`: -2 edge, -1 within the 27-box
 * of an edge, 2 other non-vacuum, 0 untouched vacuum.  Returns the edge count of the owned slab. */
int xb_edge_find(xb_ctx *c, int64_t *edges);
/* refinement.neargrid (refinement.py:17-322): retrace every known==-2 voxel of the owned slab
 * until a known==2 voxel or a maximum; relabel the start voxel if the label differs.
 * escaped: traces that left the valid x-range [x0-halo, x1+halo) (slabs only; 0 on one GPU). */
int xb_refine_trace(xb_ctx *c, int64_t *changed, int64_t *escaped);
/* slab fallback: escaped traces are parked as known == -6; after the scheduler made the whole grid
 * valid on this rank (all-gather of labels + known, xb_set_halo(nx)) this call retraces exactly them. */
int xb_refine_trace_escaped(xb_ctx *c, int64_t *changed, int64_t *escaped);
/* slab scheduler, escaped retraces without bulk traffic: the dr=0 trajectory of every parked voxel
 * (known == -6) of the owned slab -- it depends on the replicated rho only -- from the point where it first
 * leaves this rank's valid planes (the fast retrace already walked the part before without finding a stop).
 * The scheduler asks the owners of the path voxels for (label, known) with xb_gather_voxels, finds the first
 * known == 2 voxel (refinement.py:294-303) or the maximum, and writes the outcome back with
 * xb_scatter_voxels.  Layout: offsets has n_paths + 1 entries; voxels[offsets[i]] is the start voxel of path
 * i, the rest of path i follows (linear indices).  Trajectories are followed for at most max_len voxels;
 * complete[i] = 1 when path i reached its maximum, else ask again with a larger max_len. */
int xb_escaped_paths(xb_ctx *c, int64_t max_len, int64_t *n_paths, int64_t *n_voxels);
int xb_escaped_paths_fetch(xb_ctx *c, int64_t *starts, int64_t *offsets, int64_t *voxels, int8_t *complete);
int xb_gather_voxels(xb_ctx *c, const int64_t *idx, int64_t n, int32_t *labels_out, int8_t *known_out);
int xb_scatter_voxels(xb_ctx *c, const int64_t *idx, int64_t n, const int32_t *labels_in, const int8_t *known_in);
/* slab scheduler, escaped retraces carried on by their next owner.  A retrace that leaves this rank's valid planes is
 * parked (known == -6) and exported as a walker: XB_WALKER_WORDS int64 holding its start voxel and label, the voxel it
 * arrived at, the carried remainder, the path window and the step count (refinement.py:137-154, 200-235: everything
 * the loop carries).  The scheduler all-gathers the walkers; xb_walkers_continue carries on the ones that arrived on a
 * plane this rank owns (its labels / known are the authoritative ones there: a retrace only reads labels at known == 2
 * voxels and maxima, which no retrace rewrites) and yields results -- int64 pairs (start voxel | final label << 32) --
 * and the walkers that left its valid planes again; xb_walkers_apply applies the pairs whose voxel this rank owns
 * exactly as the retrace would have (refinement.py:288-291).  A walker that needs the exact slow path (path window
 * overflow) comes back as `stuck` and stays parked for xb_escaped_paths.  The reference has no counterpart: its
 * thread blocks read one shared array (thread_handlers.py:128-236). */
#define XB_WALKER_WORDS 10
int xb_walkers_count(xb_ctx *c, int64_t *n_walkers, int64_t *n_results);
int xb_walkers_fetch(xb_ctx *c, int64_t *walkers, int64_t *results);
int xb_walkers_continue(xb_ctx *c, const int64_t *walkers, int64_t n);
int xb_walkers_apply(xb_ctx *c, const int64_t *results, int64_t n, int64_t *changed, int64_t *stuck);
/* refinement.edge_check (refinement.py:409-508), bug-compatible (no vacuum test on the box voxels).
 * No size limit of its own below xb_set_grid's (round 6: the two flag bits of a queue entry moved out of the voxel's 32-bit
 * word; rounds 1-5 stopped at 2^30 voxels, exactly 1024^3). */
int xb_edge_check(xb_ctx *c, int64_t *checked, int64_t *edges);
/* refinement.edge_check across slabs ('changed' refinement on N GPUs): the greedy scan of refinement.py:420-427 is
 * global, so every rank resolves the global list of changed voxels.  _local: the owned changed voxels and their
 * edge&maximum class; _local_fetch copies them out; _global takes the all-gathered lists (label and known halos
 * refreshed before), resolves them and re-classifies the boxes that touch this rank's planes; `edges` counts the new
 * edges in the owned planes (the scheduler sums over ranks). */
int xb_edge_check_local(xb_ctx *c, int64_t *n_out);
int xb_edge_check_local_fetch(xb_ctx *c, int64_t *idx_out, int8_t *cls_out);
int xb_edge_check_global(xb_ctx *c, const int64_t *idx, const int8_t *cls, int64_t n, int64_t *checked, int64_t *edges);
/* thread_handlers.refine (thread_handlers.py:128-236): the iteration driver on one GPU.
 * iters < 0 => until nothing changes.  log[2*k] = edges, log[2*k+1] = changed of iteration k+1. */
int xb_refine(xb_ctx *c, int mode, int64_t iters, int64_t *log, int64_t log_capacity, int64_t *n_iters);
/* Bader.bader_calc + Bader.refine_volumes back to back, as Bader.__call__ issues them (interface.py:406-416, 471-490), in ONE call:
 * the refinement's first iteration is queued behind the assignment and one host wait serves both.  Results, maxima (xb_get_maxima)
 * and log equal xb_assign followed by xb_refine; combinations other than the one-GPU neargrid path without vacuum, and assignments
 * that do not end the usual way (tie voxels, walkers for the exact slow path), run as those two calls. */
int xb_assign_refine(xb_ctx *c, int method, int mode, int64_t iters, int64_t *n_maxima, int64_t *log, int64_t log_capacity, int64_t *n_iters);

/* utils.charge_sum (utils.py:235-252) via Bader.sum_volumes (interface.py:492-525) */
int xb_charge_sum(xb_ctx *c, double voxel_volume, int64_t n_labels, double *charge, double *volume);
/* utils.volume_assign (utils.py:404-421): labels[v] = swap[labels[v]] for labels >= 0 */
int xb_volume_assign(xb_ctx *c, const int64_t *swap, int64_t n_swap);
/* utils.atom_assign (utils.py:185-232): nearest atom of every maximum over the 27 periodic images (one
 * device thread per maximum; context free: buffers on the current device) */
int xb_atom_assign(const double *bader_max_cart, int64_t n_max, const double *atoms_cart, int64_t n_atoms,
                   const double lattice[9], int64_t *atom_out, double *dist_out);

/* thread_handlers.surface_distance (thread_handlers.py:239-297) + utils.surface_dist (utils.py:320-379)
 * on the resident atom map (labels = atoms_volumes): runs edge_find, then returns per atom the minimum
 * SQUARED distance of its edge voxels to the atom over the 27 periodic images (+inf: no edge voxel). */
int xb_surface_distance(xb_ctx *c, const double lattice[9], const double *atoms_cart, int64_t n_atoms,
                        double *min_d2, int64_t *edges);
/* utils.volume_mask (utils.py:461-476): density where labels == vol_num, else 0 (host float64 N) */
int xb_volume_mask(xb_ctx *c, int64_t vol_num, double *out_host);
/* sum of the resident density / count over owned voxels with labels == value (vacuum sums when the
 * reference density differs from the charge density, utils.py:396-400) */
int xb_label_sum(xb_ctx *c, int64_t value, double *sum, int64_t *count);

/* ---- windowed table build (multi-GPU): each rank builds the 32-B/voxel gradient-field table only for
 * its slab +- `margin` planes.  The trapping regions need two global inputs, exchanged by the slab
 * scheduler between the two calls: the 26-neighbour maxima of every rank (tiny) and the per-brick move
 * masks (one int per 8^3 brick; every rank contributes the bricks of its own slab).
 *   xb_set_table_window -> xb_table_build -> [all-gather seeds + masks] -> xb_table_finish -> xb_assign_trace */
int xb_set_table_window(xb_ctx *c, int64_t margin);           /* margin < 0: whole grid (default) */
int xb_table_build(xb_ctx *c, int64_t *n_local_seeds);
int xb_table_local_seeds(xb_ctx *c, int64_t *out, int64_t capacity);  /* linear voxel indices */
int xb_brick_masks(xb_ctx *c, void **dev_ptr, int64_t *n_bricks, int64_t *own_first, int64_t *own_count);
/* does this rank's window hold a voxel whose record depends on the tie rule (methods.py:324 vs refinement.py:111)? */
int xb_table_ties(xb_ctx *c, int64_t *has_ties);
/* any_ties: the OR of xb_table_ties over all ranks */
int xb_table_finish(xb_ctx *c, const int64_t *seeds, int64_t n_seeds, int64_t any_ties);

/* ---- slab halo planes (multi-GPU) -------------------------------------------------------- */
/* Device pointers of the label / known arrays and the plane size in elements, so that the slab
 * scheduler can hand plane ranges to RCCL (ncclSend/ncclRecv) or any other transport.  On a slab, writes through
 * the label pointer must stay within the planes [x0 - halo, x1 + halo): the library keeps the rest zero. */
void *xb_labels_ptr(xb_ctx *c);
void *xb_known_ptr(xb_ctx *c);
void *xb_density_ptr(xb_ctx *c);
int64_t xb_plane_elems(xb_ctx *c);
/* copy whole x-planes [xa,xb) of labels/known between host and device (halo transport over the
 * host / gloo; the RCCL transport works on the device pointers above) */
int xb_copy_planes(xb_ctx *c, int which /*0 labels,1 known*/, int to_device, void *host, int64_t xa, int64_t xb);
/* Bytes per label (1 / 2 / 4) a label halo travels in (xb_comm_exchange_planes): the narrowest signed width that holds every
 * resident label -- the reference's own dtype_calc(-n_maxima) (thread_handlers.py:70-74, utils.py:25-37).  Sender and
 * receiver must agree on it.  It changes only in calls all ranks make alike (numbering, xb_upload_labels,
 * xb_vacuum_assign, xb_volume_assign); xb_scatter_voxels / xb_copy_planes widen it only when a label they write does not fit.
 * widen_to 1 / 2 / 4 raises it (0: only ask): after a per-rank write a scheduler sets the maximum over the ranks. */
int xb_label_wire(xb_ctx *c, int widen_to, int *wire_out);
/* the chunk [first, first+count) of the per-brick move masks (xb_brick_masks) and of the bricks' single-maximum voxels
 * (2 * count ints: masks, then voxels) from (to_device 1) or to (0) host memory -- the host-staged fallback of xb_comm_share_brick_masks */
int xb_brick_masks_copy(xb_ctx *c, int to_device, int32_t *host, int64_t first, int64_t count);
int xb_set_halo(xb_ctx *c, int64_t halo); /* planes each side of [x0,x1) that hold valid neighbour data */

/* ---- the slab step with its control flow on the device (csrc/slab_step.h) ------------------------------------
 * The same work as xb_table_build .. xb_assign_finish and xb_edge_find + xb_refine_trace above (thread_handlers.py:28-75,
 * 128-236 across slabs), but no list length, counter or table goes through the host between the launches: TWO host waits
 * per assignment + refinement pass instead of about fifteen.  Between the calls the scheduler (pybader_amd/slab.py) runs the
 * exchanges on device "blocks" (xb_slab_block): 0-2 the brick arrays of pass A, 3 the ranks' tie flags, 4 their maxima
 * tables, 6 / 7 the walkers of a refinement pass (all gathered: xb_comm_allgather_block), 5 the counters of a refinement
 * pass (summed: xb_comm_allreduce_block).
 *   xb_slab_assign_masks -> [blocks 0-3] -> xb_slab_assign_trace -> [block 4] -> xb_slab_assign_finish (waits)
 *   [label halos] -> xb_slab_refine_pass -> { [block 6 / 7] -> xb_slab_walkers_round } -> [block 5] -> xb_slab_refine_counts (waits)
 * xb_slab_supported: whole-brick slabs with a table window, no vacuum, at most 64 ranks.  xb_slab_assign_finish status:
 * 0 done, 1 repeat the step (the region growth wants its long schedule; every rank sees the same verdict), 2 use the
 * host-driven calls for this density (a rank has more than 1024 maxima or trajectories for the exact slow kernel). */
int xb_slab_supported(xb_ctx *c, int nranks, int64_t *ok);
int xb_slab_assign_masks(xb_ctx *c, int rank, int nranks);
int xb_slab_assign_trace(xb_ctx *c);
int xb_slab_assign_finish(xb_ctx *c, int64_t *n_maxima, int64_t *status);
int xb_slab_refine_pass(xb_ctx *c);
/* the walkers of the pass (retraces that left this rank's valid planes) travel in blocks 6 / 7, alternately: after block
 * 6 + src was gathered, its results are applied and -- unless `last` -- its walkers that arrive on this rank's planes are
 * carried on into this rank's part of the other block.  The last round packs the counters into block 5. */
int xb_slab_walkers_round(xb_ctx *c, int src, int last);
/* what travels of a rank's part of blocks 6 / 7 (fixed sizes: no count goes through the host): [0] bytes of a part,
 * [1] header + walkers of the pass itself, [2] header + walkers of a later round, [3] offset and [4] bytes of the results */
int xb_slab_walk_layout(xb_ctx *c, int64_t out[5]);
/* (round 5) how many walkers of a rank's part travel in the gather of the NEXT passes (0: the part's capacity).  The scheduler sets it
 * on every rank to the same value -- a bound on any rank's share it knows from the previous pass's summed export count -- and asks
 * xb_slab_walk_layout again; a rank that exports more loses the surplus to the host-driven path queries, as with a full part.
 * Replaces nothing in the reference (its threads share memory, thread_handlers.py:154-232). */
int xb_slab_walk_send(xb_ctx *c, int64_t walkers);
/* local[8], global[8]: edges, changed, escaped, walkers still travelling after the last round (all ranks; not summed),
 * retraces redone by the exact slow kernel (then local[1], local[2] are the counts after it and the caller sums once more),
 * walkers lost to a full block or stuck (their voxels stay parked: xb_escaped_paths), this rank's travelling walkers
 * (xb_walkers_fetch), 0 */
int xb_slab_refine_counts(xb_ctx *c, int64_t *local, int64_t *global);
int xb_slab_block(xb_ctx *c, int which, void **dev_ptr, int64_t *bytes_total, int64_t *own_offset, int64_t *own_bytes);
/* bytes [off, off + bytes) of a block from (to_device 1) or to (0) host memory: the host-staged transports */
int xb_slab_block_copy(xb_ctx *c, int which, int to_device, void *host, int64_t off, int64_t bytes);
/* waits of the host for the card inside library calls made by the calling thread since it started */
int xb_host_waits(int64_t *n);

/* ---- measurement ------------------------------------------------------------------------- */
/* HIP-event timing of the stages, measured on the context's stream: accumulated milliseconds and launch
 * count since the last reset.  which: 0 neargrid assignment after pass A (walk list, records, walker trace),
 * 1 the ongrid pointer pass (k_og_masks), 2 edge_find, 3 refine trace, 4 pass A + region growth (and records built
 * for a refinement), 5 k_brick_masks alone, 6 the trace kernel alone, 7 k_brick_records alone. */
int xb_kernel_time(xb_ctx *c, int which, double *ms_total, int64_t *launches);
int xb_kernel_time_reset(xb_ctx *c);
/* on: 0 off, 1 every timer, otherwise a mask: bit k + 1 switches timer `which` = k on (event pairs between dependent kernels
 * cost stream time: a benchmark keeps only the dominant kernel's timer on inside its timed region) */
int xb_enable_timing(xb_ctx *c, int on);
/* Switches (nine keys).  A USER of the library sets none of them: every default is the measured best, and no switch changes a
 * result.  What each is for:
 *   6   drop the cached gradient-field table (benchmarks: a table kept from an earlier step would hide 1.6 ms per step);
 *   3   debug (bit 2 edge_check passes, bit 4 slab statistics, bit 5 wait after every stage of an assignment, bit 6 the exact
 *       slow path with tiers of 3 / 5 / 8 path voxels, so that a test reaches its last tier);
 *   24  collectives return without waiting (set by pybader_amd.slab itself);
 *   1   trapping regions (0: plain full trajectories from a record per voxel) -- the exactness cross-check of the whole design;
 *   2   cross-check bits, each selecting the second implementation of one step so that a test can compare the two:
 *       1 no mirror prefilter in pass A, 2 the generic walker instead of the lean one, 4 the full T_grad . grad product on
 *       orthogonal lattices, 8 dilation from the edge list instead of tile by tile, 16 int32 label halos, 32 no front sharing
 *       in the edge_check chase;
 *   test plumbing -- 4 / 5 workgroups and LDS queue capacity of the edge_check chase (lowered to force the overflow
 *   hand-over), 17 kill launches scheduled after a chase (1 forces the repeat), 19 a rank may exchange planes with itself.
 * (Round 4 removed 0, 2, 9-12, 15, 21; round 6 removed 7, 8, 16, 22 -- routes and launch shapes nobody set -- and folded 13, 14,
 * 18, 20, 25, 29 into the bits of key 2.) */
int xb_set_option(xb_ctx *c, int key, int value);
/* device bytes held for the grid (density, labels, flags, numbering + the table of the window planes + scratch sized by the
 * slab): what a rank of the slab decomposition costs; the reference's blocks are copies of the block extent
 * (thread_handlers.py:31-47, utils.py:424-458) */
/* Page-locked host buffers for result arrays (no counterpart in the reference: numpy allocates its own).  A device-to-host
 * copy into such a buffer is one DMA transfer; into pageable memory it is staged through the library's pinned chunks and copied
 * again.  pybader_amd._lib hands out the narrowed label arrays of bader_calc (thread_handlers.py:70-74) from a pool of these. */
int xb_host_alloc(int64_t bytes, void **out);
int xb_host_free(void *p);
int xb_memory_stats(xb_ctx *c, int64_t *bytes_total, int64_t *bytes_table, int64_t *bytes_scratch);
/* statistics of the last assignment: trapping boxes found and voxels they cover */
int xb_box_stats(xb_ctx *c, int64_t *n_boxes, int64_t *box_voxels);
/* The brick lattice of the last neargrid assignment and, per 8^3 brick (C order over dims), the trapping region it was
 * certified for (> 0), 0 (its voxels were walked) or INT32_MIN (round 6: with a vacuum tolerance, a brick whose largest
 * density lies below it -- nothing to assign, no records).  dims = ceil(shape / 8): a brick the grid cuts holds the voxels the
 * grid leaves of it.  A diagnostic of the library's own decomposition (the reference has no counterpart); capacity in ints. */
int xb_brick_labels(xb_ctx *c, int32_t *out, int64_t capacity, int64_t dims[3]);
/* trajectories / retraces handed to the exact slow kernel since the context was created */
int xb_slow_path_stats(xb_ctx *c, int64_t *assign_total, int64_t *refine_total);
/* retraces redone by the from-rho kernel since the context was created (their walk went on through a brick whose
 * records the sparse table does not hold) */
int xb_deferred_stats(xb_ctx *c, int64_t *refine_total);
/* region growth: assignments repeated because the scheduled kill launches (option 17; 6 after a chase) did not reach the
 * fixpoint, and the schedule this context uses now (raised to the worst case by the first repeat) */
int xb_growth_stats(xb_ctx *c, int64_t *retries, int64_t *kill_launches);

/* ---- multi-GPU transport: RCCL over xGMI, one process per GPU (no PyTorch) ------------------------------------
 * Replaces nothing in the reference -- its thread blocks share one address space (thread_handlers.py:28-58,
 * 154-205); these calls move what the slab scheduler (pybader_amd/slab.py) has to move between GPUs.  librccl is
 * loaded on the first call.  Rank 0 makes the unique id, the host side distributes its 128 bytes. */
int xb_comm_unique_id(uint8_t id_out[128]);
int xb_comm_init(xb_ctx *c, int rank, int nranks, const uint8_t id_in[128]);
int xb_comm_destroy(xb_ctx *c);
/* planes [xa, xb) of the label (which 0) / known (which 1) array to and from ring neighbours, one ncclGroup */
int xb_comm_exchange_planes(xb_ctx *c, int which, int n_send, const int32_t *send_peer, const int64_t *send_xa,
                            const int64_t *send_xb, int n_recv, const int32_t *recv_peer, const int64_t *recv_xa,
                            const int64_t *recv_xb);
/* n int64 reduced in place over the ranks (op 0 sum, 1 min, 2 max): the counters thread_handlers.refine sums */
int xb_comm_allreduce_i64(xb_ctx *c, int64_t *inout, int64_t n, int op);
/* n int64 from every rank, out[size * n] in rank order: maxima tables (thread_handlers.py:59-65), seeds */
int xb_comm_allgather_i64(xb_ctx *c, const int64_t *in, int64_t n, int64_t *out);
/* every rank's chunk [first[r], first[r]+count[r]) of the brick move masks (xb_brick_masks) and of the bricks'
 * single-maximum voxels to every rank */
int xb_comm_share_brick_masks(xb_ctx *c, const int64_t *first, const int64_t *count);
/* blocks 0-4 of the device-driven slab step: part [first[r], first[r] + count[r]) (bytes) from rank r to every rank;
 * block 5: summed := sum over the ranks of local.  Ordered on the context's stream, no host wait. */
int xb_comm_allgather_block(xb_ctx *c, int which, const int64_t *first, const int64_t *count);
int xb_comm_allreduce_block(xb_ctx *c);
/* what the communicator says about the run: out = {ncclCommCount, ncclCommUserRank, ncclCommCuDevice, ncclGetVersion}
 * (-1 where this librccl lacks the entry point) -- lets a multi-GPU bench line prove it ran on N ranks and N devices */
int xb_comm_info(xb_ctx *c, int64_t out[4]);
/* plane bytes this rank has sent since xb_comm_init (label halos travel as dtype_calc(-n_maxima): int8 for up to 127 basins) */
int xb_comm_stats(xb_ctx *c, int64_t *bytes_sent);

#ifdef __cplusplus
}
#endif
#endif /* BADER_HIP_H */
