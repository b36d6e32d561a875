"""Slab scheduler: the multi-GPU replacement of pybader/thread_handlers.py's block split
(thread_handlers.py:28-47, 154-174), one process per GPU.

Decomposition: contiguous slabs of axis 0 (the slow axis of the C-order
arrays, so a slab and its halo planes are contiguous memory).  The float64 density is replicated on
every GPU (8 GiB at 1024^3 of 288 GB): every trajectory step is a local read and the remainder `dr`
is carried exactly, which keeps the N-slab result identical to the 1-GPU result.  Labels / known
flags are full-size arrays on every rank of which only the owned planes + `halo` planes each side
are kept valid; halos are refreshed by point-to-point plane exchange (RCCL send/recv through the library's
own C ABI, pybader_amd/comm.py; the CPU tests plug in a gloo or TCP transport over a host backend) before
every edge sweep.  Collectives: only tiny ones (maxima tables, counters).

The scheduler is written against a small backend interface so that the CPU tests can drive it with a
host backend; the product backend is `GpuBackend` (libbader_hip.so)."""
import os
import sys
import time

import numpy as np


def slab_ranges(nx, nranks):
    """Contiguous planes of axis 0 per rank.  A grid of whole 8^3 bricks with at least one brick per rank is split on brick
    boundaries (np.array_split over the BRICKS: the first (nx / 8) % n slabs get one brick more) -- a slab that cuts bricks
    cannot use the trapping regions and traces its planes in full (round 4).  Otherwise np.array_split over the planes, as the
    reference splits its blocks (thread_handlers.py:35): the first nx % n slabs get one plane more."""
    nx, nranks = int(nx), int(nranks)
    unit = 8 if nx % 8 == 0 and nx // 8 >= nranks else 1
    base, extra = divmod(nx // unit, nranks)
    out, x = [], 0
    for r in range(nranks):
        w = (base + (1 if r < extra else 0)) * unit
        out.append((x, x + w))
        x += w
    return out


def merge_maxima_tables(tables):
    """Global basin numbering from per-rank (maximum index, smallest owned voxel reaching it) tables:
    min over ranks, then rank by that smallest voxel index -- the order the reference's C-order scan
    discovers maxima (thread_handlers.py:59-65 for one block)."""
    best = {}
    for m, f in tables:
        for mi, fi in zip(np.asarray(m).tolist(), np.asarray(f).tolist()):
            if mi not in best or fi < best[mi]:
                best[mi] = fi
    order = sorted(best, key=lambda k: best[k])
    return np.array(order, dtype=np.int64)


def halo_plan(ranges, rank, halo, nx):
    """Plane runs this rank must receive: list of (peer, xa, xb) covering [x0-halo, x0) and
    [x1, x1+halo) modulo nx, split by owner; and the mirror list of runs it must send."""
    def owner(p):
        for r, (a, b) in enumerate(ranges):
            if a <= p < b:
                return r
        raise ValueError(p)

    def runs(planes):
        out = []
        for p in planes:
            o = owner(p)
            if out and out[-1][0] == o and out[-1][2] == p:
                out[-1][2] = p + 1
            else:
                out.append([o, p, p + 1])
        return [tuple(r) for r in out]

    def wanted(r):
        a, b = ranges[r]
        if (b - a) + 2 * halo >= nx:
            need = [p for p in range(nx) if not (a <= p < b)]
        else:
            need = [(a - halo + k) % nx for k in range(halo)] + [(b + k) % nx for k in range(halo)]
        return runs(need)

    recvs = [r for r in wanted(rank) if r[0] != rank]
    sends = []
    for peer in range(len(ranges)):
        if peer == rank:
            continue
        for o, xa, xb in wanted(peer):
            if o == rank:
                sends.append((peer, xa, xb))
    return sends, recvs


class GpuBackend:
    """libbader_hip.so context behind the scheduler's backend interface (no PyTorch: planes travel through the
    library's own RCCL transport, pybader_amd/comm.py)."""

    def __init__(self, ctx, device_index=0):
        self.ctx = ctx
        self.device_index = device_index

    def set_grid(self, shape, dist_mat, T_grad, x_range, halo):
        self.ctx.set_grid(shape, dist_mat, T_grad, x_range)
        self.sliced = tuple(x_range) != (0, shape[0])
        if self.sliced:
            self.ctx.set_halo(halo)

    def set_halo(self, halo):
        if self.sliced:
            self.ctx.set_halo(halo)

    def brick_mask_range(self):
        """(first, count) of the bricks whose move masks this rank computed"""
        _, _, first, count = self.ctx.brick_masks()
        return int(first), int(count)

    def __getattr__(self, name):          # assign_trace, assign_finish, edge_find, refine_trace, sync ...
        return getattr(self.ctx, name)


class _Phase:
    """optional wall-clock accounting of the scheduler's phases (XB_SLAB_TIMING=1; bench.py reports it)"""

    def __init__(self, runner, name):
        self.r, self.name = runner, name

    def __enter__(self):
        if self.r.timing is not None:
            self.r.be.sync()
            self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        if self.r.timing is not None:
            self.r.be.sync()
            self.r.timing[self.name] = self.r.timing.get(self.name, 0.0) + time.perf_counter() - self.t0
        return False


class SlabRunner:
    """bader_calc + refine over slabs; with one rank it degenerates to the single-GPU calls."""

    def __init__(self, backend, comm, shape, dist_mat, T_grad, halo=8):
        self.be, self.comm = backend, comm
        self.shape = tuple(int(s) for s in shape)
        self.ranges = slab_ranges(self.shape[0], comm.size)
        self.x_range = self.ranges[comm.rank]
        # (two planes carry a one-GPU run; across slabs the 'changed' refinement's edge_check needs three: refuse less up
        # front rather than mid-run, after collectives have been issued)
        self.halo = max(3 if comm.size > 1 else 2, min(int(halo), self.shape[0]))
        self.be.set_grid(self.shape, dist_mat, T_grad, self.x_range, self.halo)
        self.sends, self.recvs = halo_plan(self.ranges, comm.rank, self.halo, self.shape[0])
        self.n_maxima = 0
        self.n_fallbacks = 0
        self.n_device_steps = 0      # assignments that ran as the device-driven step (csrc/slab_step.h)
        self.timing = {} if os.environ.get('XB_SLAB_TIMING') else None
        if comm.size > 1 and hasattr(comm, 'selftest_planes'):
            self.be.sync()
            comm.selftest_planes(self.be, self.ranges)

    @property
    def maxima(self):
        """linear voxel indices of the maxima in basin order (fetched from the library on first use: not part of a step)"""
        if getattr(self, '_maxima', None) is None:
            self._maxima = (np.ravel_multi_index(tuple(self.be.maxima().T), self.shape) if self.n_maxima else np.zeros(0, np.int64))
        return self._maxima

    def enable_table_window(self, margin=None):
        """Build the gradient-field table only for the owned slab +- margin planes (needs whole 8^3 bricks everywhere).
        Trajectories that leave the window are redone with records derived from rho, a kernel with a long tail: at 512^3 on
        eight slabs 32 planes leave 5-9 thousand of them per rank (0.15 ms), 64 planes a few hundred (still 0.1 ms), 128
        none -- for 80 % more records (+0.07 ms).  Default: a quarter of the axis, at least 32 planes, clamped so that a
        window remains.  Returns whether the window is active."""
        nx, ny, nz = self.shape
        if margin is None:
            fit = ((nx - max(b - a for a, b in self.ranges) - 8) // 2) // 8 * 8
            margin = max(8, min(max(32, nx // 4), fit))
        ok = (self.comm.size > 1 and hasattr(self.be, 'set_table_window') and nx % 8 == 0 and ny % 8 == 0 and nz % 8 == 0
              and all(a % 8 == 0 and b % 8 == 0 for a, b in self.ranges)
              and (self.x_range[1] - self.x_range[0]) + 2 * (margin + 7) // 8 * 8 < nx)
        # every rank must take the same branch (collectives below)
        ok = all(self.comm.allgather(bool(ok))) if self.comm.size > 1 else False
        if ok:
            self.be.set_table_window(margin)   # walks beyond it derive their records from rho on the spot
        self.windowed = ok
        self.table_margin = margin if ok else None
        return ok

    # ---- the step with its control flow on the device (csrc/slab_step.h): two host waits per assignment + refinement pass ----
    def _device_step(self):
        """whether this decomposition takes the device-driven step: voted once (every rank must take the same branch), then
        checked locally per call (vacuum switches it off; the tolerance is the same on every rank)"""
        if getattr(self, '_step_vote', None) is None:
            ok = (getattr(self, 'windowed', False) and hasattr(self.be, 'slab_supported') and hasattr(self.comm, 'allgather_block')
                  and not os.environ.get('XB_SLAB_HOST_DRIVEN'))
            self._step_vote = all(self.comm.allgather(bool(ok)))
            if self._step_vote and hasattr(self.be, 'set_option'):
                self.be.set_option(24, 1)      # collectives are ordered on the stream: nothing waits for them on the host
        return self._step_vote and not getattr(self, '_step_declined', False) and self.be.slab_supported(self.comm.size)

    def _guard(self, what):
        """the collectives of the device-driven step are ordered on the stream: a peer that never arrives shows at the
        step's host waits, so those carry the transport's hang watchdog"""
        import contextlib
        g = getattr(self.comm, '_guard', None)
        return g(what) if g is not None else contextlib.nullcontext()

    def _parts(self, which):
        """(offset, bytes) of every rank's part of exchange block `which` (static for a decomposition)"""
        cache = self.__dict__.setdefault('_block_parts', {})
        if which not in cache:
            _, _, off, n = self.be.slab_block(which)
            cache[which] = self.comm.allgather((int(off), int(n)))
        return cache[which]

    def _assign_device_step(self):
        """-> n_maxima, or None when this density is not for the device-driven step (the host-driven calls take over)"""
        for _ in range(2):
            with _Phase(self, 'masks'):
                self.be.slab_assign_masks(self.comm.rank, self.comm.size)
            with _Phase(self, 'brick_exchange'):
                for which in (0, 1, 2, 3):
                    self.comm.allgather_block(self.be, which, self._parts(which))
            with _Phase(self, 'assign_trace'):
                self.be.slab_assign_trace()
            with _Phase(self, 'maxima_exchange'):
                self.comm.allgather_block(self.be, 4, self._parts(4))
            with _Phase(self, 'assign_finish'), self._guard('assignment'):
                n, status = self.be.slab_assign_finish()
            if status == 0:
                self.n_maxima = n
                self._maxima = None       # (the sorted maxima stay in the library until somebody asks: `maxima`)
                return n
            if status == 2:
                return None
        raise RuntimeError('the region growth asked for a repeat twice')

    def _refine_pass_device(self):
        """edge sweep + retraces of one iteration: -> (edges, changed), summed over the ranks"""
        self.exchange_label_halo()
        with _Phase(self, 'refine_pass'):
            self.be.slab_refine_pass()
        # Some retraces walk out of the valid planes before they meet a known == 2 voxel (they slide along a dividing
        # surface for tens of planes: about 1 % of them with a 16-plane halo at 512^3).  They are parked and exported as
        # walkers; the rank that owns the plane they entered carries them on.  A fixed number of rounds runs without asking
        # the host whether any walker is left (an empty round costs a few small launches); what still travels afterwards
        # (or needs the exact slow path) is finished by the host-driven loop.
        rounds = int(os.environ.get('XB_SLAB_WALKER_ROUNDS', 0)) or getattr(self, '_walker_rounds', 3)
        with _Phase(self, 'walkers'):
            # (round 5) what travels of a part: as many walkers as the previous pass makes us expect (three times the mean share of
            # its summed export count) instead of the part's capacity (2.6 MB per rank at 512^3, gathered blind on every rank);
            # every rank computes the same number from the same sums.  No history, or a pass that lost walkers to the limit: the
            # capacity.  (8 slabs of 512^3: 13 K of 32 K walkers per rank travel -- 12 instead of 21 MB per gather.)
            want = getattr(self, '_walk_send', 0)
            if want != getattr(self, '_walk_send_set', None) and hasattr(self.be, 'slab_walk_send'):
                self.be.slab_walk_send(want)
                self._walk_send_set = want
                self._walk_layout = None
            if getattr(self, '_walk_layout', None) is None:
                self._walk_layout = self.be.slab_walk_layout()
            part, first, later, res_off, res_n = self._walk_layout
            src = 0
            for k in range(rounds + 1):
                # (fixed sizes: the walkers the pass may export / a later round may export again, and every result)
                self.comm.allgather_block(self.be, 6 + src, [(r * part, first if k == 0 else later) for r in range(self.comm.size)])
                if k:
                    self.comm.allgather_block(self.be, 6 + src, [(r * part + res_off, res_n) for r in range(self.comm.size)])
                self.be.slab_walkers_round(src, k == rounds)
                src ^= 1
        with _Phase(self, 'sums'):
            self.comm.allreduce_block(self.be)
        with _Phase(self, 'refine_wait'), self._guard('refinement pass'):
            loc, glo = self.be.slab_refine_counts()
        edges, changed = int(glo[0]), int(glo[1])
        if os.environ.get('XB_SLAB_DEBUG'):
            print(f'[rank {self.comm.rank}] refinement pass: local {loc.tolist()} summed {glo.tolist()}', file=sys.stderr, flush=True)
        if glo[4]:      # rare: some retraces went through the exact slow kernel after the sums were taken
            with _Phase(self, 'sums'):
                changed, = self.comm.sum(int(loc[1]))
        # (rounds of the next pass: as many as carried a walker on in this one, two more after a pass that left some travelling
        # -- those went through the host-driven loop below; every rank sees the same sums)
        used = max([k + 1 for k in range(4) if (int(glo[7]) >> (8 * k)) & 0xff] or [0])
        self._walker_rounds = min(6, max(1, used) + (2 if loc[3] else 0))
        # (glo[2]: retraces that left their rank's planes, all ranks and rounds together; three times the mean share covers the uneven
        # split between ranks -- the interior slabs of an 8-atom cell export twice the mean; glo[5]: lost or stuck -> the capacity again)
        self._walk_send = 0 if glo[5] else max(2048, 3 * int(glo[2]) // max(1, self.comm.size))
        if glo[2]:
            self.n_fallbacks += 1
        if loc[3] or glo[5]:
            changed += self._finish_escaped(int(loc[6] + loc[5]))
            self.n_fallbacks -= 1
        return edges, changed

    def assign(self, method):
        """thread_handlers.bader_calc: per-slab trajectories, then one tiny table merge for numbering."""
        if self.comm.size == 1 and hasattr(self.be, 'assign'):
            # one GPU: the library's own xb_assign (control flow on the device, one host wait)
            self.n_maxima = int(self.be.assign(method))
            self._maxima = None
            return self.n_maxima
        self._stepped = False
        if method == 'neargrid' and self._device_step():
            n = self._assign_device_step()
            if n is not None:
                self.n_device_steps += 1
                self._stepped = True      # (the refinement may use the device-driven pass: labels, regions and table are this step's)
                return n
            self._step_declined = True    # this density: host-driven from here on (every rank got the same status)
        if getattr(self, 'windowed', False) and method == 'neargrid':
            # windowed table: the trapping regions need every rank's maxima and brick masks
            with _Phase(self, 'table_build'):
                local = self.be.table_build()
            with _Phase(self, 'seeds_allgather'):
                # (rows of one int64: the owned maxima the cube seeding wants, -1 - ties flag last; brick seeding sends the flag only)
                mine = np.array(list(np.asarray(local).tolist()) + [-1 - int(bool(self.be.table_ties()))], np.int64).reshape(-1, 1)
                got = self._gather_rows(mine).reshape(-1)
                seeds = sorted(set(int(v) for v in got if v >= 0))
                any_ties = bool((got == -2).any())
            with _Phase(self, 'mask_exchange'):
                self.be.sync()
                if getattr(self, '_chunks', None) is None:      # static for a given decomposition
                    self._chunks = self.comm.allgather(self.be.brick_mask_range())
                self.comm.share_brick_masks(self.be, self._chunks)
            with _Phase(self, 'table_finish'):
                self.be.table_finish(np.array(seeds, dtype=np.int64), any_ties)
        with _Phase(self, 'assign_trace'):
            m, f = self.be.assign_trace(method)
        with _Phase(self, 'maxima_merge'):
            if self.comm.size == 1:
                order = np.argsort(f, kind='stable')
                maxima = np.asarray(m)[order]
            else:
                rows = self._gather_rows(np.stack([np.asarray(m, np.int64), np.asarray(f, np.int64)], axis=1).reshape(-1, 2))
                maxima = merge_maxima_tables([(rows[:, 0], rows[:, 1])])
        with _Phase(self, 'assign_finish'):
            self.be.assign_finish(maxima)
        self.n_maxima = int(maxima.shape[0])
        self._maxima = maxima
        return self.n_maxima

    def _gather_rows(self, rows):
        """every rank's int64 rows in rank order: through the device transport when the communicator has one"""
        rows = np.ascontiguousarray(rows, np.int64)
        if hasattr(self.comm, 'gather_rows'):
            return self.comm.gather_rows(rows)
        return np.concatenate(self.comm.allgather(rows))

    def exchange_label_halo(self):
        if self.comm.size > 1:
            with _Phase(self, 'label_halo'):
                if not getattr(self.comm, 'stream_ordered', False):
                    self.be.sync()
                self.comm.exchange_planes(self.be, 0, self.sends, self.recvs)

    def _trace(self, edges=None):
        """one retrace pass over the flagged edge voxels; `edges`: this rank's edge count of the sweep before, summed
        in the same collective as the pass's own counters.  Returns changed, or (changed, edges summed) when given."""
        if edges is not None:
            changed, total = self._trace_counted(edges)
            return changed, total
        return self._trace_counted(0)[0]

    def _trace_counted(self, edges):
        with _Phase(self, 'refine_trace'):
            changed, escaped = self.be.refine_trace()
        if self.comm.size == 1:
            assert escaped == 0
            return changed, edges
        local_escaped = escaped
        with _Phase(self, 'sums'):
            changed, escaped, edges = self.comm.sum(changed, escaped, edges)
        if escaped:
            changed += self._finish_escaped(local_escaped)
        return changed, edges

    def _finish_escaped(self, local_escaped):
        """the retraces that left the valid planes (parked, exported as walkers): -> voxels relabelled by finishing them,
        summed over the ranks"""
        # Some retraces walked out of the valid planes before meeting a known==2 voxel (they slide along a dividing
        # surface for tens of planes: about 1 % of them with a 16-plane halo at 512^3).  They were parked (known == -6)
        # untouched and exported as walkers: the rank that owns the plane they entered carries them on.
        self.n_fallbacks += 1
        changed, left = 0, local_escaped
        if hasattr(self.be, 'walkers_continue') and hasattr(self.comm, 'gather_rows'):
            with _Phase(self, 'walkers'):
                changed, left = self._migrate_walkers(local_escaped)
        with _Phase(self, 'sums'):
            changed, left = self.comm.sum(changed, left)
        if left:      # no walker transport (host backend), or walkers that need the exact slow path
            with _Phase(self, 'escaped_path_queries'):
                ch3 = self._resolve_escaped()
            with _Phase(self, 'sums'):
                ch3, = self.comm.sum(ch3)
            changed += ch3
        return changed

    def _migrate_walkers(self, local_escaped):
        """Rounds of: all-gather the open walkers together with the results of the round before; apply the results
        that concern the owned voxels; carry on the walkers that arrived on an owned plane.  Every rank sees the same
        gathered rows, so all of them leave the loop in the same round (no walker row left).  Returns (voxels
        relabelled here, voxels still parked here)."""
        words = self.be.WALKER_WORDS
        nyz = self.shape[1] * self.shape[2]
        x0, x1 = self.x_range
        rows, _ = self.be.walkers()
        left = local_escaped - rows.shape[0]          # (an export buffer that ran full: those stay parked)
        changed = 0
        for _ in range(4 * self.comm.size + 16):
            allrows = self.comm.gather_rows(rows)
            is_res = allrows[:, 1] == -1               # result rows: word 0 = voxel | label << 32, word 1 = -1
            if is_res.any():
                res = allrows[is_res, 0]
                plane = (res & 0xffffffff) // nyz      # only the pairs of owned voxels go to the card
                ch, stuck = self.be.walkers_apply(res[(plane >= x0) & (plane < x1)])
                changed += ch
                left += stuck
            if is_res.all():
                return changed, left
            wk = allrows[~is_res]
            plane = ((wk[:, 1] >> 32) & 0xffffffff) // nyz     # word 1 = last voxel | voxel arrived at << 32
            w, res = self.be.walkers_continue(wk[(plane >= x0) & (plane < x1)])
            rrows = np.full((res.size, words), -1, np.int64)
            rrows[:, 0] = res
            rows = np.concatenate([w, rrows])
        raise RuntimeError('walkers still travelling after a full tour of the ranks')

    def _resolve_escaped(self):
        """Finish the parked retraces without moving planes: a retrace's path depends on rho only (replicated),
        so its owner records the whole trajectory up to the maximum; the ranks owning the path voxels answer
        (label, known) for them; the retrace ends on the first known == 2 voxel of its path
        (refinement.py:294-303) or on the maximum (283-292) and takes that voxel's label.  Traces only read
        labels at known==2 voxels / maxima, which no retrace rewrites, so asking in the middle of an
        iteration is safe.  Two small all-gathers instead of every rank's planes."""
        total = 0
        # most retraces stop a few voxels beyond the halo: ask about short tails first
        for max_len in (self.halo + 16, self.halo + 64, 1024, 1 << 15):
            t0 = time.perf_counter()
            starts, offsets, vox, complete = self.be.escaped_paths(max_len)
            t1 = time.perf_counter()
            moved, open_ = self._ask_owners(starts, offsets, vox, complete)
            if os.environ.get('XB_SLAB_DEBUG'):
                print(f'[rank {self.comm.rank}] path queries max_len {max_len}: {starts.size} paths, {vox.size} voxels, '
                      f'dump {1e3 * (t1 - t0):.1f} ms, ask {1e3 * (time.perf_counter() - t1):.1f} ms, open {open_}',
                      file=sys.stderr, flush=True)
            total += moved
            if self.comm.sum(open_)[0] == 0:
                self._agree_label_wire()
                return total
        raise RuntimeError('escaped retraces unresolved after full-length path queries')

    def _agree_label_wire(self):
        """Only the ranks with parked retraces wrote voxels (scatter_voxels): the width a label halo travels in must be the
        same on both ends of every send / receive pair, so the ranks take the maximum of theirs (the library
        widens on such a write only when a label does not fit, so this is normally a no-op)."""
        wire = getattr(self.be, 'label_wire', None)
        if wire is None or self.comm.size == 1:
            return
        self.be.label_wire(max(int(w) for w in self.comm.allgather(int(wire()))))

    def _ask_owners(self, starts, offsets, vox, complete):
        nyz = self.shape[1] * self.shape[2]
        bounds = np.array([b for _, b in self.ranges], np.int64)           # rank r owns planes [a_r, b_r)
        uniq = np.unique(vox)
        owner = np.searchsorted(bounds, uniq // nyz, side='right')
        ask = {int(r): uniq[owner == r] for r in np.unique(owner)}           # what I ask of each owner
        all_asks = self.comm.allgather(ask)
        mine = {src: self.be.gather_voxels(q[self.comm.rank]) for src, q in enumerate(all_asks) if self.comm.rank in q}
        all_answers = self.comm.allgather(mine)
        if starts.size == 0:
            return 0, 0
        lab = np.zeros(uniq.size, np.int32)
        kn = np.zeros(uniq.size, np.int8)
        for r, q in ask.items():
            a_lab, a_kn = all_answers[r][self.comm.rank]
            pos = np.searchsorted(uniq, q)
            lab[pos], kn[pos] = a_lab, a_kn
        pos = np.searchsorted(uniq, vox)
        p_lab, p_kn = lab[pos], kn[pos]
        # first known == 2 voxel after the start of each path (segmented search)
        is2 = p_kn == 2
        is2[offsets[:-1]] = False
        idx2 = np.flatnonzero(is2)
        first2 = np.searchsorted(idx2, offsets[:-1])                        # position in idx2 of the first hit >= path start
        hit = (first2 < idx2.size)
        stop = np.where(hit, idx2[np.minimum(first2, max(idx2.size - 1, 0))] if idx2.size else 0, 0)
        hit &= stop < offsets[1:]
        done = hit | complete
        stop = np.where(hit, stop, offsets[1:] - 1)
        old = p_lab[offsets[:-1]]
        new_lab = p_lab[stop]
        moved = done & (new_lab != old)
        sel = np.flatnonzero(done)
        # refinement.py:288-291: a relabelled voxel stays flagged (-2), the others become plain near-edge (-1)
        self.be.scatter_voxels(starts[sel], np.where(moved, new_lab, old)[sel], np.where(moved, -2, -1).astype(np.int8)[sel])
        return int(moved.sum()), int((~done).sum())

    def _edge_check_slabs(self):
        """refinement.edge_check (refinement.py:409-508) across slabs.  Its greedy scan is global -- whether a changed
        voxel is processed depends on its C-order earlier changed neighbours, in chains that run through slab
        boundaries -- but it only involves the changed voxels and one class bit each: every rank lists its owned
        changed voxels with their class, the lists are all-gathered, and every rank resolves the global list with
        the single-GPU kernels and applies the boxes that touch its own planes (xb_edge_check_local / _global)."""
        with _Phase(self, 'label_halo'):
            self.be.sync()
            self.comm.exchange_planes(self.be, 0, self.sends, self.recvs)
            self.comm.exchange_planes(self.be, 1, self.sends, self.recvs)     # the flags the retraces rewrote
        with _Phase(self, 'edge_check'):
            idx, cls = self.be.edge_check_local()
            # (one int64 per changed voxel: index | class << 32 -- through the device transport when there is one)
            packed = np.asarray(idx, np.int64) | (np.asarray(cls, np.int64) << 32)
            allp = self._gather_rows(packed.reshape(-1, 1)).reshape(-1)
            gidx = allp & 0xffffffff
            gcls = (allp >> 32).astype(np.int8)
            _, edges = self.be.edge_check_global(gidx, gcls)
        with _Phase(self, 'sums'):
            edges, = self.comm.sum(edges)
        return edges

    def assign_refine(self, method, mode, iters):
        """bader_calc + refine back to back, as Bader.__call__ issues them -> (n_maxima, log).  One GPU: ONE library call, the
        refinement's first iteration queued behind the assignment (xb_assign_refine: one host wait for both); slabs: the two steps."""
        if self.comm.size == 1 and hasattr(self.be, 'assign_refine') and hasattr(self.be, 'ctx'):
            self.n_maxima, log = self.be.assign_refine(method, mode, iters)
            self.n_maxima = int(self.n_maxima)
            self._maxima = None
            return self.n_maxima, log
        n = self.assign(method)
        return n, self.refine(mode, iters)

    def refine(self, mode, iters):
        """thread_handlers.refine (thread_handlers.py:128-236) across slabs.  Returns [(edges, changed)]."""
        log = []
        if iters == 0:
            return log
        if self.comm.size == 1 and hasattr(self.be, 'refine') and hasattr(self.be, 'ctx'):
            return self.be.refine(mode, iters)      # one GPU: xb_refine (one host wait per iteration)
        stepped = getattr(self, '_stepped', False) and self._device_step()
        if stepped:
            edges, changed = self._refine_pass_device()
        else:
            self.exchange_label_halo()
            if hasattr(self.be, 'prepare_refine') and not getattr(self, 'windowed', False):
                with _Phase(self, 'table_build'):
                    self.be.prepare_refine()     # the retraces need the table anyway; edge_find profits from it
            with _Phase(self, 'edge_find'):
                edges = self.be.edge_find()
            # (the edge count travels with the retrace counters: one collective; without an edge anywhere the pass is empty)
            changed, edges = self._trace(edges)
        if edges == 0:
            return log
        log.append((edges, changed))
        it = 2
        while iters < 0 or it <= iters:
            if mode.lower() == 'all':
                if stepped:
                    edges, changed = self._refine_pass_device()
                else:
                    self.exchange_label_halo()
                    edges = self.be.edge_find()
                    changed, edges = self._trace(edges)
            else:
                if self.comm.size == 1:
                    _, edges = self.be.edge_check()
                    changed = self._trace()
                elif changed == 0:
                    # no voxel is flagged -2: edge_check is the identity (refinement.py:425-427) and the retrace has no work
                    edges, changed = 0, 0
                else:
                    edges = self._edge_check_slabs()
                    changed = self._trace()
            log.append((edges, changed))
            if changed == 0:
                break
            it += 1
        return log
