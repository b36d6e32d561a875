"""Worker of tests/test_slab_gloo.py: one rank of a world_size-N gloo group driving the product's
slab scheduler (pybader_amd/slab.py) with a HOST backend built on the CPU oracle.

The backend mimics libbader_hip's slab contract: full-size label / known arrays of which only the
owned planes + halo are valid -- everything else is poisoned, so a scheduler that exchanges the wrong
planes (or too few) produces a different map than the single-rank run."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

POISON = -7


class OracleSlabBackend:
    def __init__(self, rho):
        self.rho = rho

    def set_grid(self, shape, dist_mat, T_grad, x_range, halo):
        import torch
        self.shape, self.dm, self.tg = tuple(shape), dist_mat, T_grad
        self.x0, self.x1 = x_range
        self.halo = halo
        self.whole = (self.x1 - self.x0) == shape[0]
        self.labels = np.zeros(shape, np.int32)
        self.known = np.zeros(shape, np.int8)
        nx = shape[0]
        self._t = (torch.from_numpy(self.labels.reshape(nx, -1)), torch.from_numpy(self.known.reshape(nx, -1)))

    def tensors(self):
        return self._t

    def planes(self, which):
        """(nx, plane) numpy view of the label (0) / known (1) array (HostComm's transport)"""
        return (self.labels if which == 0 else self.known).reshape(self.shape[0], -1)

    def sync(self):
        pass

    def _planes(self, ext):
        nx = self.shape[0]
        if self.whole or (self.x1 - self.x0) + 2 * ext >= nx:
            return np.arange(nx)
        return np.arange(self.x0 - ext, self.x1 + ext) % nx

    def _poisoned(self, arr, ext, poison):
        out = np.full_like(arr, poison)
        pl = self._planes(ext)
        out[pl] = arr[pl]
        return out

    def vacuum_assign(self, tol):
        import oracle
        oracle.vacuum_assign(self.rho, self.labels, float('nan') if tol is None else tol, self.rho, 1.0)

    def assign_trace(self, method):
        import oracle
        vol0 = self.labels.copy()
        if method == 'neargrid':
            F = oracle.own_trajectory_map(self.rho, vol0, self.dm, self.tg)
        else:
            bmax, main = oracle.bader_calc('ongrid', self.rho, vol0, self.dm, self.tg, 1)
            lin = np.ravel_multi_index(tuple(bmax.T), self.shape)
            F = np.where(main >= 0, lin[np.maximum(main, 0)], -1)
        own = F[self.x0:self.x1].reshape(-1)
        base = self.x0 * self.shape[1] * self.shape[2]
        self._own_F = own
        nv = own >= 0
        maxima, first = np.unique(own[nv], return_index=True)
        return maxima.astype(np.int64), (np.flatnonzero(nv)[first] + base).astype(np.int64)

    def assign_finish(self, maxima_sorted):
        rank = {int(m): k for k, m in enumerate(np.asarray(maxima_sorted).tolist())}
        own = np.array([rank[int(m)] if m >= 0 else -1 for m in self._own_F], np.int32)
        self.labels[...] = POISON if not self.whole else self.labels
        self.labels[self.x0:self.x1] = own.reshape((self.x1 - self.x0,) + self.shape[1:])

    def edge_find(self):
        import oracle
        lab = self._poisoned(self.labels, self.halo, POISON)
        known = np.zeros(self.shape, np.int8)
        oracle.edge_find(known, self.rho, lab)
        # outside the valid range: poisoned labels flagged "known", so a trace that strays there
        # terminates on a poisoned label (a visible mismatch) instead of walking on garbage
        nx = self.shape[0]
        valid = np.arange(nx) if (self.x1 - self.x0) + 2 * self.halo >= nx else self._planes(self.halo - 2)
        self.known[...] = 2
        self.known[valid] = known[valid]
        self.valid = valid
        return int((known[self.x0:self.x1] == -2).sum())

    def set_halo(self, halo):
        self.halo_now = halo

    def _retrace(self, flag):
        import oracle
        known = self.known.copy()
        mask = np.zeros(self.shape, bool)
        mask[self.x0:self.x1] = True
        start = (known == flag) & mask
        known[known == -2] = -1                     # only the owned voxels flagged `flag` are retraced
        known[known == -6] = -1
        known[start] = -2
        lab = self.labels.copy()
        nx = self.shape[0]
        if getattr(self, 'halo_now', self.halo) < nx:          # not in the all-valid fallback
            bad = np.ones(nx, bool)
            bad[self.valid] = False
            lab[bad] = POISON
            known[bad] = 2
        oracle.refine_neargrid(known, known.copy(), self.rho, lab, self.dm, self.tg)
        esc = start & (lab == POISON)               # the trace ended on a poisoned (invalid) plane
        ok = start & ~esc
        changed = int((ok & (lab != self.labels)).sum())
        self.labels[ok] = lab[ok]
        self.known[ok] = known[ok]
        self.known[esc] = -6
        return changed, int(esc.sum())

    def refine_trace(self):
        return self._retrace(-2)

    def escaped_paths(self, max_len=1 << 15):
        import oracle
        mask = np.zeros(self.shape, bool)
        mask[self.x0:self.x1] = True
        starts = np.flatnonzero((self.known == -6) & mask).astype(np.int64)
        nyz = self.shape[1] * self.shape[2]
        valid = np.zeros(self.shape[0], bool)
        valid[self.valid] = True
        paths, complete = [], []
        for v in starts:
            full = oracle.trajectory_path(self.rho, self.dm, self.tg, v)
            complete.append(full.size <= max_len)
            p = full[:max_len]
            out = np.flatnonzero(~valid[p // nyz])
            out = out[out >= 1]
            first = out[0] if out.size else p.size
            paths.append(np.concatenate([p[:1], p[first:]]))
        offsets = np.zeros(starts.size + 1, np.int64)
        offsets[1:] = np.cumsum([p.size for p in paths])
        vox = np.concatenate(paths) if paths else np.zeros(0, np.int64)
        return starts, offsets, vox, np.array(complete, bool)

    def gather_voxels(self, idx):
        # answers come from OWNED planes only: anything else would hide a scheduler that asks the wrong rank
        planes = np.asarray(idx) // (self.shape[1] * self.shape[2])
        assert np.all((planes >= self.x0) & (planes < self.x1))
        return self.labels.reshape(-1)[idx].copy(), self.known.reshape(-1)[idx].copy()

    def scatter_voxels(self, idx, labels, known):
        self.labels.reshape(-1)[idx] = labels
        self.known.reshape(-1)[idx] = known

    def edge_check(self):
        import oracle
        return oracle.edge_check(self.known, self.rho, self.labels)

    # 'changed' refinement across slabs.  The product resolves the all-gathered list of changed voxels on every rank
    # (xb_edge_check_local / _global); this host stand-in checks the scheduler's choreography: it assembles the
    # global arrays from every rank's owned planes (test-only shortcut through `self.comm`), runs the oracle's
    # sequential edge_check on them and keeps the planes a slab may rely on -- everything else is poisoned again.
    def edge_check_local(self):
        mask = np.zeros(self.shape, bool)
        mask[self.x0:self.x1] = True
        idx = np.flatnonzero((self.known == -2) & mask).astype(np.int64)
        return idx, np.zeros(idx.size, np.int8)

    def edge_check_global(self, gidx, gcls):
        import oracle
        parts = self.comm.allgather((self.x0, self.x1, self.labels[self.x0:self.x1].copy(), self.known[self.x0:self.x1].copy()))
        lab = np.concatenate([p[2] for p in sorted(parts, key=lambda p: p[0])])
        kn = np.concatenate([p[3] for p in sorted(parts, key=lambda p: p[0])])
        assert sorted(np.flatnonzero(kn == -2).tolist()) == sorted(np.asarray(gidx).tolist())
        # the halo planes this rank holds must already equal their owners' (label + known halo refreshed before)
        ok = self._planes(self.halo - 2)
        assert np.array_equal(self.labels[ok], lab[ok]) and np.array_equal(self.known[ok] == -2, kn[ok] == -2)
        checked, _ = oracle.edge_check(kn, self.rho, lab)
        self.known[...] = 2
        self.known[self.valid] = kn[self.valid]
        return checked, int((kn[self.x0:self.x1] == -2).sum())


def main():
    import torch.distributed as dist
    from pybader_amd import slab
    from conftest import case_density, load_golden
    case, method, mode, iters, out = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    halo = int(sys.argv[6]) if len(sys.argv) > 6 else 4
    transport = sys.argv[7] if len(sys.argv) > 7 else 'gloo'
    if transport == 'gloo':
        dist.init_process_group('gloo')
        from torch_comm import TorchComm
        comm = TorchComm(dist)
    else:                                   # the product's own host transport (pybader_amd/comm.py), no torch
        from pybader_amd import comm as xcomm
        store = xcomm.SocketStore()
        comm = xcomm.HostComm(store)
    g = load_golden(case)
    rho = case_density(g)
    be = OracleSlabBackend(rho)
    be.comm = comm
    runner = slab.SlabRunner(be, comm, rho.shape, g['dist_mat'], g['T_grad'], halo=halo)
    tol = float(g['vacuum_tol'])
    be.vacuum_assign(None if np.isnan(tol) else tol)
    n = runner.assign(method)
    pre = be.labels[be.x0:be.x1].copy()
    log = runner.refine(mode, iters)
    parts = comm.allgather((be.x0, be.x1, pre, be.labels[be.x0:be.x1].copy()))
    if comm.rank == 0:
        full_pre = np.concatenate([p[2] for p in sorted(parts, key=lambda p: p[0])])
        full = np.concatenate([p[3] for p in sorted(parts, key=lambda p: p[0])])
        np.savez(out, pre=full_pre, post=full, n=n, log=np.array(log, np.int64).reshape(-1, 2),
                 maxima=np.asarray(runner.maxima), fallbacks=runner.n_fallbacks)
    if transport == 'gloo':
        dist.destroy_process_group()
    else:
        comm.barrier()
        store.close()


if __name__ == '__main__':
    main()
