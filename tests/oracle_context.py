"""Test infrastructure: a stand-in for pybader_amd._lib.Context whose "device" is the CPU oracle.

The reference cannot travel to the GPU box and this container has no GPU, so the REAL pybader.interface.Bader can never be
run against libbader_hip.so.  What can be checked here is everything between the two: the INTEGRATION.md binding and the
Python drop-in layer (pybader_amd.thread_handlers / utils: argument order, in-place semantics, dtypes, residency tokens)
under the real class, with this context in place of the GPU one (tests/golden/check_real_class_binding.py).  Each method
restates what the corresponding xb_* entry point is specified to do (include/bader_hip.h), on host arrays."""
import numpy as np

import oracle
from rough_common import own_map, rank_labels


class OracleContext:
    def __init__(self):
        self.shape = None
        self.pinned_density = self.resident_density = self.resident_labels = None
        self._labels_host = None
        self.n_maxima = 0
        self.calls = []

    # -- the residency protocol of _lib.Context -------------------------------------------------
    def drop_label_token(self):
        a = self._labels_host
        if a is not None and getattr(self, '_labels_was_writeable', False):
            a.flags.writeable = True
        self._labels_host = None
        self.resident_labels = None

    def set_grid(self, shape, dist_mat, T_grad, x_range=None):
        shape = tuple(int(s) for s in shape)
        if shape != self.shape:
            self.rho = None
            self.labels = np.zeros(shape, np.int32)
        self.shape = shape
        if np.any(dist_mat):
            self.dm, self.tg = np.array(dist_mat, np.float64).reshape(3, 3, 3), np.array(T_grad, np.float64).reshape(3, 3)

    def upload_density(self, rho):
        self.calls.append('upload_density')
        assert rho.shape == self.shape
        self.resident_density = None
        self.rho = np.array(rho, np.float64, order='C')

    def upload_labels(self, labels):
        self.calls.append('upload_labels')
        self.drop_label_token()
        assert labels.shape == self.shape
        self.labels = np.array(labels, np.int32, order='C')

    def download_labels(self, dtype=np.int32, out=None, pooled=False):   # (pooled: page-locked result arrays -- the GPU context's business)
        self.calls.append('download_labels')
        if out is None:
            out = np.empty(self.shape, dtype)
        out.flags.writeable or out.setflags(write=True)
        out[...] = self.labels
        return out

    def sync(self):
        pass

    # -- the hot path ----------------------------------------------------------------------------
    def vacuum_assign(self, vac_tol, voxel_volume):          # xb_vacuum_assign: labels = -1 where rho <= tol else 0
        self.drop_label_token()
        tol = float('nan') if vac_tol is None else float(vac_tol)
        self.labels = np.zeros(self.shape, np.int32)
        _, charge, volume = oracle.vacuum_assign(self.rho, self.labels, tol, self.rho, float(voxel_volume))
        return charge, volume

    def assign(self, method):                                 # xb_assign: ongrid map / neargrid own-trajectory map, numbered by first voxel
        self.drop_label_token()
        vol0 = np.where(self.labels == -1, -1, 0).astype(np.int32)
        if method == 'ongrid':
            bmax, lab = oracle.bader_calc('ongrid', self.rho, vol0, self.dm, self.tg, 1)
            self.labels = lab.astype(np.int32)
            self._maxima = np.asarray(bmax, np.int64)
        else:
            lab, maxima = rank_labels(own_map(self.rho, vol0, self.dm, self.tg, main_ties=True))
            self.labels = lab.astype(np.int32)
            self._maxima = np.array(np.unravel_index(maxima, self.shape), np.int64).T.reshape(-1, 3)
        self.n_maxima = int(self._maxima.shape[0])
        return self.n_maxima

    def maxima(self):
        return self._maxima.copy()

    def refine(self, mode, iters):                            # xb_refine: thread_handlers.refine's loop, log of (edges, changed)
        self.drop_label_token()
        log = []
        oracle.refine('neargrid', (mode, int(iters)), self.rho, self.labels, self.dm, self.tg, 1, log=log)
        return [tuple(int(v) for v in row) for row in log]

    def assign_refine(self, method, mode, iters):             # xb_assign_refine: the two calls, one after the other
        n = self.assign(method)
        return n, self.refine(mode, iters)

    def charge_sum(self, voxel_volume, n_labels):             # xb_charge_sum: per-label sums, charge already times voxel_volume
        ch, vo = np.zeros(n_labels), np.zeros(n_labels)
        oracle.charge_sum(ch, vo, float(voxel_volume), self.rho, self.labels)
        return ch, vo

    def volume_assign(self, swap):                            # xb_volume_assign
        self.drop_label_token()
        sw = np.asarray(swap, np.int64)
        self.labels = np.where(self.labels >= 0, sw[np.maximum(self.labels, 0)], self.labels).astype(np.int32)

    def label_sum(self, value):                               # xb_label_sum
        m = self.labels == int(value)
        return float(self.rho[m].sum()), int(m.sum())

    def volume_mask(self, vol_num):                           # xb_volume_mask (utils.py:461-476)
        return np.where(self.labels == int(vol_num), self.rho, 0.0)

    def surface_distance(self, lattice, atoms_cart):          # xb_surface_distance: min squared distance per atom, edge count
        """thread_handlers.surface_distance / utils.surface_dist (thread_handlers.py:239-297, utils.py:320-379): edge voxels
        of the atom map, each measured against ITS atom over the 27 periodic images"""
        known = np.zeros(self.shape, np.int8)
        edges = oracle.edge_find(known, self.rho, self.labels)
        lattice = np.asarray(lattice, np.float64).reshape(3, 3)
        atoms = np.asarray(atoms_cart, np.float64).reshape(-1, 3)
        d2 = np.full(atoms.shape[0], np.inf)
        idx = np.argwhere(known == -2)
        if idx.size:
            vol = self.labels[tuple(idx.T)]
            pc = (idx / np.array(self.shape, np.float64)) @ lattice
            images = np.array([[x, y, z] for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1)], np.float64) @ lattice
            best = np.full(idx.shape[0], np.inf)
            for im in images:
                diff = pc - (atoms[vol] + im)
                best = np.minimum(best, (diff ** 2).sum(1))
            np.minimum.at(d2, vol, best)
        return d2, int(edges)
