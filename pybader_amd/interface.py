"""A minimal counterpart of pybader.interface.Bader for the hot path (interface.py:105-631).

Not a re-implementation of the reference's class: no config file, no I/O, no pandas report.  It
exposes the attribute and method names the hot path touches, so the parity tests read like the
reference's own call sequence (examples/bader.py:15-21) and so INTEGRATION.md can show the
one-line swap inside the real class.  The small host-side matrices are computed with the same
numpy operations, in the same order, as the reference (they are inputs of every kernel and must be
bit-identical)."""
import numpy as np

from .thread_handlers import assign_to_atoms, bader_calc, bader_calc_refine, dtype_calc, refine, surface_distance
from .utils import charge_sum, resident, vacuum_assign


def distance_matrix(voxel_lattice):
    """Bader.distance_matrix (interface.py:242-259): d[i,j,k] = 1/|i*a + j*b + k*c| over i,j,k in
    (0, +1, -1) -- index 2 is the step -1 -- and d[0,0,0] = 0."""
    vl = np.asarray(voxel_lattice, dtype=np.float64)
    sign = (0.0, 1.0, -1.0)
    step = np.zeros((3, 3, 3, 3), dtype=np.float64)
    for axis in range(3):               # a, then b, then c: the reference's accumulation order
        for s in (1, 2):
            sel = [slice(None)] * 3
            sel[axis] = s
            step[tuple(sel)] += sign[s] * vl[axis]
    norm2 = np.sum(step**2, axis=3)
    out = norm2.copy()
    nz = norm2 != 0
    out[nz] = norm2[nz]**-.5
    return out


def gradient_transform(voxel_lattice):
    """Bader.T_grad (interface.py:285-290): (L^-1)^T . L^-1 with L the voxel lattice."""
    inv_l = np.linalg.inv(np.asarray(voxel_lattice, dtype=np.float64))
    return np.matmul(inv_l.T, inv_l)


class Bader:
    """Hot-path subset of pybader.interface.Bader: same constructor arguments, same step methods
    (volumes_init, bader_calc, refine_volumes, sum_volumes, bader_to_atom_distance, __call__)."""

    def __init__(self, density_dict, lattice, atoms, file_info=None, **kwargs):
        self._density = density_dict
        self._lattice = np.asarray(lattice, dtype=np.float64)
        self._atoms = np.ascontiguousarray(np.asarray(atoms, dtype=np.float64).reshape(-1, 3))
        self._file_info = file_info or {'voxel_offset': np.zeros(3)}
        self.density = self.charge if self.charge is not None else self.spin
        self.reference = self.density
        # DEFAULT profile (entry_points.py:326-339 / README.md:15-29)
        self.method = 'neargrid'
        self.refine_method = 'neargrid'
        self.vacuum_tol = None
        self.refine_mode = ('changed', 2)
        self.threads = 1
        self.speed_flag = False
        self.spin_flag = False
        self.export_mode = None
        self.fortran_format = 0
        for k, v in kwargs.items():
            setattr(self, k, v)

    # -- properties with the reference's names ----------------------------------------------
    @property
    def info(self):
        return self._file_info

    @property
    def charge(self):
        return self._density.get('charge', None)

    @property
    def spin(self):
        return self._density.get('spin', None)

    @property
    def lattice(self):
        return self._lattice

    @property
    def atoms(self):
        return self._atoms

    @property
    def lattice_volume(self):
        return np.abs(np.dot(self.lattice[0], np.cross(*self.lattice[1:])))       # interface.py:235-240

    @property
    def voxel_lattice(self):
        return np.divide(self.lattice, self.density.shape)                         # interface.py:261-265

    @property
    def voxel_volume(self):
        return self.lattice_volume / np.prod(self.density.shape)                   # interface.py:267-271

    @property
    def voxel_offset_fractional(self):
        return self.info['voxel_offset']

    @property
    def distance_matrix(self):
        return distance_matrix(self.voxel_lattice)

    @property
    def T_grad(self):
        return gradient_transform(self.voxel_lattice)

    @property
    def bader_maxima_fractional(self):
        return self._bader_maxima

    @property
    def bader_maxima(self):
        return np.dot(self._bader_maxima, self.lattice)                            # interface.py:312-316

    @bader_maxima.setter
    def bader_maxima(self, maxima):                                                # interface.py:318-324
        maxima = np.add(maxima, self.voxel_offset_fractional)
        self._bader_maxima = np.ascontiguousarray(np.divide(maxima, self.density.shape))

    # -- the step methods ---------------------------------------------------------------------
    def volumes_init(self, volumes=None):
        """interface.py:449-469."""
        if volumes is None:
            volumes = np.zeros(self.density.shape, dtype=dtype_calc(-np.prod(self.density.shape)))
        tol = np.float64('nan') if self.vacuum_tol is None else np.float64(self.vacuum_tol)
        volumes, self.vacuum_charge, self.vacuum_volume = vacuum_assign(
            self.reference, volumes, tol, self.density, self.voxel_volume)
        self.bader_volumes = volumes

    def bader_calc(self):
        """interface.py:471-477."""
        self.bader_maxima, self.bader_volumes = bader_calc(
            self.method, self.reference, self.bader_volumes, self.distance_matrix, self.T_grad, self.threads)

    def refine_volumes(self, volumes):
        """interface.py:486-490."""
        refine(self.refine_method, self.refine_mode, self.reference, volumes,
               self.distance_matrix, self.T_grad, self.threads)

    def bader_to_atom_distance(self):
        """interface.py:479-484."""
        self.bader_atoms, self.bader_distance, self.atoms_volumes = assign_to_atoms(
            self.bader_maxima, self.atoms, self.lattice, self.bader_volumes, self.threads)

    @property
    def voxel_offset(self):
        return np.dot(self.voxel_offset_fractional, self.voxel_lattice)                # interface.py:273-277

    def min_surface_distance(self):
        """interface.py:527-534."""
        atoms = self.atoms - self.voxel_offset
        self.atoms_surface_distance = surface_distance(self.reference, self.atoms_volumes, self.lattice, atoms,
                                                       self.threads)

    @property
    def spin_bool(self):
        """interface.py:215-221: the spin density is summed as well when it exists and spin_flag is set."""
        return bool(self.spin_flag) if self.spin is not None else False

    def sum_volumes(self, bader=False):
        """interface.py:492-525: charge (and, with spin_bool, spin) and volume per Bader volume or per atom;
        like the reference the volume array is summed again in the spin pass."""
        if bader:
            n = self.bader_maxima.shape[0]
            self.bader_charge, self.bader_volume = np.zeros(n), np.zeros(n)
            charge_sum(self.bader_charge, self.bader_volume, self.voxel_volume, self.density, self.bader_volumes)
            if self.spin_bool:
                self.bader_spin, self.bader_volume = np.zeros(n), np.zeros(n)
                charge_sum(self.bader_spin, self.bader_volume, self.voxel_volume, self.spin, self.bader_volumes)
        else:
            n = self.atoms.shape[0]
            self.atoms_charge, self.atoms_volume = np.zeros(n), np.zeros(n)
            charge_sum(self.atoms_charge, self.atoms_volume, self.voxel_volume, self.density, self.atoms_volumes)
            if self.spin_bool:
                self.atoms_spin, self.atoms_volume = np.zeros(n), np.zeros(n)
                charge_sum(self.atoms_spin, self.atoms_volume, self.voxel_volume, self.spin, self.atoms_volumes)

    def __call__(self, **kwargs):
        """The compute part of Bader.__call__ (interface.py:399-416); export and the pickle/dat output stay
        with the reference class (out of the hot path)."""
        for k, v in kwargs.items():
            setattr(self, k, v)
        ref = self.reference
        if isinstance(ref, np.ndarray) and ref.dtype == np.float64 and ref.flags.c_contiguous:
            with resident(ref):          # held still (and read-only) for the run: uploaded once, not per call
                self._run()
        else:
            self._run()

    def bader_calc_refine(self):
        """bader_calc() + refine_volumes(self.bader_volumes) (interface.py:406-409) in one library call
        (thread_handlers.bader_calc_refine): what __call__ runs when it owns both steps."""
        self.bader_maxima, self.bader_volumes = bader_calc_refine(
            self.method, self.refine_method, self.refine_mode, self.reference, self.bader_volumes,
            self.distance_matrix, self.T_grad, self.threads)

    fused = True      # _run issues bader_calc + refine as one call where the two are adjacent (False: the reference's two calls)

    def _run(self):
        self.volumes_init()
        if not self.speed_flag and self.fused:
            self.bader_calc_refine()
            self.sum_volumes(bader=True)
        else:
            self.bader_calc()
        if not self.speed_flag and not self.fused:
            self.refine_volumes(self.bader_volumes)
            self.sum_volumes(bader=True)
        self.bader_to_atom_distance()
        if self.speed_flag:
            self.refine_volumes(self.atoms_volumes)
            del self.bader_volumes
        self.min_surface_distance()
        self.sum_volumes()
        self.export_volumes()

    def export_volumes(self):
        """The export loop of Bader.__call__ (interface.py:417-436): `export_mode` = ('volumes' | 'atoms', [numbers]),
        [-2] meaning every volume / atom (and the vacuum when a tolerance is set)."""
        if self.export_mode is None:
            return
        kind, which = self.export_mode[0], list(self.export_mode[1])
        if kind not in ('volumes', 'atoms'):
            return
        if which[0] == -2:
            which = list(range(self.bader_maxima.shape[0] if kind == 'volumes' else self.atoms.shape[0]))
            if self.vacuum_tol is not None:
                which.append(-1)
        for vol_num in which:
            self.write_volume(vol_num)

    def write_volume(self, vol_num):
        """Bader.write_volume (interface.py:600-621): the charge (and spin) density of one Bader volume or atom, zero
        elsewhere (utils.volume_mask on the GPU), handed to the file type's own writer (`info['write_function']`,
        pybader.io untouched)."""
        from .utils import volume_mask
        density = {}
        volumes = self.bader_volumes if self.export_mode[0] == 'volumes' else self.atoms_volumes
        if self.charge is not None:
            density['charge'] = volume_mask(volumes, self.charge, vol_num)
        if self.spin is not None:
            density['spin'] = volume_mask(volumes, self.spin, vol_num)
        num = vol_num if vol_num != -1 else 'vacuum'
        self._file_info['comment'] = f"Bader {self.export_mode[0]}: {num}\n"
        self._file_info['fortran_format'] = self.fortran_format
        self.info['write_function'](f"Bader-{self.export_mode[0]}-{num}", self.atoms, self.lattice, density, self.info,
                                    prefix=self.info['prefix'])
