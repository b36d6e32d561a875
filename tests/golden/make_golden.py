#!/opt/conda/bin/python3.9
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE (pybader v0.3.12).

Runs ONLY in the build container (needs /root/reference and the conda python3.9 + numba 0.54.1):

    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden.py [case ...]

Nothing of the reference is copied: this script drives the reference's own functions
(`Bader.volumes_init/bader_calc/refine_volumes/...`, `refinement.edge_find/neargrid/edge_check`)
on synthetic densities from pybader_amd/synth.py and stores inputs' parameters + outputs as
compressed .npz files (data only).  The numba import shim is the one documented in SURVEY.md A.2.
"""
import hashlib
import io
import json
import os
import sys
import tempfile
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, '/root/reference')

# ---- environment the reference needs -------------------------------
SCRATCH = tempfile.mkdtemp(prefix='golden_')
os.makedirs(os.path.join(SCRATCH, '.config', 'bader'))
with open(os.path.join(SCRATCH, '.config', 'bader', 'config.ini'), 'w') as f:
    f.write("[DEFAULT]\nmethod = neargrid\nrefine_method = neargrid\nvacuum_tol = None\n"
            "refine_mode = ('changed', 2)\nbader_volume_tol = 0.001\nexport_mode = None\nprefix = ''\n"
            "output = pickle\nthreads = 1\nfortran_format = 0\nspeed_flag = False\nspin_flag = False\n\n"
            "[speed]\nmethod = ongrid\nrefine_method = neargrid\nrefine_mode = ('changed', 3)\nspeed_flag = True\n")
os.environ['HOME'] = SCRATCH
os.environ.setdefault('NUMBA_CACHE_DIR', '/tmp/golden_nbcache')
sys.dont_write_bytecode = True

# ---- numba 0.54.1 vs numpy 1.26 import shim ------------------------
m = types.ModuleType('numba.np.ufunc._internal')
m.PyUFunc_None, m.PyUFunc_Zero, m.PyUFunc_One, m.PyUFunc_ReorderableNone = -1, 0, 1, -2
m._DUFunc = type('_DUFunc', (), {})
m.fromfunc = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
sys.modules['numba.np.ufunc._internal'] = m
np.MachAr = type('MachAr', (), {})
_v = np.__version__
np.__version__ = '1.20.3'
import numba  # noqa: E402
np.__version__ = _v

from pybader import refinement, thread_handlers  # noqa: E402
from pybader.interface import Bader  # noqa: E402
from pybader.utils import nostdout  # noqa: E402

from pybader_amd import synth  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_bader(rho, lattice, atoms_cart, **kw):
    info = {'filename': 'synth', 'prefix': '', 'file_type': 'synthetic', 'write_function': None,
            'voxel_offset': np.zeros(3), 'out_dest': os.devnull}
    return Bader({'charge': rho}, lattice, atoms_cart, info, threads=1, **kw)


def ref_refine_logged(b, volumes, mode):
    """thread_handlers.refine (thread_handlers.py:128-236) for threads=1, restated as direct calls of
    the reference kernels so the per-iteration (edges, changed) counts and `known` can be captured."""
    rho, dm, tg = b.reference, b.distance_matrix, b.T_grad
    check_mode, iters = mode
    idx = np.zeros(3, np.int64)
    log = []
    known = np.zeros(rho.shape, dtype=np.int8)
    edges = refinement.edge_find(known, rho, volumes)
    known0 = known.copy()
    if edges == 0 or iters == 0:
        return log, known0, known
    known, changed = refinement.neargrid(known, known.copy(), rho, volumes, idx, dm, tg, np.zeros(1, np.int64))
    log.append((edges, changed))
    if iters < 0:
        iters = float('inf')
    it = 2
    while it <= iters:
        if check_mode == 'all':
            known = np.zeros(rho.shape, dtype=np.int8)
            edges = refinement.edge_find(known, rho, volumes)
        else:
            _, edges = refinement.edge_check(known, rho, volumes)
        known, changed = refinement.neargrid(known, known.copy(), rho, volumes, idx, dm, tg, np.zeros(1, np.int64))
        log.append((edges, changed))
        if changed == 0:
            break
        it += 1
    return log, known0, known


def own_trajectory(b, main):
    """Map F from the reference's own refinement kernel."""
    v = main.copy()
    known = np.where(v == -1, 0, -2).astype(np.int8)
    refinement.neargrid(known, np.zeros_like(known), b.reference, v, np.zeros(3, np.int64),
                        b.distance_matrix, b.T_grad, np.zeros(1, np.int64))
    return v


def run_default_vs_converged(name, shape, lattice, atoms=synth.ATOMS8, vacuum_tol=None, **_):
    """Round 5: where the reference's default ('changed', 2) stops unconverged (1024^3: its log ends
    [377, 4]) the voxels on which its map differs from its own converged ('changed', -1) map are stored -- linear
    indices, the reference's labels in both maps -- and ADDED to the existing fixture after checking that both maps
    hash to what the full run of `run_case` recorded.  Only the neargrid leg of run_case is repeated."""
    t0 = time.time()
    path = os.path.join(HERE, name + '.npz')
    out = dict(np.load(path))
    lattice = np.asarray(lattice, np.float64)
    rho = synth.synth_density(shape, lattice, atoms)
    assert sha(rho) == str(out['rho_sha256'])
    with nostdout():
        b = make_bader(rho, lattice, synth.atoms_cartesian(atoms, lattice), vacuum_tol=vacuum_tol)
        b.volumes_init()
        b.bader_calc()
        main = b.bader_volumes.copy()
        assert sha(main) == str(out['ng_main_sha256'])
        v2 = main.copy()
        thread_handlers.refine('neargrid', ('changed', 2), b.reference, v2, b.distance_matrix, b.T_grad, 1)
        assert sha(v2) == str(out['ng_changed_2_sha256'])
        vc = main
        thread_handlers.refine('neargrid', ('changed', -1), b.reference, vc, b.distance_matrix, b.T_grad, 1)
        assert sha(vc) == str(out['ng_changed_inf_sha256'])
    idx = np.flatnonzero(v2.ravel() != vc.ravel()).astype(np.int64)
    out['ng_changed_2_vs_inf_idx'] = idx
    out['ng_changed_2_vs_inf_default_labels'] = v2.ravel()[idx].astype(np.int64)
    out['ng_changed_2_vs_inf_converged_labels'] = vc.ravel()[idx].astype(np.int64)
    np.savez_compressed(path, **out)
    print(f"{name}: default vs converged: {idx.size} voxels differ {idx.tolist()[:16]}; {time.time() - t0:.1f}s", flush=True)


def run_case(name, shape, lattice, atoms=synth.ATOMS8, vacuum_tol=None, full_maps=True, do_F=True,
             modes=(('changed', 2), ('changed', -1), ('all', -1), ('all', 2))):
    t0 = time.time()
    lattice = np.asarray(lattice, np.float64)
    rho = synth.synth_density(shape, lattice, atoms)
    atoms_cart = synth.atoms_cartesian(atoms, lattice)
    out = {'shape': np.array(shape, np.int64), 'lattice': lattice, 'atoms': np.asarray(atoms, np.float64),
           'background': np.float64(synth.BACKGROUND), 'rho_sha256': np.array(sha(rho)),
           'vacuum_tol': np.float64(np.nan if vacuum_tol is None else vacuum_tol)}

    def keep(key, a):
        a = np.ascontiguousarray(a)
        out[key + '_sha256'] = np.array(sha(a))
        if full_maps or a.size < 100000:
            out[key] = a

    with nostdout():
        # ---------------- neargrid, default profile ----------------
        b = make_bader(rho, lattice, atoms_cart, vacuum_tol=vacuum_tol)
        out['dist_mat'] = b.distance_matrix
        out['T_grad'] = b.T_grad
        out['voxel_volume'] = np.float64(b.voxel_volume)
        b.volumes_init()
        out['vacuum_charge'] = np.float64(b.vacuum_charge)
        out['vacuum_volume'] = np.float64(b.vacuum_volume)
        keep('ng_init', b.bader_volumes.astype(np.int8))
        b.bader_calc()
        main = b.bader_volumes.copy()
        keep('ng_main', main)
        ng_max = np.rint(b.bader_maxima_fractional * np.array(shape)).astype(np.int64)
        out['ng_bader_max'] = ng_max
        if do_F:
            keep('ng_F', own_trajectory(b, main))
        first = True
        for mode in modes:
            tag = f"ng_{mode[0]}_{'inf' if mode[1] < 0 else mode[1]}"
            v = main.copy()
            log, known0, known_last = ref_refine_logged(b, v, mode)
            v2 = main.copy()
            thread_handlers.refine('neargrid', mode, b.reference, v2, b.distance_matrix, b.T_grad, 1)
            assert np.array_equal(v, v2), "restated refine driver != thread_handlers.refine"
            keep(tag, v)
            out[tag + '_log'] = np.array(log, np.int64).reshape(-1, 2)
            if first:
                keep('ng_known0', known0)
                first = False
            if mode == ('changed', 2):
                keep('ng_changed_2_known_last', known_last)
        # the rest of Bader.__call__ on the default ('changed', 2) result (interface.py:408-416)
        b.refine_mode = ('changed', 2)
        b.refine_volumes(b.bader_volumes)
        b.sum_volumes(bader=True)
        b.bader_to_atom_distance()
        b.min_surface_distance()
        b.sum_volumes()
        out['ng_bader_charge'] = b.bader_charge
        out['ng_bader_volume'] = b.bader_volume
        out['ng_bader_atoms'] = b.bader_atoms
        out['ng_bader_distance'] = b.bader_distance
        keep('ng_atoms_volumes', b.atoms_volumes)
        out['ng_atoms_charge'] = b.atoms_charge
        out['ng_atoms_volume'] = b.atoms_volume
        out['ng_atoms_surface_distance'] = b.atoms_surface_distance
        out['bader_maxima_cart'] = b.bader_maxima

        # ---------------- ongrid main pass, then the `speed` profile flow ----------------
        b = make_bader(rho, lattice, atoms_cart, vacuum_tol=vacuum_tol, method='ongrid')
        b.volumes_init()
        b.bader_calc()
        og_main = b.bader_volumes.copy()
        keep('og_main', og_main)
        out['og_bader_max'] = np.rint(b.bader_maxima_fractional * np.array(shape)).astype(np.int64)
        # BASELINE config 5: ongrid assign + neargrid edge refinement on the Bader volumes
        v = og_main.copy()
        log, _, _ = ref_refine_logged(b, v, ('changed', 2))
        keep('og_ngrefine_changed_2', v)
        out['og_ngrefine_changed_2_log'] = np.array(log, np.int64).reshape(-1, 2)
        # speed profile (entry_points.py:340-345): refine the *atom* map with ('changed', 3)
        b.bader_to_atom_distance()
        out['og_bader_atoms'] = b.bader_atoms
        out['og_bader_distance'] = b.bader_distance
        keep('og_atoms_volumes_pre', b.atoms_volumes.copy())
        b.refine_mode = ('changed', 3)
        b.refine_volumes(b.atoms_volumes)
        keep('og_atoms_volumes_speed', b.atoms_volumes)
        b.sum_volumes()
        out['og_atoms_charge_speed'] = b.atoms_charge
        out['og_atoms_volume_speed'] = b.atoms_volume

    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f"{name}: {time.time() - t0:.1f}s -> {os.path.getsize(path) / 1024:.0f} KiB; "
          f"ng maxima {ng_max.shape[0]}, og maxima {out['og_bader_max'].shape[0]}, "
          f"logs " + json.dumps({k: out[k].tolist() for k in out if k.endswith('_log')}), flush=True)


def tables():
    """Small pure-function tables: dtype_calc, factor_3d, distance_matrix / T_grad for 3 lattices."""
    from pybader.utils import dtype_calc, factor_3d
    out = {}
    vals = [0, 1, 127, 128, 255, 256, 32767, 32768, 65535, 65536, 2**31 - 1, 2**31, 2**32 - 1, 2**32]
    args = [s * v for v in vals for s in (1, -1)]
    out['dtype_calc_args'] = np.array(args, dtype=np.float64)
    out['dtype_calc_out'] = np.array([dtype_calc(int(a)) for a in args])
    out['factor_3d'] = np.array([factor_3d(i) for i in range(1, 65)], np.int64)
    lats = [synth.CUBIC6, synth.TRICLINIC, np.array([[4.1, 0.2, -0.3], [0.0, 7.7, 1.9], [-2.2, 0.4, 5.3]])]
    shapes = [(64, 64, 64), (40, 48, 56), (24, 36, 30)]
    for k, (lat, shp) in enumerate(zip(lats, shapes)):
        b = make_bader(np.ones(shp), np.asarray(lat, np.float64), np.zeros((1, 3)))
        out[f'lat{k}'] = np.asarray(lat, np.float64)
        out[f'shape{k}'] = np.array(shp, np.int64)
        out[f'dist_mat{k}'] = b.distance_matrix
        out[f'T_grad{k}'] = b.T_grad
        out[f'voxel_volume{k}'] = np.float64(b.voxel_volume)
    np.savez_compressed(os.path.join(HERE, 'tables.npz'), **out)
    print('tables done', flush=True)


def run_rough_case(name, shape, lattice, vacuum_tol=None, modes=(('changed', 2), ('changed', -1), ('all', -1)),
                   **rough):
    """Non-smooth densities: noise (thousands of maxima), 5-significant-digit rounding (what a
    CHG file holds: exact ties in the low-density regions), exact plateaus, vacuum where refinement changes
    voxels.  Stored: the reference's sequential main map, its final map + log per refine mode, its own-trajectory
    map (refinement.py stepping, strict ties), maxima lists, the ongrid map and its ('changed',2) refinement."""
    t0 = time.time()
    lattice = np.asarray(lattice, np.float64)
    rho = synth.rough_density(shape, lattice, synth.ATOMS8, **rough)
    atoms_cart = synth.atoms_cartesian(synth.ATOMS8, lattice)
    out = {'shape': np.array(shape, np.int64), 'lattice': lattice, 'rho_sha256': np.array(sha(rho)),
           'vacuum_tol': np.float64(np.nan if vacuum_tol is None else vacuum_tol),
           'rough_json': np.array(json.dumps(rough, sort_keys=True))}
    with nostdout():
        b = make_bader(rho, lattice, atoms_cart, vacuum_tol=vacuum_tol)
        out['dist_mat'] = b.distance_matrix
        out['T_grad'] = b.T_grad
        out['voxel_volume'] = np.float64(b.voxel_volume)
        b.volumes_init()
        out['ng_init'] = b.bader_volumes.astype(np.int8)
        b.bader_calc()
        main = b.bader_volumes.copy()
        out['ng_main'] = main
        out['ng_bader_max'] = np.rint(b.bader_maxima_fractional * np.array(shape)).astype(np.int64)
        out['ng_F'] = own_trajectory(b, main)
        for mode in modes:
            tag = f"ng_{mode[0]}_{'inf' if mode[1] < 0 else mode[1]}"
            v = main.copy()
            log, known0, known_last = ref_refine_logged(b, v, mode)
            v2 = main.copy()
            thread_handlers.refine('neargrid', mode, b.reference, v2, b.distance_matrix, b.T_grad, 1)
            assert np.array_equal(v, v2)
            out[tag] = v
            out[tag + '_log'] = np.array(log, np.int64).reshape(-1, 2)
            # the rest of Bader.__call__ on this mode's result (interface.py:408-416): what north_star gates on -- the
            # voxel -> atom map and the per-atom charges / volumes (independent of the basin numbering)
            b.bader_volumes = v.copy()
            b.sum_volumes(bader=True)
            b.bader_to_atom_distance()
            b.sum_volumes()
            out[tag + '_bader_charge'] = b.bader_charge
            out[tag + '_bader_volume'] = b.bader_volume
            out[tag + '_bader_atoms'] = b.bader_atoms
            out[tag + '_bader_distance'] = b.bader_distance
            out[tag + '_atoms_volumes'] = b.atoms_volumes
            out[tag + '_atoms_charge'] = b.atoms_charge
            out[tag + '_atoms_volume'] = b.atoms_volume
        out['atoms_cart'] = atoms_cart
        b = make_bader(rho, lattice, atoms_cart, vacuum_tol=vacuum_tol, method='ongrid')
        b.volumes_init()
        b.bader_calc()
        out['og_main'] = b.bader_volumes.copy()
        out['og_bader_max'] = np.rint(b.bader_maxima_fractional * np.array(shape)).astype(np.int64)
        v = out['og_main'].copy()
        log, _, _ = ref_refine_logged(b, v, ('changed', 2))
        out['og_ngrefine_changed_2'] = v
        out['og_ngrefine_changed_2_log'] = np.array(log, np.int64).reshape(-1, 2)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f"{name}: {time.time() - t0:.1f}s -> {os.path.getsize(path) / 1024:.0f} KiB; ng maxima "
          f"{out['ng_bader_max'].shape[0]}, og maxima {out['og_bader_max'].shape[0]}, logs "
          + json.dumps({k: out[k].tolist() for k in out if k.endswith('_log')}), flush=True)


def trajectory_vectors(name='traj_vectors', shape=(20, 18, 22), n_starts=400, n_steps=6):
    """G5: step-level vectors from the reference's own refinement kernel (strict tie test,
    refinement.py:111): refinement.neargrid with ONE voxel flagged -2, labels == linear index and rknown == 2
    everywhere except on the voxels already known to be on the path: the trace stops on the first voxel outside
    that set and hands its label (== its index) to the start voxel.  Growing the set one voxel at a time yields
    the first n_steps voxels of the trajectory, carried remainder included.
    (The non-strict variant of methods.py:324 cannot be isolated that way -- methods.neargrid has no rknown input
    and a lone path among vacuum voxels ends on a vacuum maximum and is labelled -1; it is pinned by the full
    main-pass maps of the rough cases below, where exact ties abound.)
    Densities: quantised + noisy (ties, plateaus, hundreds of maxima)."""
    t0 = time.time()
    lattice = synth.TRICLINIC
    out = {'shape': np.array(shape, np.int64), 'lattice': lattice}
    dens = {'q': dict(noise=0.3, seed=3, quantum=0.0625), 's': dict(noise=0.05, seed=5, sig_digits=3)}
    b0 = make_bader(np.ones(shape), lattice, np.zeros((1, 3)))
    dm, tg = b0.distance_matrix, b0.T_grad
    out['dist_mat'], out['T_grad'] = dm, tg
    N = int(np.prod(shape))
    idx0, ic = np.zeros(3, np.int64), np.zeros(1, np.int64)
    for key, kw in dens.items():
        rho = synth.rough_density(shape, lattice, synth.ATOMS8, **kw)
        out[key + '_rough_json'] = np.array(json.dumps(kw, sort_keys=True))
        out[key + '_rho_sha256'] = np.array(sha(rho))
        starts = (np.arange(n_starts, dtype=np.int64) * 7919 + 13) % N
        out[key + '_starts'] = starts
        lab0 = np.arange(N, dtype=np.int64).reshape(shape)
        strict = np.full((n_starts, n_steps), -1, np.int64)
        for t, s in enumerate(starts):
            on_path = [int(s)]
            for k in range(n_steps):
                known = np.zeros(shape, np.int8)
                rknown = np.full(shape, 2, np.int8)
                known.reshape(-1)[s] = -2
                rknown.reshape(-1)[on_path] = 0
                vol = lab0.copy()
                refinement.neargrid(known, rknown, rho, vol, idx0, dm, tg, ic)
                q = int(vol.reshape(-1)[s])
                if q == int(s) or q in on_path:      # ended on a maximum that is on the path already
                    strict[t, k] = q
                    break
                strict[t, k] = q
                on_path.append(q)
        out[key + '_strict_steps'] = strict
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f"{name}: {time.time() - t0:.1f}s -> {os.path.getsize(path) / 1024:.0f} KiB", flush=True)


def run_export_case(name='export_volumes'):
    """Bader.write_volume / the export loop of Bader.__call__ (interface.py:417-436, 600-621) with a capturing
    write_function: file names, comments and sha256 of the masked densities."""
    t0 = time.time()
    out = {}
    for cname, kw_b, mode in (('c40x48x56_tric', {}, ('volumes', [0, 3])), ('c40x48x56_tric', {}, ('atoms', [-2])),
                              ('c48_cubic_vac', {'vacuum_tol': 0.03}, ('volumes', [-2]))):
        kw = CASES[cname]
        lattice = np.asarray(kw['lattice'], np.float64)
        rho = synth.synth_density(kw['shape'], lattice, synth.ATOMS8)
        spin = np.ascontiguousarray(rho[::-1] * 0.25)
        calls = []

        def capture(fname, atoms, lat, density, info, prefix='', **k):
            calls.append((fname, info['comment'], info['fortran_format'], sha(density['charge']), sha(density['spin']),
                          float(density['charge'].sum())))
        info = {'filename': 'synth', 'prefix': '', 'file_type': 'synthetic', 'write_function': capture,
                'voxel_offset': np.zeros(3), 'out_dest': os.devnull}
        b = Bader({'charge': rho, 'spin': spin}, lattice, synth.atoms_cartesian(synth.ATOMS8, lattice), info, threads=1, **kw_b)
        b.export_mode = mode
        b.fortran_format = 2
        with nostdout():
            try:
                b()
            except AttributeError:       # to_file() cannot pickle the capturing writer; the export loop has run by then
                pass
        key = f'{cname}_{mode[0]}_{mode[1][0]}'
        out[key] = np.array(json.dumps(calls))
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f"{name}: {time.time() - t0:.1f}s -> {os.path.getsize(path) / 1024:.0f} KiB", flush=True)


def run_contract_case(name='pickle_contract'):
    """The on-disk contract of the reference (SURVEY.md 8(b) last row): Bader.__call__ with output='pickle' pickles the slotted
    object itself (interface.py:122-126, 593-598) and `bader-read` loads it back (entry_points.py:235-236).  Recorded per
    profile: every slot the pickled object carries -- its type and, for arrays, dtype and shape -- so that the drop-in's
    outputs can be held against what the real class would have stored.  Data only (names / dtypes / shapes)."""
    import pickle
    t0 = time.time()
    kw = CASES['c40x48x56_tric']
    lattice = np.asarray(kw['lattice'], np.float64)
    rho = synth.synth_density(kw['shape'], lattice, synth.ATOMS8)
    spin = np.ascontiguousarray(rho[::-1] * 0.25)
    atoms_cart = synth.atoms_cartesian(synth.ATOMS8, lattice)
    profiles = {'default': dict(), 'default_vacuum_spin': dict(vacuum_tol=0.03, spin_flag=True),
                'speed': dict(method='ongrid', refine_method='neargrid', refine_mode=('changed', 3), speed_flag=True)}
    out = {'shape': list(kw['shape']), 'n_atoms': int(atoms_cart.shape[0]), 'slots': list(Bader.__slots__), 'profiles': {}}
    for pname, conf in profiles.items():
        dest = os.path.join(SCRATCH, pname + '.p')
        info = {'filename': 'synth', 'prefix': '', 'file_type': 'synthetic', 'write_function': None,
                'voxel_offset': np.zeros(3), 'out_dest': dest}
        b = Bader({'charge': rho, 'spin': spin}, lattice, atoms_cart, info, threads=1, **conf)
        with nostdout():
            b()
        with open(dest, 'rb') as f:
            loaded = pickle.load(f)
        assert type(loaded).__module__ == 'pybader.interface' and type(loaded).__name__ == 'Bader'
        rec = {}
        for slot in Bader.__slots__:
            try:
                v = object.__getattribute__(loaded, slot)
            except AttributeError:
                continue
            if isinstance(v, np.ndarray):
                rec[slot] = {'type': 'ndarray', 'dtype': v.dtype.str, 'shape': list(v.shape), 'c_contiguous': bool(v.flags.c_contiguous)}
            elif isinstance(v, dict):
                rec[slot] = {'type': 'dict', 'keys': sorted(v.keys()),
                             'values': {k: (['ndarray', x.dtype.str, list(x.shape)] if isinstance(x, np.ndarray) else type(x).__name__)
                                        for k, x in v.items()}}
            else:
                rec[slot] = {'type': type(v).__name__, 'value': v if isinstance(v, (int, float, str, bool, type(None))) else repr(v)}
        out['profiles'][pname] = {'config': {k: (list(v) if isinstance(v, tuple) else v) for k, v in conf.items()}, 'slots_set': rec,
                                  'n_maxima': int(loaded._bader_maxima.shape[0])}
    path = os.path.join(HERE, name + '.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write('\n')
    print(f"{name}: {time.time() - t0:.1f}s -> {os.path.getsize(path) / 1024:.0f} KiB", flush=True)


def run_threads_case(name='threads_blocks'):
    """The reference's threads > 1 path (thread_handlers.py:15-75, 128-236): factor_3d block split, methods.neargrid
    per block with the block-extension branches, volume_offset / volume_merge / array_merge / edge_assign, then the
    threaded refinement.  The reference merges blocks in COMPLETION order, which makes its numbering depend on thread
    timing; here `as_completed` is replaced by submission order (a deterministic schedule of the same code), which is
    what oracle/bader_oracle_blocks.c restates."""
    t0 = time.time()
    thread_handlers.as_completed = lambda fs: list(fs)        # dict of futures -> submission order; .result() waits
    out = {}
    cases = {'c12_cubic': [2, 4, 8], 'c40x48x56_tric': [2, 3, 6, 8], 'c48_cubic_vac': [4, 12], 'r40_noise04': [5, 8]}
    out['cases_json'] = np.array(json.dumps(cases))
    with nostdout():
        for cname, tlist in cases.items():
            if cname in CASES:
                kw = CASES[cname]
                lattice = np.asarray(kw['lattice'], np.float64)
                rho = synth.synth_density(kw['shape'], lattice, synth.ATOMS8)
                tol = kw.get('vacuum_tol')
            else:
                kw = ROUGH[cname]
                lattice = np.asarray(kw['lattice'], np.float64)
                rho = synth.rough_density(kw['shape'], lattice, synth.ATOMS8, noise=kw.get('noise', 0.), sig_digits=kw.get('sig_digits'),
                                          quantum=kw.get('quantum'))
                tol = kw.get('vacuum_tol')
            atoms_cart = synth.atoms_cartesian(synth.ATOMS8, lattice)
            out[cname + '_rho_sha256'] = np.array(sha(rho))
            for t in tlist:
                b = make_bader(rho, lattice, atoms_cart, vacuum_tol=tol)
                b.threads = t
                b.volumes_init()
                b.bader_calc()
                key = f'{cname}_t{t}'
                out[key + '_main'] = b.bader_volumes.copy()
                out[key + '_max'] = np.rint(b.bader_maxima_fractional * np.array(rho.shape)).astype(np.int64)
                b.refine_mode = ('changed', 2)
                b.refine_volumes(b.bader_volumes)
                out[key + '_changed_2'] = b.bader_volumes.copy()
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f"{name}: {time.time() - t0:.1f}s -> {os.path.getsize(path) / 1024:.0f} KiB", flush=True)


CASES = {
    # G6: tiny hand-checkable
    'c12_cubic': dict(shape=(12, 12, 12), lattice=synth.CUBIC6),
    # G1: 64^3 cubic + one ragged triclinic grid
    'c64_cubic': dict(shape=(64, 64, 64), lattice=synth.CUBIC6),
    'c40x48x56_tric': dict(shape=(40, 48, 56), lattice=synth.TRICLINIC),
    # G3: vacuum (tolerance chosen so ~1/3 of the cell is vacuum for these atoms)
    'c48_cubic_vac': dict(shape=(48, 48, 48), lattice=synth.CUBIC6, vacuum_tol=0.03),
    # G4: larger grids, hashes only
    'c128_tric': dict(shape=(128, 128, 128), lattice=synth.TRICLINIC, full_maps=False,
                      modes=(('changed', 2), ('all', -1))),
    'c256_cubic': dict(shape=(256, 256, 256), lattice=synth.CUBIC6, full_maps=False, do_F=True,
                       modes=(('changed', 2),)),
    # round 4: the headline size itself (BASELINE configs 3 and 5), hashes only
    'c512_cubic': dict(shape=(512, 512, 512), lattice=synth.CUBIC6, full_maps=False, do_F=True,
                       modes=(('changed', 2),)),
    # round 5: a large NON-cubic pin -- the generic Grid instantiations (no mirror prefilter, full T_grad) above 128^3
    'c320_tric': dict(shape=(320, 320, 320), lattice=synth.TRICLINIC, full_maps=False, do_F=True,
                      modes=(('changed', 2),)),
    # round 6: many atoms -- 216 maxima, bricks with several of them, twice the dividing surface per voxel
    'c128_216atoms': dict(shape=(128, 128, 128), lattice=synth.CUBIC6, atoms=synth.atoms_jittered_grid(6, 5), full_maps=False,
                          modes=(('changed', 2), ('all', -1))),
    # (the reference's default two iterations do not converge here -- its log ends with 33 relabelled voxels -- so the converged
    # ('changed', -1) result is captured as well: that one is the own-trajectory map)
    'c1024_cubic': dict(shape=(1024, 1024, 1024), lattice=synth.CUBIC6, full_maps=False, do_F=False,
                        modes=(('changed', 2), ('changed', -1))),
}

ROUGH = {
    # noise: hundreds to thousands of maxima, no ties
    'r40_noise005': dict(shape=(40, 40, 40), lattice=synth.TRICLINIC, noise=0.05),
    'r40_noise04': dict(shape=(40, 40, 40), lattice=synth.TRICLINIC, noise=0.4),
    'r64_noise04': dict(shape=(64, 64, 64), lattice=synth.CUBIC6, noise=0.4, modes=(('changed', 2), ('all', -1))),
    # what a CHG file holds (%13.5E): exact ties wherever the density is low
    'r48_sig5': dict(shape=(48, 48, 48), lattice=synth.CUBIC6, sig_digits=5),
    'r48_sig5_noise': dict(shape=(48, 48, 48), lattice=synth.TRICLINIC, noise=0.02, sig_digits=5),
    # exact plateaus
    'r32_quant8': dict(shape=(32, 24, 24), lattice=synth.TRICLINIC, quantum=0.125),
    # vacuum where refinement changes voxels ('changed' mode carries the reference's vacuum bug)
    'r40_vac_noise': dict(shape=(40, 40, 40), lattice=synth.CUBIC6, noise=0.05, vacuum_tol=0.06),
    # round 3: the judge's two fresh inputs -- plateaus + noise + a vacuum tolerance (the default mode stops
    # while both the reference and this library are still changing voxels), 4 significant digits + strong noise -- and a
    # %13.5E-rounded density WITH a vacuum tolerance
    'r36_plateau_vac': dict(shape=(36, 36, 28), lattice=synth.TRICLINIC, noise=0.02, quantum=0.03125, vacuum_tol=0.05, seed=777),
    'r30_noise_sig4': dict(shape=(30, 26, 34), lattice=np.array([[4.1, 0.2, -0.3], [0.0, 7.7, 1.9], [-2.2, 0.4, 5.3]]), noise=0.1,
                           sig_digits=4, seed=12345),
    'r48_sig5_vac': dict(shape=(48, 48, 48), lattice=synth.CUBIC6, sig_digits=5, vacuum_tol=0.04),
}

if __name__ == '__main__':
    # warm the JIT on a tiny grid first
    which = sys.argv[1:] or ['tables', 'traj_vectors', 'threads_blocks', 'export_volumes', 'pickle_contract'] + list(CASES) + list(ROUGH)
    for name in which:
        if name == 'tables':
            tables()
        elif name == 'traj_vectors':
            trajectory_vectors()
        elif name == 'threads_blocks':
            run_threads_case()
        elif name == 'export_volumes':
            run_export_case()
        elif name == 'pickle_contract':
            run_contract_case()
        elif name in ROUGH:
            run_rough_case(name, **ROUGH[name])
        elif name.endswith(':default_vs_converged'):
            c = dict(CASES[name.split(':')[0]])
            run_default_vs_converged(name.split(':')[0], **{k: c[k] for k in ('shape', 'lattice', 'atoms', 'vacuum_tol') if k in c})
        else:
            run_case(name, **CASES[name])
            if ('changed', 2) in CASES[name].get('modes', ()) and ('changed', -1) in CASES[name].get('modes', ()) \
                    and not CASES[name].get('full_maps', True):
                run_default_vs_converged(name, **CASES[name])
