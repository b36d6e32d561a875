"""ctypes front-end of the CPU oracle (oracle/bader_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by pybader_amd (the product path has no CPU fallback).

The functions mirror the reference's call signatures for the threads=1 path so the parity tests
read like calls into pybader itself; each cites the reference function it stands for.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liboracle.so')
_lib = None

_f64p = np.ctypeslib.ndpointer(np.float64, flags='C_CONTIGUOUS')
_i64p = np.ctypeslib.ndpointer(np.int64, flags='C_CONTIGUOUS')
_i32p = np.ctypeslib.ndpointer(np.int32, flags='C_CONTIGUOUS')
_i8p = np.ctypeslib.ndpointer(np.int8, flags='C_CONTIGUOUS')


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ('bader_oracle.c', 'bader_oracle_blocks.c', 'Makefile')]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(['make', '-s', '-C', _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        pp = C.POINTER(C.POINTER(C.c_int64))
        L.orc_ongrid.restype = C.c_int64
        L.orc_ongrid.argtypes = [_f64p, _i64p, _i32p, _f64p, pp]
        L.orc_neargrid.restype = C.c_int64
        L.orc_neargrid.argtypes = [_f64p, _i64p, _i32p, _f64p, _f64p, pp]
        L.orc_edge_find.restype = C.c_int64
        L.orc_edge_find.argtypes = [_i8p, _f64p, _i64p, _i32p]
        L.orc_edge_check.restype = None
        L.orc_edge_check.argtypes = [_i8p, _f64p, _i64p, _i32p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.orc_refine_neargrid.restype = C.c_int64
        L.orc_refine_neargrid.argtypes = [_i8p, _i8p, _f64p, _i64p, _i32p, _f64p, _f64p]
        L.orc_own_trajectory.restype = None
        L.orc_own_trajectory.argtypes = [_f64p, _i64p, _i32p, _f64p, _f64p, _i64p]
        L.orc_own_trajectory_main.restype = None
        L.orc_own_trajectory_main.argtypes = [_f64p, _i64p, _i32p, _f64p, _f64p, _i64p]
        L.orc_trajectory_path.restype = C.c_int64
        L.orc_trajectory_path.argtypes = [_f64p, _i64p, _f64p, _f64p, C.c_int64, _i64p, C.c_int64]
        L.orc_parse_density_text.restype = C.c_int64
        L.orc_parse_density_text.argtypes = [C.c_char_p, C.c_int64, _i64p, C.c_double, _f64p]
        L.orc_vacuum_assign.restype = None
        L.orc_vacuum_assign.argtypes = [_f64p, _i32p, C.c_double, _f64p, C.c_double, C.c_int64,
                                        C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_volume_offset.restype = None
        L.orc_volume_offset.argtypes = [_i32p, C.c_int64]
        L.orc_charge_sum.restype = None
        L.orc_charge_sum.argtypes = [_f64p, _f64p, C.c_int64, C.c_double, _f64p, _i32p, C.c_int64]
        L.orc_atom_assign.restype = None
        L.orc_atom_assign.argtypes = [_f64p, C.c_int64, _f64p, C.c_int64, _f64p, _i64p, _f64p]
        L.orc_volume_assign.restype = None
        L.orc_volume_assign.argtypes = [_i32p, _i64p, C.c_int64]
        L.orc_synth_density.restype = None
        L.orc_synth_density.argtypes = [_i64p, _f64p, _f64p, C.c_int64, C.c_double, _f64p]
        L.orc_free.restype = None
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_factor_3d.restype = None
        L.orc_factor_3d.argtypes = [C.c_int64, _i64p]
        L.orc_block_table.restype = C.c_int64
        L.orc_block_table.argtypes = [_i64p, C.c_int64, _i64p, _i64p, C.c_int64]
        L.orc_bader_calc_blocks.restype = C.c_int64
        L.orc_bader_calc_blocks.argtypes = [_f64p, _i64p, _i32p, _f64p, _f64p, C.c_int64, C.c_int64, pp]
        L.orc_refine_neargrid_blocks.restype = C.c_int64
        L.orc_refine_neargrid_blocks.argtypes = [_i8p, _i8p, _f64p, _i64p, _i32p, _f64p, _f64p, C.c_int64, C.c_int64]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _shape(a):
    return np.array(a.shape, dtype=np.int64)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def dtype_calc(max_val):
    """utils.dtype_calc (utils.py:15-37)."""
    names = (['int8', 'int16', 'int32', 'int64'], ['uint8', 'uint16', 'uint32', 'uint64'])
    if max_val < 0:
        max_val *= -2
        names = names[0]
    else:
        names = names[1]
    if max_val <= 255:
        return names[0]
    if max_val <= 65535:
        return names[1]
    if max_val <= 4294967295:
        return names[2]
    return names[3]


def factor_3d(x):
    """utils.factor_3d (utils.py:283-317)"""
    out = np.zeros(3, np.int64)
    lib().orc_factor_3d(int(x), out)
    return tuple(int(v) for v in out)


def block_table(shape, threads):
    """the thread blocks of thread_handlers.py:28-47: (origins int64[n,3], shapes int64[n,3]) in submission order"""
    sh = np.array(shape, np.int64)
    cap = max(1, int(threads)) * 8 + 8
    idx, ln = np.zeros((cap, 3), np.int64), np.zeros((cap, 3), np.int64)
    n = lib().orc_block_table(sh, int(threads), idx.reshape(-1), ln.reshape(-1), cap)
    return idx[:n].copy(), ln[:n].copy()


def bader_calc(method, density, volumes, dist_mat, T_grad, threads=1, workers=0):
    """thread_handlers.bader_calc (thread_handlers.py:15-75): kernel -> volume_offset -> merge -> dtype narrowing.
    threads in {0,1}: one block.  threads > 1 (neargrid): the factor_3d block split with the blocks merged in
    submission order (see bader_oracle_blocks.c), run by `workers` OpenMP threads (0: one per block).
    Returns (bader_max int64[N,3], volumes)."""
    L = lib()
    rho = _c(density, np.float64)
    vol = _c(volumes, np.int32).copy()
    out = C.POINTER(C.c_int64)()
    if threads > 1:
        if method != 'neargrid':
            raise NotImplementedError('the block path is restated for neargrid (the headline method) only')
        n = L.orc_bader_calc_blocks(rho, _shape(rho), vol.reshape(-1), _c(dist_mat, np.float64), _c(T_grad, np.float64),
                                    int(threads), int(workers), C.byref(out))
        bader_max = np.ctypeslib.as_array(out, shape=(n, 3)).copy() if n else np.zeros((0, 3), np.int64)
        L.orc_free(out)
        return bader_max, vol.astype(dtype_calc(-n))
    if method == 'ongrid':
        n = L.orc_ongrid(rho, _shape(rho), vol, _c(dist_mat, np.float64), C.byref(out))
    elif method == 'neargrid':
        n = L.orc_neargrid(rho, _shape(rho), vol, _c(dist_mat, np.float64), _c(T_grad, np.float64), C.byref(out))
    else:
        raise AttributeError(method)
    bader_max = np.ctypeslib.as_array(out, shape=(n, 3)).copy() if n else np.zeros((0, 3), np.int64)
    L.orc_free(out)
    L.orc_volume_offset(vol.reshape(-1), vol.size)
    return bader_max, vol.astype(dtype_calc(-n))


def edge_find(known, density, volumes):
    """refinement.edge_find (refinement.py:326-405); `known` int8 is updated in place."""
    assert known.dtype == np.int8 and known.flags.c_contiguous
    return int(lib().orc_edge_find(known, _c(density, np.float64), _shape(density), _c(volumes, np.int32)))


def edge_check(known, density, volumes):
    """refinement.edge_check (refinement.py:409-508) -> (checked, edges)."""
    assert known.dtype == np.int8 and known.flags.c_contiguous
    a, b = C.c_int64(), C.c_int64()
    lib().orc_edge_check(known, _c(density, np.float64), _shape(density), _c(volumes, np.int32), C.byref(a), C.byref(b))
    return a.value, b.value


def refine_neargrid(known, rknown, density, volumes, dist_mat, T_grad, threads=1, workers=0):
    """refinement.neargrid (refinement.py:17-322); `known` and `volumes` (any int dtype) in place.  threads > 1:
    over the blocks of thread_handlers.refine (thread_handlers.py:154-205), `workers` OpenMP threads."""
    assert known.dtype == np.int8 and known.flags.c_contiguous
    vol = _c(volumes, np.int32)
    copy = vol is not volumes
    if copy:
        vol = vol.copy()
    args = (known, _c(rknown, np.int8), _c(density, np.float64), _shape(density), vol, _c(dist_mat, np.float64),
            _c(T_grad, np.float64))
    ch = lib().orc_refine_neargrid_blocks(*args, int(threads), int(workers)) if threads > 1 else lib().orc_refine_neargrid(*args)
    if copy:
        volumes[...] = vol
    return int(ch)


def refine(method, refine_mode, density, volumes, dist_mat, T_grad, threads=1, log=None, workers=0):
    """thread_handlers.refine (thread_handlers.py:128-236).  In place; returns None like the reference.  `log` (a
    list) receives (edges, changed) per iteration.  threads > 1: the retraces run over the reference's blocks
    (edge_find / edge_check stay sequential, as in the reference)."""
    if method != 'neargrid':      # getattr(refinement, method) AttributeError -> silent return
        return
    check_mode, iters = tuple(refine_mode)
    if iters == 0:
        return
    known = np.zeros(density.shape, dtype=np.int8)
    edges = edge_find(known, density, volumes)
    if edges == 0:
        return
    changed = refine_neargrid(known, known.copy(), density, volumes, dist_mat, T_grad, threads, workers)
    if log is not None:
        log.append((edges, changed))
    if iters < 0:
        iters = float('inf')
    iter_num = 2
    while iter_num <= iters:
        if check_mode.lower() == 'all':
            known = np.zeros(density.shape, dtype=np.int8)
            edges = edge_find(known, density, volumes)
        else:
            _, edges = edge_check(known, density, volumes)
        changed = refine_neargrid(known, known.copy(), density, volumes, dist_mat, T_grad, threads, workers)
        if log is not None:
            log.append((edges, changed))
        if changed == 0:
            break
        iter_num += 1


def own_trajectory_map(density, volumes, dist_mat, T_grad, main_ties=False):
    """Map F of SURVEY.md 7.3: linear index of the maximum each voxel's own trajectory reaches (refinement.py's
    strict tie test; main_ties=True: the tie test of methods.py:324)."""
    out = np.empty(density.shape, dtype=np.int64)
    (lib().orc_own_trajectory_main if main_ties else lib().orc_own_trajectory)(_c(density, np.float64), _shape(density), _c(volumes, np.int32),
                             _c(dist_mat, np.float64), _c(T_grad, np.float64), out)
    return out


def trajectory_path(density, dist_mat, T_grad, start, cap=1 << 16):
    """Linear voxel indices of the own (dr=0) trajectory of `start`, maximum last."""
    out = np.empty(cap, dtype=np.int64)
    n = lib().orc_trajectory_path(_c(density, np.float64), _shape(density), _c(dist_mat, np.float64),
                                  _c(T_grad, np.float64), int(start), out, cap)
    if n < 0:
        raise RuntimeError('trajectory longer than the buffer')
    return out[:n].copy()


def parse_density_text(text, shape, divisor):
    """The density block of a CHGCAR (bytes, Fortran order) -> float64 [x][y][z] / divisor (io/vasp.py:90-104)."""
    out = np.zeros(tuple(int(s) for s in shape), dtype=np.float64)
    n = lib().orc_parse_density_text(text, len(text), np.array(shape, dtype=np.int64), float(divisor), out)
    if n != out.size:
        raise ValueError(f'density block: {n} numbers converted, {out.size} expected')
    return out


def vacuum_assign(reference, volumes, vac_tol, density, voxel_volume):
    """utils.vacuum_assign (utils.py:382-401); volumes must be int32 here."""
    assert volumes.dtype == np.int32
    a, b = C.c_double(), C.c_double()
    lib().orc_vacuum_assign(_c(reference, np.float64), volumes, float(vac_tol), _c(density, np.float64),
                            float(voxel_volume), volumes.size, C.byref(a), C.byref(b))
    return volumes, a.value, b.value


def charge_sum(charge, volume, voxel_volume, density, volumes):
    """utils.charge_sum (utils.py:235-252), in place on charge/volume."""
    lib().orc_charge_sum(charge, volume, charge.shape[0], float(voxel_volume), _c(density, np.float64),
                         _c(volumes, np.int32), volumes.size)


def atom_assign(bader_max, atoms, lattice):
    """utils.atom_assign (utils.py:185-232)."""
    n = bader_max.shape[0]
    a = np.zeros(n, np.int64)
    d = np.zeros(n, np.float64)
    lib().orc_atom_assign(_c(bader_max, np.float64), n, _c(atoms, np.float64), atoms.shape[0],
                          _c(lattice, np.float64), a, d)
    return a, d


def assign_to_atoms(bader_max, atoms, lattice, volumes, threads=1):
    """thread_handlers.assign_to_atoms (thread_handlers.py:78-125)."""
    bader_atoms, bader_distance = atom_assign(bader_max, atoms, lattice)
    av = _c(volumes, np.int32).copy()
    lib().orc_volume_assign(av.reshape(-1), bader_atoms, av.size)
    return bader_atoms, bader_distance, av.astype(dtype_calc(-atoms.shape[0]))


def synth_density(shape, lattice, atoms, background):
    rho = np.empty(tuple(shape), dtype=np.float64)
    lib().orc_synth_density(np.array(shape, np.int64), _c(lattice, np.float64), _c(atoms, np.float64),
                            atoms.shape[0], float(background), rho)
    return rho
