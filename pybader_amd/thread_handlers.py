"""Host-side mirror of pybader/thread_handlers.py for the hot path: same function names,
argument order, return values and silent-return cases, backed by libbader_hip.so.

`threads` is accepted for signature compatibility and ignored: the reference splits the grid into
thread blocks (thread_handlers.py:28-47); here the GPU library owns the decomposition."""
import numpy as np

from . import _lib, methods, refinement
from .utils import atom_assign, dtype_calc, ensure_density, ensure_labels, fetch_labels, track_labels

__all__ = ['bader_calc', 'refine', 'bader_calc_refine', 'assign_to_atoms', 'surface_distance', 'dtype_calc']

VERBOSE = True


def _say(*a):
    if VERBOSE:
        print(*a)


def bader_calc(method, density, volumes, dist_mat, T_grad, threads):
    """thread_handlers.bader_calc (thread_handlers.py:15-75).

    returns (bader_max int64[N,3] voxel indices, volumes narrowed to dtype_calc(-N))."""
    if method not in methods.__contains__:
        raise AttributeError(f"module 'pybader.methods' has no attribute '{method}'")   # getattr, line 26
    ctx = _lib.default_context()
    ctx.set_grid(density.shape, dist_mat, T_grad)
    ensure_density(ctx, density)
    ensure_labels(ctx, volumes)
    n = ctx.assign(method)
    bader_max = ctx.maxima()
    dtype = np.dtype(dtype_calc(-n))                                                   # lines 70-74
    if volumes.dtype == dtype and volumes.flags.c_contiguous:
        fetch_labels(ctx, volumes)
    else:
        volumes = fetch_labels(ctx, dtype=dtype)
    return bader_max, volumes


def refine(method, refine_mode, density, volumes, dist_mat, T_grad, threads):
    """thread_handlers.refine (thread_handlers.py:128-236): in place on `volumes`, returns None."""
    if method not in refinement.__contains__:      # getattr AttributeError -> silent return (140-143)
        return
    check_mode, iters = tuple(refine_mode)
    if iters == 0:                                 # lines 146-147
        return
    _say(f"\n  Refining {check_mode} edges:")
    ctx = _lib.default_context()
    ctx.set_grid(density.shape, dist_mat, T_grad)
    ensure_density(ctx, density)
    ensure_labels(ctx, volumes)
    log = ctx.refine(check_mode, iters)
    refine.last_log = log
    if not any(changed for _, changed in log):
        track_labels(ctx, volumes)                 # no voxel was relabelled: the host copy is still the device's
    elif volumes.flags.c_contiguous:
        fetch_labels(ctx, volumes)
    else:
        volumes.__setitem__(Ellipsis, ctx.download_labels(volumes.dtype))
    _say_log(log)


def _say_log(log):
    if not log:
        _say("  No edges found.")                  # thread_handlers.py:151-153
        return
    for k, (edges, changed) in enumerate(log):
        _say(f"  Iteration {k + 1}:\n  Refining {edges} edges: {changed} points changed.")


def bader_calc_refine(method, refine_method, refine_mode, density, volumes, dist_mat, T_grad, threads):
    """bader_calc followed by refine on its result -- what Bader.__call__ issues back to back (interface.py:406-409) -- as ONE
    library call (xb_assign_refine: the refinement's first iteration is queued behind the assignment on the device, one host wait
    for both, one label download instead of two).  Not a function of the reference: an entry point for callers that own both
    steps (pybader_amd.interface.Bader._run).  Same maxima, same final volumes and the same refinement log as the two calls
    (tests/test_gpu_parity.py::test_assign_refine_in_one_call_equals_the_two_calls); the pre-refinement map is never brought
    to the host.  Falls back to the two calls where the reference's refine returns silently.

    returns (bader_max int64[N,3], refined volumes narrowed to dtype_calc(-N))."""
    if method not in methods.__contains__:
        raise AttributeError(f"module 'pybader.methods' has no attribute '{method}'")
    check_mode, iters = tuple(refine_mode)
    if refine_method not in refinement.__contains__ or iters == 0:     # refine would return silently (140-147)
        return bader_calc(method, density, volumes, dist_mat, T_grad, threads)
    ctx = _lib.default_context()
    ctx.set_grid(density.shape, dist_mat, T_grad)
    ensure_density(ctx, density)
    ensure_labels(ctx, volumes)
    n, log = ctx.assign_refine(method, check_mode, iters)
    _say(f"\n  Refining {check_mode} edges:")
    refine.last_log = log
    bader_max = ctx.maxima()
    dtype = np.dtype(dtype_calc(-n))
    if volumes.dtype == dtype and volumes.flags.c_contiguous:
        fetch_labels(ctx, volumes)
    else:
        volumes = fetch_labels(ctx, dtype=dtype)
    _say_log(log)
    return bader_max, volumes


def assign_to_atoms(bader_max, atoms, lattice, volumes, threads):
    """thread_handlers.assign_to_atoms (thread_handlers.py:78-125)
    -> (bader_atoms int64[N], bader_distance f64[N], atoms_volumes narrowed to dtype_calc(-n_atoms))."""
    bader_atoms, bader_distance = atom_assign(bader_max, atoms, lattice)
    ctx = _lib.default_context()
    if ctx.shape != tuple(volumes.shape):
        ctx.set_grid(volumes.shape, np.zeros(27), np.zeros(9))
    ensure_labels(ctx, volumes)
    ctx.volume_assign(bader_atoms)
    atom_volumes = fetch_labels(ctx, dtype=np.dtype(dtype_calc(-atoms.shape[0])))
    return bader_atoms, bader_distance, atom_volumes


def surface_distance(density, volumes, lattice, atoms, threads):
    """thread_handlers.surface_distance (thread_handlers.py:239-297): minimum distance from every atom
    to the surface (edge voxels) of its volume; 0 for atoms whose volume has no edge voxel."""
    _say("\n  Calculating min. surface disance:")
    ctx = _lib.default_context()
    if ctx.shape != tuple(volumes.shape):
        ctx.set_grid(volumes.shape, np.zeros(27), np.zeros(9))
    ensure_density(ctx, density)
    ensure_labels(ctx, volumes)
    d2, edges = ctx.surface_distance(lattice, atoms)
    if edges == 0:
        _say("  No edges found.")          # thread_handlers.py:256-258 (returns None)
        return
    out = np.zeros(atoms.shape[0], dtype=np.float64)
    hit = np.isfinite(d2)
    out[hit] = d2[hit]**.5                   # utils.py:377
    return out
