"""Host-side mirror of the njit half of pybader/utils.py that sits on the hot path.

Same names, argument order and in-place behaviour as the reference; the sweeps run in
libbader_hip.so on the GPU (no CPU fallback)."""
import numpy as np

from . import _lib

def dtype_calc(max_val):
    """utils.dtype_calc (utils.py:15-37): smallest dtype holding max_val; a negative argument asks
    for a signed type able to hold 2*|max_val|."""
    signed = max_val < 0
    if signed:
        max_val *= -2
    width = 0 if max_val <= 255 else 1 if max_val <= 65535 else 2 if max_val <= 4294967295 else 3
    return ('int8', 'int16', 'int32', 'int64')[width] if signed else ('uint8', 'uint16', 'uint32', 'uint64')[width]


# ---- which host array is resident on the device -------------------------------------------------------------
# The reference passes the same `density` ndarray to bader_calc, refine and charge_sum.  By default every call
# uploads it again (about 0.1 s per GiB): nothing is assumed about the content of a host array.  A caller who
# holds the array still for a while says so with `resident(density)`: inside that block the array is uploaded
# once, is read-only (an in-place edit raises instead of being missed) and is recognised by identity (address,
# shape, strides).  The token lives in the Context, and every route that rewrites the device density (upload,
# CHGCAR text parse -- also a failed one --, synthetic generator) clears it.
def _identity(a):
    return (a.ctypes.data, a.shape, a.strides)


class resident:
    """with resident(density): ... -- a promise not to modify `density` inside the block; uploaded at most once."""

    def __init__(self, density, ctx=None):
        self.array = density
        self.ctx = ctx

    def __enter__(self):
        a = self.array
        if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous):
            raise ValueError('resident(): a C-contiguous float64 ndarray is required')
        self.ctx = self.ctx or _lib.default_context()
        self._was_writeable = bool(a.flags.writeable)
        a.flags.writeable = False
        self.ctx.pinned_density = _identity(a)
        return a

    def __exit__(self, *exc):
        release_labels(self.ctx)
        self.ctx.pinned_density = None
        self.ctx.resident_density = None
        if self._was_writeable:
            self.array.flags.writeable = True
        return False


# ---- the label array on the device -------------------------------------------------------------------------------
# bader_calc -> refine -> charge_sum -> assign_to_atoms hand the SAME host label array from call to call
# (interface.py:406-416); uploading and downloading it every time made the drop-in PCIe-bound (22.6 ms per
# bader_calc + refine pair at 256^3 against 1 ms resident).  Inside a `resident(density)` block the library keeps track of
# the one host array whose content equals the device labels: it is recognised by identity (address, shape, strides,
# dtype), it is READ-ONLY while the token lives (numpy raises on a write instead of the copy going stale; the library's own
# in-place updates lift the flag for the duration of the download), the context holds a reference to it (its address
# cannot be reused), and every device call that rewrites the labels drops the token.  Outside such a block nothing is
# assumed: every call uploads.
def _lab_identity(a):
    return (a.ctypes.data, a.shape, a.strides, a.dtype.str)


def release_labels(ctx):
    ctx.drop_label_token()


def track_labels(ctx, volumes):
    """the device labels equal `volumes` (just uploaded from it / downloaded into it)"""
    release_labels(ctx)
    # (a view of a writeable array is not tracked: a write through its base would leave the device copy stale unnoticed)
    if (ctx.pinned_density is not None and isinstance(volumes, np.ndarray) and volumes.flags.c_contiguous
            and (_lib.pool_owned(volumes) or not (isinstance(volumes.base, np.ndarray) and volumes.base.flags.writeable))):
        ctx._labels_host, ctx._labels_was_writeable = volumes, bool(volumes.flags.writeable)
        volumes.flags.writeable = False
        ctx.resident_labels = _lab_identity(volumes)


def labels_resident(ctx, volumes):
    return (ctx.pinned_density is not None and getattr(ctx, '_labels_host', None) is volumes and not volumes.flags.writeable
            and ctx.resident_labels == _lab_identity(volumes))


def ensure_labels(ctx, volumes):
    """make `volumes` the device labels: uploads unless it is the tracked array"""
    if labels_resident(ctx, volumes):
        return
    ctx.upload_labels(volumes)
    track_labels(ctx, volumes)


def fetch_labels(ctx, out=None, dtype=None):
    """device labels -> host (in place into `out`, else a new array of `dtype`), and track the result"""
    release_labels(ctx)
    out = ctx.download_labels(out=out) if out is not None else ctx.download_labels(dtype, pooled=True)
    track_labels(ctx, out)
    return out


def ensure_density(ctx, density):
    """Make `density` the device density: uploads, unless the caller pinned this very array with `resident()` and
    it is the one on the device."""
    density = np.ascontiguousarray(density, dtype=np.float64)
    ident = _identity(density)
    if ctx.pinned_density is not None and ctx.pinned_density == ident and ctx.resident_density == ident:
        return density
    ctx.upload_density(density)                 # clears ctx.resident_density
    if ctx.pinned_density == ident:
        ctx.resident_density = ident
    return density


def remember_density(ctx, density):
    """`density` (a pinned host array) was just downloaded from the context: it is the resident one."""
    if ctx.pinned_density is not None and ctx.pinned_density == _identity(density):
        ctx.resident_density = _identity(density)


def forget_density(ctx):
    ctx.resident_density = None


def vacuum_assign(reference, volumes, vac_tol, density, voxel_volume):
    """utils.vacuum_assign (utils.py:382-401): volumes[reference <= vac_tol] = -1; returns
    (volumes, vacuum charge, vacuum volume).  `volumes` is updated in place."""
    ctx = _lib.default_context()
    if ctx.shape != tuple(volumes.shape):
        ctx.set_grid(volumes.shape, np.zeros(27), np.zeros(9))
    same = reference is density or (reference.shape == density.shape and np.shares_memory(reference, density))
    ensure_density(ctx, reference)
    release_labels(ctx)
    charge, volume = ctx.vacuum_assign(vac_tol, voxel_volume)
    if not same and volume:
        # vacuum is decided on `reference`, its charge is summed over `density` (utils.py:396-400)
        ensure_density(ctx, density)
        s, n = ctx.label_sum(-1)
        charge, volume = s * voxel_volume, n * voxel_volume
    # the device sets non-vacuum voxels to 0; the reference leaves them untouched
    if _lib.fast_any(volumes):
        keep = volumes.copy()
        ctx.download_labels(out=volumes)
        np.copyto(volumes, keep, where=volumes != -1)
    else:
        if volume:                       # some voxel became vacuum: fetch the -1 marks
            ctx.download_labels(out=volumes)
        track_labels(ctx, volumes)       # all zeros before, so host == device now (also without a download)
    return volumes, charge, volume


def charge_sum(charge, volume, voxel_volume, density, volumes):
    """utils.charge_sum (utils.py:235-252): in place on `charge` / `volume` like the reference."""
    ctx = _lib.default_context()
    if ctx.shape != tuple(volumes.shape):
        ctx.set_grid(volumes.shape, np.zeros(27), np.zeros(9))
    ensure_density(ctx, density)
    ensure_labels(ctx, volumes)
    ch, vo = ctx.charge_sum(voxel_volume, charge.shape[0])
    # utils.py:251-252 scales the accumulated charge (whatever it held on entry) by voxel_volume
    charge *= voxel_volume
    charge += ch
    volume += vo


def atom_assign(bader_max, atoms, lattice, i_c=None):
    """utils.atom_assign (utils.py:185-232) -> (assigned_atom int64[N], assigned_distance f64[N])."""
    return _lib.atom_assign(bader_max, atoms, lattice)


def volume_assign(volumes, swap, i_c=None):
    """utils.volume_assign (utils.py:404-421): volumes[v] = swap[volumes[v]] for labels >= 0, in place."""
    ctx = _lib.default_context()
    if ctx.shape != tuple(volumes.shape):
        ctx.set_grid(volumes.shape, np.zeros(27), np.zeros(9))
    ensure_labels(ctx, volumes)
    ctx.volume_assign(swap)
    fetch_labels(ctx, volumes)


def volume_mask(volumes, density, vol_num):
    """utils.volume_mask (utils.py:461-476): `density` where volumes == vol_num, zero elsewhere."""
    ctx = _lib.default_context()
    if ctx.shape != tuple(volumes.shape):
        ctx.set_grid(volumes.shape, np.zeros(27), np.zeros(9))
    ensure_density(ctx, density)
    ensure_labels(ctx, volumes)
    return ctx.volume_mask(vol_num)
