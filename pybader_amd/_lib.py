"""ctypes binding of libbader_hip.so (include/bader_hip.h).  No PyTorch, no CPU fallback: if the
library or a GPU is missing, every entry point raises."""
import ctypes as C
import threading
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('XB_LIBRARY') or os.path.join(HERE, 'libbader_hip.so')   # (XB_LIBRARY: A/B runs of two builds)

METHODS = {'ongrid': 0, 'neargrid': 1}          # methods.__contains__ (methods.py:12)
REFINE_MODES = {'all': 0, 'changed': 1}         # refine_mode[0] (thread_handlers.py:201-205)
DTYPE_CODE = {np.dtype(np.int8): 1, np.dtype(np.int16): 2, np.dtype(np.int32): 4, np.dtype(np.int64): 8}

# every symbol include/bader_hip.h declares: (restype, argtypes)
_vp, _i64, _dbl, _int = C.c_void_p, C.c_int64, C.c_double, C.c_int
_pi64, _pdbl = C.POINTER(C.c_int64), C.POINTER(C.c_double)
SYMBOLS = {
    'xb_last_error': (C.c_char_p, []),
    'xb_device_count': (_int, []),
    'xb_create': (_int, [_int, C.POINTER(_vp)]),
    'xb_destroy': (None, [_vp]),
    'xb_sync': (_int, [_vp]),
    'xb_stream': (_vp, [_vp]),
    'xb_set_grid': (_int, [_vp, _pi64, _pdbl, _pdbl, _i64, _i64]),
    'xb_upload_density': (_int, [_vp, _vp]),
    'xb_synth_density': (_int, [_vp, _pdbl, _pdbl, _i64, _dbl]),
    'xb_parse_density_text': (_int, [_vp, _vp, _i64, _dbl, _pi64, _pi64]),
    'xb_download_density': (_int, [_vp, _vp]),
    'xb_upload_labels': (_int, [_vp, _vp, _int]),
    'xb_download_labels': (_int, [_vp, _vp, _int]),
    'xb_upload_known': (_int, [_vp, _vp]),
    'xb_download_known': (_int, [_vp, _vp]),
    'xb_vacuum_assign': (_int, [_vp, _dbl, _dbl, _pdbl, _pdbl]),
    'xb_assign': (_int, [_vp, _int, _pi64]),
    'xb_get_maxima': (_int, [_vp, _vp, _i64]),
    'xb_assign_trace': (_int, [_vp, _int, _pi64]),
    'xb_assign_local_table': (_int, [_vp, _vp, _vp, _i64]),
    'xb_assign_finish': (_int, [_vp, _vp, _i64]),
    'xb_prepare_refine': (_int, [_vp]),
    'xb_edge_find': (_int, [_vp, _pi64]),
    'xb_refine_trace': (_int, [_vp, _pi64, _pi64]),
    'xb_refine_trace_escaped': (_int, [_vp, _pi64, _pi64]),
    'xb_escaped_paths': (_int, [_vp, _i64, _pi64, _pi64]),
    'xb_escaped_paths_fetch': (_int, [_vp, _vp, _vp, _vp, _vp]),
    'xb_gather_voxels': (_int, [_vp, _vp, _i64, _vp, _vp]),
    'xb_scatter_voxels': (_int, [_vp, _vp, _i64, _vp, _vp]),
    'xb_walkers_count': (_int, [_vp, _pi64, _pi64]),
    'xb_walkers_fetch': (_int, [_vp, _vp, _vp]),
    'xb_walkers_continue': (_int, [_vp, _vp, _i64]),
    'xb_walkers_apply': (_int, [_vp, _vp, _i64, _pi64, _pi64]),
    'xb_edge_check': (_int, [_vp, _pi64, _pi64]),
    'xb_edge_check_local': (_int, [_vp, _pi64]),
    'xb_edge_check_local_fetch': (_int, [_vp, _vp, _vp]),
    'xb_edge_check_global': (_int, [_vp, _vp, _vp, _i64, _pi64, _pi64]),
    'xb_refine': (_int, [_vp, _int, _i64, _vp, _i64, _pi64]),
    'xb_assign_refine': (_int, [_vp, _int, _int, _i64, _pi64, _vp, _i64, _pi64]),
    'xb_charge_sum': (_int, [_vp, _dbl, _i64, _vp, _vp]),
    'xb_volume_assign': (_int, [_vp, _vp, _i64]),
    'xb_atom_assign': (_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    'xb_surface_distance': (_int, [_vp, _vp, _vp, _i64, _vp, _pi64]),
    'xb_volume_mask': (_int, [_vp, _i64, _vp]),
    'xb_label_sum': (_int, [_vp, _i64, _pdbl, _pi64]),
    'xb_set_table_window': (_int, [_vp, _i64]),
    'xb_table_build': (_int, [_vp, _pi64]),
    'xb_table_local_seeds': (_int, [_vp, _vp, _i64]),
    'xb_brick_masks': (_int, [_vp, C.POINTER(_vp), _pi64, _pi64, _pi64]),
    'xb_table_ties': (_int, [_vp, _pi64]),
    'xb_table_finish': (_int, [_vp, _vp, _i64, _i64]),
    'xb_labels_ptr': (_vp, [_vp]),
    'xb_known_ptr': (_vp, [_vp]),
    'xb_density_ptr': (_vp, [_vp]),
    'xb_plane_elems': (_i64, [_vp]),
    'xb_copy_planes': (_int, [_vp, _int, _int, _vp, _i64, _i64]),
    'xb_label_wire': (_int, [_vp, _int, _vp]),
    'xb_host_alloc': (_int, [_i64, _vp]),
    'xb_host_free': (_int, [_vp]),
    'xb_brick_masks_copy': (_int, [_vp, _int, _vp, _i64, _i64]),
    'xb_set_halo': (_int, [_vp, _i64]),
    'xb_kernel_time': (_int, [_vp, _int, _pdbl, _pi64]),
    'xb_kernel_time_reset': (_int, [_vp]),
    'xb_enable_timing': (_int, [_vp, _int]),
    'xb_set_option': (_int, [_vp, _int, _int]),
    'xb_box_stats': (_int, [_vp, _pi64, _pi64]),
    'xb_brick_labels': (_int, [_vp, _vp, _i64, _pi64]),
    'xb_slow_path_stats': (_int, [_vp, _pi64, _pi64]),
    'xb_deferred_stats': (_int, [_vp, _pi64]),
    'xb_growth_stats': (_int, [_vp, _pi64, _pi64]),
    'xb_comm_unique_id': (_int, [_vp]),
    'xb_comm_init': (_int, [_vp, _int, _int, _vp]),
    'xb_comm_destroy': (_int, [_vp]),
    'xb_comm_exchange_planes': (_int, [_vp, _int, _int, _vp, _pi64, _pi64, _int, _vp, _pi64, _pi64]),
    'xb_comm_allreduce_i64': (_int, [_vp, _pi64, C.c_int64, _int]),
    'xb_comm_allgather_i64': (_int, [_vp, _pi64, C.c_int64, _pi64]),
    'xb_comm_share_brick_masks': (_int, [_vp, _pi64, _pi64]),
    'xb_comm_allgather_block': (_int, [_vp, _int, _pi64, _pi64]),
    'xb_comm_allreduce_block': (_int, [_vp]),
    'xb_slab_supported': (_int, [_vp, _int, _pi64]),
    'xb_slab_assign_masks': (_int, [_vp, _int, _int]),
    'xb_slab_assign_trace': (_int, [_vp]),
    'xb_slab_assign_finish': (_int, [_vp, _pi64, _pi64]),
    'xb_slab_refine_pass': (_int, [_vp]),
    'xb_slab_walkers_round': (_int, [_vp, _int, _int]),
    'xb_slab_walk_layout': (_int, [_vp, _pi64]),
    'xb_slab_walk_send': (_int, [_vp, _i64]),
    'xb_slab_refine_counts': (_int, [_vp, _pi64, _pi64]),
    'xb_slab_block': (_int, [_vp, _int, C.POINTER(_vp), _pi64, _pi64, _pi64]),
    'xb_slab_block_copy': (_int, [_vp, _int, _int, _vp, _i64, _i64]),
    'xb_host_waits': (_int, [_pi64]),
    'xb_comm_stats': (_int, [_vp, _pi64]),
    'xb_comm_info': (_int, [_vp, _pi64]),
    'xb_memory_stats': (_int, [_vp, _pi64, _pi64, _pi64]),
}

_lib = None


class BaderHipError(RuntimeError):
    """carries the library's XB_E_* code in `.code` (None for host-side failures)"""
    code = None


XB_E_SHORT = -6   # xb_parse_density_text: the text holds fewer numbers than the grid has voxels


def load():
    """dlopen libbader_hip.so and bind every declared symbol; raises if the library is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BaderHipError(f"{LIB_PATH} is missing: build it with `python -m pybader_amd.build` "
                                "(hipcc, gfx950); pybader_amd has no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)      # AttributeError here == ABI drift; fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


# ---- result arrays in page-locked memory ----------------------------------------------------------------------------------
# bader_calc returns a NEW narrowed label array per call (thread_handlers.py:70-74).  As a plain numpy allocation that is fresh
# pageable memory every time: the device-to-host copy is staged through a pinned chunk, copied again and takes a page fault per
# 4 KiB (2 ms per call for 16 MB at 256^3 against 0.35 ms on the bus).  pinned_empty() hands out arrays that live in page-locked
# buffers from a small pool; a buffer returns to the pool when the last reference to its array is gone (weakref.finalize), so a
# loop of calls stops allocating after its first pass.  The arrays are ordinary ndarrays (picklable, writeable).
_POOL_KEEP = 4            # free buffers kept per size; more are given back to the driver
_POOL_MAX_BYTES = 4 << 30  # free page-locked bytes the pool may hold in total (a batch of CHGCARs of many shapes must not pin the host)


class _PinnedBuf:
    __slots__ = ('ptr', 'nbytes', '__weakref__')

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes

    @property
    def __array_interface__(self):
        return {'shape': (self.nbytes,), 'typestr': '|u1', 'data': (self.ptr, False), 'version': 3}


_pool = {}                 # nbytes -> free pointers of that size (insertion order of the sizes = age)
_pool_bytes = 0            # bytes of all free buffers
_pool_lock = threading.Lock()


def _host_free(ptr):
    if _lib is not None:
        _lib.xb_host_free(C.c_void_p(ptr))


def _pool_evict(need):
    """(lock held) give free buffers back to the driver, oldest size first, until `need` more bytes fit under the cap"""
    global _pool_bytes
    for size in list(_pool):
        free = _pool[size]
        while free and _pool_bytes + need > _POOL_MAX_BYTES:
            _host_free(free.pop())
            _pool_bytes -= size
        if not free:
            del _pool[size]
        if _pool_bytes + need <= _POOL_MAX_BYTES:
            return


def _pool_release(ptr, nbytes):
    global _pool_bytes
    with _pool_lock:
        free = _pool.get(nbytes)
        if nbytes > _POOL_MAX_BYTES or (free is not None and len(free) >= _POOL_KEEP):
            _host_free(ptr)
            return
        _pool_evict(nbytes)
        _pool.setdefault(nbytes, []).append(ptr)
        _pool_bytes += nbytes


def pinned_empty(shape, dtype):
    """np.empty(shape, dtype) in page-locked memory from the pool (np.empty for small arrays, and when the driver has no more
    page-locked memory to give: a pageable result is slower, not wrong)"""
    import weakref
    global _pool_bytes
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dtype.itemsize
    if nbytes < (1 << 20):
        return np.empty(shape, dtype)
    ptr = None
    with _pool_lock:
        free = _pool.get(nbytes)
        if free:
            ptr = free.pop()
            _pool_bytes -= nbytes
            if not free:
                del _pool[nbytes]
    if ptr is None:
        p = C.c_void_p()
        rc = load().xb_host_alloc(nbytes, C.byref(p))
        if rc != 0 or not p.value:
            with _pool_lock:          # make room once (every free buffer of another size) and try again
                _pool_evict(_POOL_MAX_BYTES)
            rc = load().xb_host_alloc(nbytes, C.byref(p))
            if rc != 0 or not p.value:
                return np.empty(shape, dtype)
        ptr = p.value
    buf = _PinnedBuf(ptr, nbytes)
    weakref.finalize(buf, _pool_release, ptr, nbytes)
    return np.asarray(buf).view(dtype).reshape(shape)      # (a view of the buffer's uint8 array, which keeps `buf` alive through .base)


def pool_owned(a):
    """`a` came from pinned_empty: its base is the buffer's own uint8 array, which nobody else holds"""
    return isinstance(a.base, np.ndarray) and isinstance(a.base.base, _PinnedBuf)


def _numpy_owned(a):
    """the memory of `a` was allocated by numpy itself (malloc / calloc: private anonymous pages) -- not a memmap, a shared-memory
    segment or any other foreign buffer, whose untouched pages are NOT zeros"""
    owner = a
    while isinstance(owner, np.ndarray) and owner.base is not None:
        if isinstance(owner, np.memmap):
            return False
        owner = owner.base
    return isinstance(owner, np.ndarray) and not isinstance(owner, np.memmap) and owner.flags.owndata


def fast_any(a):
    """np.any(a) for a C-contiguous array WITHOUT touching pages nobody has touched: a fresh np.zeros() array is untouched
    anonymous memory -- the kernel's pagemap says so per page (neither present nor swapped: it reads as zeros) -- and scanning
    it would fault every page in (4.7 ms for the 64 MB label array of a 256^3 grid, of a 5 ms call pair).  Pages that are in
    use are scanned.  Only for memory numpy allocated itself: in a file-backed or shared mapping (np.memmap, shared_memory) a
    page this process has not touched holds data all the same.  Anything else, or anything unexpected, is np.any."""
    import mmap
    n = a.nbytes
    if n < (1 << 22) or not a.flags.c_contiguous or not _numpy_owned(a):
        return bool(np.any(a))
    try:
        addr = a.ctypes.data
        ps = mmap.PAGESIZE
        first = -(-addr // ps)            # first whole page
        last = (addr + n) // ps           # one past the last whole page
        if last <= first:
            return bool(np.any(a))
        with open('/proc/self/pagemap', 'rb', buffering=0) as f:
            f.seek(first * 8)
            flags = np.frombuffer(f.read((last - first) * 8), np.uint64)
        if flags.size != last - first:
            return bool(np.any(a))
        if np.any((flags >> np.uint64(61)) & np.uint64(1)):      # bit 61: file-mapped or shared-anonymous page -- not ours to guess about
            return bool(np.any(a))
        used = (flags >> np.uint64(62)) != 0          # bit 63 present, bit 62 swapped
        b = a.reshape(-1).view(np.uint8)
        head, tail = first * ps - addr, (addr + n) - last * ps
        if (head and b[:head].any()) or (tail and b[n - tail:].any()):
            return True
        idx = np.flatnonzero(used)
        if idx.size == 0:
            return False
        if idx.size * 4 > flags.size:
            return bool(np.any(a))
        brk = np.flatnonzero(np.diff(idx) != 1)
        starts = np.concatenate(([idx[0]], idx[brk + 1]))
        ends = np.concatenate((idx[brk], [idx[-1]])) + 1
        return any(b[head + s * ps: head + e * ps].any() for s, e in zip(starts, ends))
    except (OSError, ValueError):
        return bool(np.any(a))


def check(rc):
    if rc != 0:
        err = BaderHipError(f"libbader_hip error {rc}: {load().xb_last_error().decode()}")
        err.code = rc
        raise err


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Context:
    """One GPU context: device-resident density / labels / known + the hot-path calls."""

    def __init__(self, device=0):
        self.lib = load()
        n = self.lib.xb_device_count()
        if n <= 0:
            raise BaderHipError("no HIP device visible: the MI355X path cannot run (no CPU fallback)")
        h = C.c_void_p()
        check(self.lib.xb_create(int(device), C.byref(h)))
        self.h = h
        self.shape = None
        # utils.resident(): identity of the host array the caller pinned / of the one whose content is on the device
        self.pinned_density = None
        self.resident_density = None
        # utils.track_labels(): identity of the host array whose content equals the device labels (inside resident() only)
        self.resident_labels = None
        self._labels_host = None
        self.n_maxima = 0

    def drop_label_token(self):
        """the device labels are about to change (or be replaced): no host array equals them any more"""
        a = getattr(self, '_labels_host', None)
        if a is not None and getattr(self, '_labels_was_writeable', False):
            a.flags.writeable = True
        self._labels_host = None
        self.resident_labels = None

    def close(self):
        if getattr(self, 'h', None):
            self.lib.xb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- residency --------------------------------------------------------------------------
    def set_grid(self, shape, dist_mat, T_grad, x_range=None):
        shape = tuple(int(s) for s in shape)
        if shape != getattr(self, 'shape', None):
            self.drop_label_token()     # (another grid: no host array equals the device labels any more)
        x0, x1 = (0, shape[0]) if x_range is None else x_range
        sh = np.array(shape, dtype=np.int64)
        dm, tg = _f64(dist_mat).reshape(27), _f64(T_grad).reshape(9)
        check(self.lib.xb_set_grid(self.h, sh.ctypes.data_as(_pi64), dm.ctypes.data_as(_pdbl),
                                   tg.ctypes.data_as(_pdbl), int(x0), int(x1)))
        self.shape = shape
        self.x_range = (int(x0), int(x1))

    def set_halo(self, halo):
        check(self.lib.xb_set_halo(self.h, int(halo)))

    def upload_density(self, rho):
        rho = _f64(rho)
        assert rho.shape == self.shape, (rho.shape, self.shape)
        self.resident_density = None
        check(self.lib.xb_upload_density(self.h, _ptr(rho)))

    def synth_density(self, lattice, atoms, background):
        lat, at = _f64(lattice).reshape(9), _f64(atoms)
        self.resident_density = None
        check(self.lib.xb_synth_density(self.h, lat.ctypes.data_as(_pdbl), at.ctypes.data_as(_pdbl),
                                        at.shape[0], float(background)))

    def download_density(self):
        out = np.empty(self.shape, dtype=np.float64)
        check(self.lib.xb_download_density(self.h, _ptr(out)))
        return out

    def upload_labels(self, labels):
        self.drop_label_token()
        labels = np.ascontiguousarray(labels)
        assert labels.shape == self.shape
        check(self.lib.xb_upload_labels(self.h, _ptr(labels), DTYPE_CODE[labels.dtype]))

    def download_labels(self, dtype=np.int32, out=None, pooled=False):
        if out is None:
            out = pinned_empty(self.shape, dtype) if pooled else np.empty(self.shape, dtype=dtype)
        assert out.flags.c_contiguous and out.shape == self.shape
        check(self.lib.xb_download_labels(self.h, _ptr(out), DTYPE_CODE[out.dtype]))
        return out

    def upload_known(self, known):
        known = np.ascontiguousarray(known, dtype=np.int8)
        check(self.lib.xb_upload_known(self.h, _ptr(known)))

    def download_known(self):
        out = np.empty(self.shape, dtype=np.int8)
        check(self.lib.xb_download_known(self.h, _ptr(out)))
        return out

    # -- hot path ---------------------------------------------------------------------------
    def vacuum_assign(self, vac_tol, voxel_volume):
        self.drop_label_token()
        a, b = C.c_double(), C.c_double()
        tol = float('nan') if vac_tol is None else float(vac_tol)
        check(self.lib.xb_vacuum_assign(self.h, tol, float(voxel_volume), C.byref(a), C.byref(b)))
        return a.value, b.value

    def assign(self, method):
        self.drop_label_token()
        n = C.c_int64()
        check(self.lib.xb_assign(self.h, METHODS[method], C.byref(n)))
        self.n_maxima = n.value
        return n.value

    def maxima(self):
        out = np.zeros((self.n_maxima, 3), dtype=np.int64)
        check(self.lib.xb_get_maxima(self.h, _ptr(out), self.n_maxima))
        return out

    def assign_trace(self, method):
        self.drop_label_token()
        n = C.c_int64()
        check(self.lib.xb_assign_trace(self.h, METHODS[method], C.byref(n)))
        m, f = np.zeros(n.value, np.int64), np.zeros(n.value, np.int64)
        check(self.lib.xb_assign_local_table(self.h, _ptr(m), _ptr(f), n.value))
        return m, f

    def assign_finish(self, max_sorted):
        self.drop_label_token()
        ms = np.ascontiguousarray(max_sorted, dtype=np.int64)
        check(self.lib.xb_assign_finish(self.h, _ptr(ms), ms.shape[0]))
        self.n_maxima = int(ms.shape[0])

    def prepare_refine(self):
        check(self.lib.xb_prepare_refine(self.h))

    def edge_find(self):
        n = C.c_int64()
        check(self.lib.xb_edge_find(self.h, C.byref(n)))
        return n.value

    def refine_trace(self):
        self.drop_label_token()
        a, b = C.c_int64(), C.c_int64()
        check(self.lib.xb_refine_trace(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def refine_trace_escaped(self):
        self.drop_label_token()
        a, b = C.c_int64(), C.c_int64()
        check(self.lib.xb_refine_trace_escaped(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def parse_density_text(self, text, divisor):
        """the density block of a CHGCAR (bytes or a uint8 array, Fortran order) -> resident density / divisor;
        returns (numbers found, numbers converted by the host fallback)"""
        buf = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray, memoryview)) else text
        assert buf.dtype == np.uint8 and buf.flags.c_contiguous
        a, b = C.c_int64(), C.c_int64()
        self.resident_density = None            # also when the parse fails midway: rho is partly rewritten
        check(self.lib.xb_parse_density_text(self.h, _ptr(buf), buf.size, float(divisor), C.byref(a), C.byref(b)))
        return a.value, b.value

    def escaped_paths(self, max_len=1 << 15):
        """(starts, offsets, voxels, complete): for every parked (known == -6) voxel of the owned slab its start
        voxel followed by its trajectory from the first voxel outside the valid planes on; trajectories are
        followed for at most max_len voxels, complete[i] tells whether path i reached its maximum"""
        n, m = C.c_int64(), C.c_int64()
        check(self.lib.xb_escaped_paths(self.h, int(max_len), C.byref(n), C.byref(m)))
        starts = np.zeros(n.value, np.int64)
        offsets = np.zeros(n.value + 1, np.int64)
        vox = np.zeros(m.value, np.int64)
        complete = np.zeros(n.value, np.int8)
        check(self.lib.xb_escaped_paths_fetch(self.h, _ptr(starts), _ptr(offsets), _ptr(vox), _ptr(complete)))
        return starts, offsets, vox, complete.astype(bool)

    def gather_voxels(self, idx):
        idx = np.ascontiguousarray(idx, np.int64)
        lab = np.zeros(idx.size, np.int32)
        kn = np.zeros(idx.size, np.int8)
        check(self.lib.xb_gather_voxels(self.h, _ptr(idx), idx.size, _ptr(lab), _ptr(kn)))
        return lab, kn

    def scatter_voxels(self, idx, labels, known):
        self.drop_label_token()
        idx = np.ascontiguousarray(idx, np.int64)
        lab = np.ascontiguousarray(labels, np.int32)
        kn = np.ascontiguousarray(known, np.int8)
        check(self.lib.xb_scatter_voxels(self.h, _ptr(idx), idx.size, _ptr(lab), _ptr(kn)))

    def label_wire(self, widen_to=0):
        """bytes per label of a label halo on the wire; `widen_to` raises it (the ranks of a slab run must agree)"""
        w = C.c_int()
        check(self.lib.xb_label_wire(self.h, int(widen_to), C.byref(w)))
        return w.value

    WALKER_WORDS = 10

    def walkers(self):
        """(walkers, results) left by the last refine_trace / walkers_continue: (n, 10) int64 records of the retraces
        that left this rank's valid planes, and the (start voxel | label << 32) pairs of the ones it finished"""
        n, m = C.c_int64(), C.c_int64()
        check(self.lib.xb_walkers_count(self.h, C.byref(n), C.byref(m)))
        w = np.zeros((n.value, self.WALKER_WORDS), np.int64)
        r = np.zeros(m.value, np.int64)
        check(self.lib.xb_walkers_fetch(self.h, _ptr(w), _ptr(r)))
        return w, r

    def walkers_continue(self, walkers):
        w = np.ascontiguousarray(walkers, np.int64).reshape(-1, self.WALKER_WORDS)
        check(self.lib.xb_walkers_continue(self.h, _ptr(w), w.shape[0]))
        return self.walkers()

    def walkers_apply(self, results):
        """-> (changed, stuck)"""
        self.drop_label_token()
        r = np.ascontiguousarray(results, np.int64).reshape(-1)
        a, b = C.c_int64(), C.c_int64()
        check(self.lib.xb_walkers_apply(self.h, _ptr(r), r.size, C.byref(a), C.byref(b)))
        return a.value, b.value

    def edge_check(self):
        a, b = C.c_int64(), C.c_int64()
        check(self.lib.xb_edge_check(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def edge_check_local(self):
        """slabs: the owned changed voxels (int64 linear indices) and their edge&maximum class (int8)"""
        n = C.c_int64()
        check(self.lib.xb_edge_check_local(self.h, C.byref(n)))
        idx, cls = np.empty(n.value, np.int64), np.empty(n.value, np.int8)
        if n.value:
            check(self.lib.xb_edge_check_local_fetch(self.h, _ptr(idx), _ptr(cls)))
        return idx, cls

    def edge_check_global(self, idx, cls):
        """slabs: resolve the all-gathered list; returns (checked, new edges in the owned planes)"""
        idx, cls = np.ascontiguousarray(idx, np.int64), np.ascontiguousarray(cls, np.int8)
        a, b = C.c_int64(), C.c_int64()
        check(self.lib.xb_edge_check_global(self.h, _ptr(idx), _ptr(cls), idx.size, C.byref(a), C.byref(b)))
        return a.value, b.value

    def assign_refine(self, method, mode, iters):
        """bader_calc + refine in one library call (one host wait for both on the fused one-GPU neargrid path); -> (n_maxima, log)"""
        self.drop_label_token()
        cap = 4096
        log = np.zeros(cap, dtype=np.int64)
        n, k = C.c_int64(), C.c_int64()
        check(self.lib.xb_assign_refine(self.h, METHODS[method], REFINE_MODES[mode.lower()], int(iters), C.byref(n), _ptr(log), cap, C.byref(k)))
        m = min(k.value, cap // 2)
        self.n_maxima = n.value
        return n.value, [(int(log[2 * i]), int(log[2 * i + 1])) for i in range(m)]

    def refine(self, mode, iters):
        self.drop_label_token()
        cap = 4096
        log = np.zeros(cap, dtype=np.int64)
        n = C.c_int64()
        check(self.lib.xb_refine(self.h, REFINE_MODES[mode.lower()], int(iters), _ptr(log), cap, C.byref(n)))
        k = min(n.value, cap // 2)
        return [(int(log[2 * i]), int(log[2 * i + 1])) for i in range(k)]

    def charge_sum(self, voxel_volume, n_labels):
        ch, vo = np.zeros(n_labels, np.float64), np.zeros(n_labels, np.float64)
        check(self.lib.xb_charge_sum(self.h, float(voxel_volume), int(n_labels), _ptr(ch), _ptr(vo)))
        return ch, vo

    def volume_assign(self, swap):
        self.drop_label_token()
        sw = np.ascontiguousarray(swap, dtype=np.int64)
        check(self.lib.xb_volume_assign(self.h, _ptr(sw), sw.shape[0]))

    def surface_distance(self, lattice, atoms_cart):
        lat, at = _f64(lattice).reshape(9), _f64(atoms_cart).reshape(-1, 3)
        out = np.zeros(at.shape[0], np.float64)
        e = C.c_int64()
        check(self.lib.xb_surface_distance(self.h, _ptr(lat), _ptr(at), at.shape[0], _ptr(out), C.byref(e)))
        return out, e.value

    def volume_mask(self, vol_num):
        out = np.empty(self.shape, np.float64)
        check(self.lib.xb_volume_mask(self.h, int(vol_num), _ptr(out)))
        return out

    def label_sum(self, value):
        s, n = C.c_double(), C.c_int64()
        check(self.lib.xb_label_sum(self.h, int(value), C.byref(s), C.byref(n)))
        return s.value, n.value

    def set_table_window(self, margin):
        check(self.lib.xb_set_table_window(self.h, int(margin)))

    def table_build(self):
        n = C.c_int64()
        check(self.lib.xb_table_build(self.h, C.byref(n)))
        out = np.zeros(n.value, np.int64)
        check(self.lib.xb_table_local_seeds(self.h, _ptr(out), n.value))
        return out

    def brick_masks(self):
        p, n, f, k = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int64()
        check(self.lib.xb_brick_masks(self.h, C.byref(p), C.byref(n), C.byref(f), C.byref(k)))
        return p.value, n.value, f.value, k.value

    def table_ties(self):
        t = C.c_int64()
        check(self.lib.xb_table_ties(self.h, C.byref(t)))
        return bool(t.value)

    def table_finish(self, seeds, any_ties=True):
        sd = np.ascontiguousarray(seeds, dtype=np.int64)
        check(self.lib.xb_table_finish(self.h, _ptr(sd), sd.shape[0], int(bool(any_ties))))

    def copy_planes(self, which, to_device, host, xa, xb):
        self.drop_label_token()
        check(self.lib.xb_copy_planes(self.h, int(which), int(to_device), _ptr(host), int(xa), int(xb)))

    # -- measurement ------------------------------------------------------------------------
    def plane_elems(self):
        return int(self.lib.xb_plane_elems(self.h))

    def brick_masks_copy(self, host, first, count):
        """the chunk [first, first+count) of the per-brick move masks: host=None downloads it, else uploads `host`"""
        if host is None:
            out = np.empty(2 * int(count), np.int32)     # the masks, then the bricks' single-maximum voxels
            check(self.lib.xb_brick_masks_copy(self.h, 0, _ptr(out), int(first), int(count)))
            return out
        buf = np.ascontiguousarray(host, np.int32)
        assert buf.size == 2 * int(count)
        check(self.lib.xb_brick_masks_copy(self.h, 1, _ptr(buf), int(first), int(count)))

    # -- multi-GPU transport (csrc/comm.h: RCCL through the C ABI) ---------------------------------
    def comm_unique_id(self):
        buf = np.zeros(128, np.uint8)
        check(self.lib.xb_comm_unique_id(_ptr(buf)))
        return buf.tobytes()

    def comm_init(self, rank, size, unique_id):
        buf = np.frombuffer(unique_id, np.uint8).copy()
        assert buf.size == 128
        check(self.lib.xb_comm_init(self.h, int(rank), int(size), _ptr(buf)))
        self.comm_size = int(size)

    def comm_allreduce(self, vals, op='sum'):
        a = np.array([int(v) for v in vals], np.int64)
        check(self.lib.xb_comm_allreduce_i64(self.h, a.ctypes.data_as(_pi64), a.size, {'sum': 0, 'min': 1, 'max': 2}[op]))
        return a.tolist()

    def comm_allgather(self, vals, size=None):
        a = np.ascontiguousarray(vals, np.int64).reshape(-1)
        n = int(self.comm_size if size is None else size)
        out = np.empty(n * a.size, np.int64)
        check(self.lib.xb_comm_allgather_i64(self.h, a.ctypes.data_as(_pi64), a.size, out.ctypes.data_as(_pi64)))
        return out

    def comm_exchange_planes(self, which, sends, recvs):
        self.drop_label_token()
        def cols(ops):
            peer = np.array([o[0] for o in ops], np.int32)
            xa = np.array([o[1] for o in ops], np.int64)
            xb = np.array([o[2] for o in ops], np.int64)
            return peer, xa, xb
        sp, sa, sb = cols(sends)
        rp, ra, rb = cols(recvs)
        check(self.lib.xb_comm_exchange_planes(self.h, int(which), len(sends), _ptr(sp), sa.ctypes.data_as(_pi64),
                                               sb.ctypes.data_as(_pi64), len(recvs), _ptr(rp), ra.ctypes.data_as(_pi64),
                                               rb.ctypes.data_as(_pi64)))

    def comm_share_brick_masks(self, first, count):
        f, n = np.array(first, np.int64), np.array(count, np.int64)
        check(self.lib.xb_comm_share_brick_masks(self.h, f.ctypes.data_as(_pi64), n.ctypes.data_as(_pi64)))

    # ---- the slab step with its control flow on the device (csrc/slab_step.h) ----
    def slab_supported(self, nranks):
        a = C.c_int64(0)
        check(self.lib.xb_slab_supported(self.h, int(nranks), C.byref(a)))
        return bool(a.value)

    def slab_assign_masks(self, rank, nranks):
        check(self.lib.xb_slab_assign_masks(self.h, int(rank), int(nranks)))

    def slab_assign_trace(self):
        self.drop_label_token()
        check(self.lib.xb_slab_assign_trace(self.h))

    def slab_assign_finish(self):
        """-> (n_maxima, status): 0 done, 1 repeat the step, 2 use the host-driven calls"""
        self.drop_label_token()
        a, b = C.c_int64(0), C.c_int64(0)
        check(self.lib.xb_slab_assign_finish(self.h, C.byref(a), C.byref(b)))
        if b.value == 0:
            self.n_maxima = int(a.value)
        return int(a.value), int(b.value)

    def slab_refine_pass(self):
        self.drop_label_token()
        check(self.lib.xb_slab_refine_pass(self.h))

    def slab_walkers_round(self, src, last):
        self.drop_label_token()
        check(self.lib.xb_slab_walkers_round(self.h, int(src), 1 if last else 0))

    def slab_walk_send(self, walkers):
        """how many walkers of a rank's part travel in the gather of the next passes (0: the capacity); the same value on every rank"""
        check(self.lib.xb_slab_walk_send(self.h, int(walkers)))

    def slab_walk_layout(self):
        """(part bytes, header + walkers of round 0, header + walkers of later rounds, results offset, results bytes)"""
        out = np.zeros(5, np.int64)
        check(self.lib.xb_slab_walk_layout(self.h, out.ctypes.data_as(_pi64)))
        return [int(v) for v in out]

    def slab_refine_counts(self):
        """-> (local, summed): int64[8] each -- edges, changed, escaped, walkers still travelling (all ranks), slow-path
        retraces, walkers lost or stuck, this rank's travelling walkers, 0"""
        loc, glo = np.zeros(8, np.int64), np.zeros(8, np.int64)
        check(self.lib.xb_slab_refine_counts(self.h, loc.ctypes.data_as(_pi64), glo.ctypes.data_as(_pi64)))
        return loc, glo

    def slab_block(self, which):
        """(device pointer, bytes, offset of this rank's part, its bytes) of exchange block `which`"""
        p, a, b, d = _vp(), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(self.lib.xb_slab_block(self.h, int(which), C.byref(p), C.byref(a), C.byref(b), C.byref(d)))
        return int(p.value or 0), int(a.value), int(b.value), int(d.value)

    def slab_block_copy(self, which, host, off, to_device):
        """bytes [off, off + host.nbytes) of block `which` from (to_device) or into the uint8 array `host`"""
        assert host.dtype == np.uint8 and host.flags.c_contiguous
        check(self.lib.xb_slab_block_copy(self.h, int(which), 1 if to_device else 0, _ptr(host), int(off), host.nbytes))

    def comm_allgather_block(self, which, first, count):
        f, n = np.array(first, np.int64), np.array(count, np.int64)
        check(self.lib.xb_comm_allgather_block(self.h, int(which), f.ctypes.data_as(_pi64), n.ctypes.data_as(_pi64)))

    def comm_allreduce_block(self):
        check(self.lib.xb_comm_allreduce_block(self.h))

    def host_waits(self):
        """waits of the host for the card inside library calls made by this thread"""
        a = C.c_int64(0)
        check(self.lib.xb_host_waits(C.byref(a)))
        return int(a.value)

    def memory_stats(self):
        """(total, table, scratch) device bytes this context holds for the grid"""
        a, b, d = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(self.lib.xb_memory_stats(self.h, C.byref(a), C.byref(b), C.byref(d)))
        return int(a.value), int(b.value), int(d.value)

    def comm_info(self):
        """{ranks, rank, device, rccl_version} as the RCCL communicator itself reports them (-1: not available)"""
        out = (C.c_int64 * 4)()
        check(self.lib.xb_comm_info(self.h, out))
        return {'nccl_comm_count': int(out[0]), 'nccl_user_rank': int(out[1]), 'nccl_device': int(out[2]), 'rccl_version': int(out[3])}

    def comm_bytes_sent(self):
        n = C.c_int64(0)
        check(self.lib.xb_comm_stats(self.h, C.byref(n)))
        return int(n.value)

    def enable_timing(self, on=True, only=None):
        """on: all timers; only=[k, ...]: just these (xb_kernel_time's `which`)"""
        mask = int(bool(on)) if only is None else sum(2 << int(k) for k in only)
        check(self.lib.xb_enable_timing(self.h, mask))

    def kernel_time(self, which):
        ms, n = C.c_double(), C.c_int64()
        check(self.lib.xb_kernel_time(self.h, int(which), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_time_reset(self):
        check(self.lib.xb_kernel_time_reset(self.h))

    def set_option(self, key, value):
        check(self.lib.xb_set_option(self.h, int(key), int(value)))

    def box_stats(self):
        a, b = C.c_int64(), C.c_int64()
        check(self.lib.xb_box_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def brick_labels(self):
        """per 8^3 brick of the last neargrid assignment: the trapping region it was certified for (> 0) or 0 (walked)"""
        dims = (C.c_int64 * 3)()
        check(self.lib.xb_brick_labels(self.h, None, 0, dims))
        out = np.zeros(tuple(int(d) for d in dims), np.int32)
        check(self.lib.xb_brick_labels(self.h, _ptr(out), out.size, dims))
        return out

    def slow_path_stats(self):
        a, b = C.c_int64(), C.c_int64()
        check(self.lib.xb_slow_path_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def growth_stats(self):
        """(assignments repeated with the long kill schedule, kill launches scheduled now)"""
        a, b = C.c_int64(), C.c_int64()
        check(self.lib.xb_growth_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def deferred_stats(self):
        a = C.c_int64()
        check(self.lib.xb_deferred_stats(self.h, C.byref(a)))
        return a.value

    def sync(self):
        check(self.lib.xb_sync(self.h))


def atom_assign(bader_max_cart, atoms_cart, lattice):
    """utils.atom_assign (utils.py:185-232) through the C ABI (host-side, tiny)."""
    lib = load()
    bm, at, lat = _f64(bader_max_cart).reshape(-1, 3), _f64(atoms_cart).reshape(-1, 3), _f64(lattice).reshape(9)
    a, d = np.zeros(bm.shape[0], np.int64), np.zeros(bm.shape[0], np.float64)
    check(lib.xb_atom_assign(_ptr(bm), bm.shape[0], _ptr(at), at.shape[0], _ptr(lat), _ptr(a), _ptr(d)))
    return a, d


_default_ctx = {}


def default_context(device=None):
    """Process-wide context (one per device), created on first use."""
    if device is None:
        device = int(os.environ.get('PYBADER_AMD_DEVICE', os.environ.get('LOCAL_RANK', '0')))
        if load().xb_device_count() == 1:
            device = 0
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
