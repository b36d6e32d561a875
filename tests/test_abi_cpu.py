"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/bader_hip.h declares; without a GPU the product path fails loudly (no fallback)."""
import os
import re

import numpy as np
import pytest

from pybader_amd import _lib, build, interface, utils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    build.build_library()
    return _lib.load()


def test_header_and_binding_agree(lib):
    hdr = open(os.path.join(ROOT, 'include', 'bader_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(xb_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_no_gpu_fails_loudly(lib):
    if lib.xb_device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(_lib.BaderHipError):
        _lib.Context(0)
    with pytest.raises(_lib.BaderHipError):
        from pybader_amd import thread_handlers
        rho = np.ones((4, 4, 4))
        thread_handlers.bader_calc('neargrid', rho, np.zeros((4, 4, 4), np.int32), np.zeros((3, 3, 3)), np.eye(3), 1)


def test_atom_assign_needs_the_gpu_too(lib):
    """atom_assign runs on the device like the rest of the path: without a GPU it fails loudly"""
    if lib.xb_device_count() > 0:
        pytest.skip('a GPU is present')
    lat = np.eye(3) * 6.0
    with pytest.raises(_lib.BaderHipError):
        _lib.atom_assign(np.ones((2, 3)), np.ones((1, 3)), lat)


def test_dtype_calc_table(golden):
    g = golden('tables')
    for a, want in zip(g['dtype_calc_args'], g['dtype_calc_out']):
        assert utils.dtype_calc(int(a)) == str(want)


def test_host_matrices_match_reference(golden):
    """distance_matrix / T_grad are host-side numpy in the reference (interface.py:242-290); ours use the
    same operations.  numpy's pow/LAPACK differ in the last ulp between numpy versions (observed:
    1 ulp between numpy 1.26 and 2.2), so the kernels take these matrices as DATA and the bit-exact
    parity tests feed them the reference's captured matrices."""
    g = golden('tables')
    for k in range(3):
        vl = np.divide(g[f'lat{k}'], g[f'shape{k}'])
        np.testing.assert_allclose(interface.distance_matrix(vl), g[f'dist_mat{k}'], rtol=4e-16, atol=0)
        np.testing.assert_allclose(interface.gradient_transform(vl), g[f'T_grad{k}'], rtol=1e-13, atol=1e-14)


def test_synth_sha_is_stable(golden):
    from conftest import case_density
    case_density(golden('c12_cubic'))


def test_fast_any_equals_np_any_without_touching_fresh_pages():
    """_lib.fast_any (round 5): the zero test of utils.vacuum_assign reads the kernel's pagemap instead of faulting in every
    page of a fresh np.zeros array; every other case must give np.any's answer."""
    from pybader_amd import _lib
    shape = (160, 160, 160)                     # 16 MB of int32: above the size from which the pagemap is asked
    assert _lib.fast_any(np.zeros(shape, np.int32)) is False
    for where in (0, 12345678 % (160 ** 3), 160 ** 3 - 1):
        v = np.zeros(shape, np.int32)
        v.reshape(-1)[where] = -1
        assert _lib.fast_any(v) is True
    v = np.zeros(shape, np.int32)
    v[:] = 0                                    # touched, still zero
    assert _lib.fast_any(v) is False
    assert _lib.fast_any(np.ones(shape, np.int8)) is True
    assert _lib.fast_any(np.zeros((8, 8, 8), np.int32)) is False and _lib.fast_any(np.arange(8)) is True
    assert _lib.fast_any(np.zeros(shape, np.int32)[:, ::2]) is False     # not contiguous: np.any


def test_fast_any_does_not_guess_about_memory_numpy_did_not_allocate(tmp_path):
    """an untouched page of a file-backed or shared mapping is not present in this process and holds data all the
    same -- the pagemap shortcut is only for memory numpy allocated itself; everything else is np.any."""
    from multiprocessing import shared_memory
    from pybader_amd import _lib
    n = 2 << 20                                  # 8 MB of int32: above the size from which the pagemap is asked
    path = tmp_path / 'ones.bin'
    np.ones(n, np.int32).tofile(path)
    for mode in ('r', 'c', 'r+'):                # read-only, copy-on-write, shared
        m = np.memmap(path, dtype=np.int32, mode=mode)
        assert _lib.fast_any(m) is True
        assert _lib.fast_any(m.reshape(128, -1)) is True          # a view of a memmap
        assert _lib.fast_any(np.asarray(m)) is True               # a plain ndarray over the mapping
        del m
    shm = shared_memory.SharedMemory(create=True, size=4 * n)
    try:
        np.ndarray((n,), np.int32, buffer=shm.buf)[:] = 1
        other = shared_memory.SharedMemory(name=shm.name)        # a second mapping: none of its pages touched here
        try:
            assert _lib.fast_any(np.ndarray((n,), np.int32, buffer=other.buf)) is True
        finally:
            other.close()
    finally:
        shm.close()
        shm.unlink()
    assert _lib._numpy_owned(np.zeros(8)) and _lib._numpy_owned(np.zeros((4, 4))[1:]) and not _lib._numpy_owned(np.frombuffer(bytearray(64), np.uint8))


def test_pinned_pool_is_capped_and_falls_back(monkeypatch):
    """the pool of page-locked result buffers holds a bounded number of bytes, gives the oldest sizes back first and a
    failed allocation yields a pageable array instead of an error.  (A fake allocator: no GPU here.)"""
    import ctypes
    import gc
    from pybader_amd import _lib
    live, freed, fail = {}, [], {'on': False}

    class FakeLib:
        def xb_host_alloc(self, nbytes, pp):
            if fail['on']:
                return -3
            buf = ctypes.create_string_buffer(nbytes)
            addr = ctypes.addressof(buf)
            live[addr] = buf
            ctypes.cast(pp, ctypes.POINTER(ctypes.c_void_p))[0] = addr
            return 0

        def xb_host_free(self, p):
            freed.append(p.value)
            live.pop(p.value, None)
            return 0

    fake = FakeLib()
    monkeypatch.setattr(_lib, 'load', lambda: fake)
    monkeypatch.setattr(_lib, '_lib', fake)
    monkeypatch.setattr(_lib, '_pool', {})
    monkeypatch.setattr(_lib, '_pool_bytes', 0)
    monkeypatch.setattr(_lib, '_POOL_MAX_BYTES', 5 << 20)
    a = _lib.pinned_empty((1 << 20,), np.int16)          # 2 MB
    assert _lib.pool_owned(a)
    del a
    gc.collect()
    assert _lib._pool_bytes == 2 << 20 and not freed
    b = _lib.pinned_empty((1 << 20,), np.int16)          # reuses the free buffer
    assert _lib._pool_bytes == 0 and len(live) == 1
    c = _lib.pinned_empty((1 << 20,), np.int32)          # 4 MB
    del b, c
    gc.collect()
    assert _lib._pool_bytes <= 5 << 20 and len(freed) == 1      # 2 + 4 MB exceed the cap: the older size went back to the driver
    fail['on'] = True
    d = _lib.pinned_empty((3 << 20,), np.int8)           # no page-locked memory to be had: a pageable array, not an error
    assert isinstance(d, np.ndarray) and d.shape == (3 << 20,) and not _lib.pool_owned(d)


def test_xcd_range_hands_out_every_chunk_exactly_once():
    """The dealing of work lists to XCDs (csrc/bader_kernels.h xcd_range), restated: whatever the list length, the grid and the
    granularity, every chunk is taken by exactly one workgroup, and with a grid that is a multiple of 8 the chunks a workgroup
    takes lie in the contiguous part of its XCD (blockIdx % 8)."""
    def xcd_range(n_chunks, grid, block, whole=1):
        n_xcd = 8 if grid % 8 == 0 else 1
        k = block % n_xcd
        per = -(-n_chunks // n_xcd)
        per = -(-per // whole) * whole
        return k * per + block // n_xcd, min(n_chunks, (k + 1) * per), grid // n_xcd, k, per

    rng = np.random.default_rng(7)
    for _ in range(300):
        n_chunks = int(rng.integers(0, 5000))
        grid = int(rng.choice([1, 3, 8, 16, 40, 512, 4096, int(rng.integers(1, 700))]))
        whole = int(rng.choice([1, 8]))
        taken = np.zeros(n_chunks, np.int32)
        for b in range(grid):
            begin, end, step, k, per = xcd_range(n_chunks, grid, b, whole)
            for c in range(begin, end, step):
                assert k * per <= c < (k + 1) * per
                taken[c] += 1
        assert (taken == 1).all(), (n_chunks, grid, whole)
