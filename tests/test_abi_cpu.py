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
