"""GPU tests of the CHGCAR text path: xb_parse_density_text through the C ABI against the fixtures captured
from the reference's reader (bit for bit) and against the oracle's strtod restatement on seeded text that
mixes every number shape (fast path, host fallback, Fortran order, ragged lines)."""
import os

import numpy as np
import pytest

import oracle
from pybader_amd import _lib, io_vasp

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
CASES = ['chgcar_py_12x11x14', 'chgcar_f90_16x16x16', 'chg_10col_9x11x13']


@pytest.fixture(scope='module')
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize('name', CASES)
def test_reader_equals_reference(ctx, name, tmp_path):
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    path = tmp_path / 'CHGCAR'
    path.write_bytes(g['file_bytes'].tobytes())
    spin = 'spin' in g.files
    density, lattice, atoms, info = io_vasp.read(str(path), spin_flag=spin, ctx=ctx)
    assert np.array_equal(density['charge'], g['charge'])          # bit for bit
    if spin:
        assert np.array_equal(density['spin'], g['spin'])
    assert np.array_equal(lattice, g['lattice'])
    assert np.allclose(atoms, g['atoms'], rtol=0, atol=1e-12)
    # the charge density is the resident one: the hot path starts without an upload
    assert np.array_equal(ctx.download_density(), g['charge'])


def seeded_text(shape, seed):
    rng = np.random.default_rng(seed)
    n = int(np.prod(shape))
    vals = rng.lognormal(0.0, 6.0, n) * rng.choice([-1.0, 1.0], n)
    toks = []
    for k, v in enumerate(vals):
        m = k % 9
        if m == 0:
            toks.append('%.11E' % v)
        elif m == 1:
            toks.append('%.5e' % v)
        elif m == 2:
            toks.append(repr(float('%.3f' % (v % 1000.0))))
        elif m == 3:
            toks.append('%d' % int(v % 100000))
        elif m == 4:
            toks.append('%.17E' % v)                     # 18 digits: host fallback
        elif m == 5:
            toks.append('%.11E' % (v * 1e-200))          # exponent outside the exact fast path: host
        elif m == 6:
            toks.append('0.0')
        elif m == 7:
            toks.append('+%.8E' % abs(v))
        else:
            toks.append('-.%011dE-%02d' % (int(abs(v) * 1e6) % 10**11, k % 20))
    lines, i = [], 0
    while i < n:                                         # ragged lines, tabs, blank lines
        w = 1 + int(rng.integers(0, 11))
        lines.append((' ' if w % 2 else '\t') + '  '.join(toks[i:i + w]))
        if w == 3:
            lines.append('')
        i += w
    return ('\n'.join(lines) + '\naugmentation occupancies 1 15\n 0.1 0.2\n').encode()


@pytest.mark.parametrize('shape,seed', [((7, 5, 3), 1), ((16, 8, 24), 2), ((40, 48, 56), 3)])
def test_parser_equals_oracle_on_seeded_text(ctx, shape, seed):
    text = seeded_text(shape, seed)
    ctx.set_grid(shape, np.zeros(27), np.zeros(9))
    n_tokens, n_host = ctx.parse_density_text(text, 3.25)
    want = oracle.parse_density_text(text, shape, 3.25)
    got = ctx.download_density()
    assert n_tokens >= want.size and n_host > 0           # both the device fast path and the host fallback ran
    assert np.array_equal(got.view(np.int64), want.view(np.int64))   # bit patterns (signed zeros included)


def test_parser_errors(ctx):
    ctx.set_grid((4, 4, 4), np.zeros(27), np.zeros(9))
    with pytest.raises(_lib.BaderHipError):
        ctx.parse_density_text(b' 1.0 2.0 3.0\n', 1.0)                       # fewer numbers than voxels
    with pytest.raises(_lib.BaderHipError):
        ctx.parse_density_text(b' '.join([b'1.0'] * 63 + [b'1.0x']), 1.0)    # a malformed number


def test_end_to_end_example(tmp_path, capsys):
    """examples/chgcar_charges.py on a fixture file: the flow CHGCAR -> resident density -> Bader steps runs
    through the Python mirror and conserves the charge."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('chgcar_charges', os.path.join(root, 'examples', 'chgcar_charges.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = np.load(os.path.join(GOLDEN, 'chgcar_f90_16x16x16.npz'))
    path = tmp_path / 'CHGCAR'
    path.write_bytes(g['file_bytes'].tobytes())
    import sys
    argv, sys.argv = sys.argv, ['chgcar_charges.py', str(path)]
    try:
        b = mod.main()
    finally:
        sys.argv = argv
    assert np.array_equal(b.charge, g['charge'])
    assert b.atoms_charge.shape[0] == g['atoms'].shape[0]
    total = b.atoms_charge.sum() + b.vacuum_charge
    assert abs(total - g['charge'].sum() * b.voxel_volume) < 1e-9 * abs(total)


def test_spin_sums_through_the_mirror(ctx, tmp_path):
    """with spin_flag the per-atom / per-volume sums run a second time over the spin
    density (interface.py:505-511, 519-525); checked against plain numpy sums over the final maps."""
    from pybader_amd.interface import Bader
    g = np.load(os.path.join(GOLDEN, 'chgcar_f90_16x16x16.npz'))
    path = tmp_path / 'CHGCAR'
    path.write_bytes(g['file_bytes'].tobytes())
    density, lattice, atoms, info = io_vasp.read(str(path), spin_flag=True, ctx=ctx)
    b = Bader(density, lattice, atoms, info, spin_flag=True)
    b()
    vv = b.voxel_volume
    for lab, q, s_, vol in ((b.bader_volumes, b.bader_charge, b.bader_spin, b.bader_volume),
                            (b.atoms_volumes, b.atoms_charge, b.atoms_spin, b.atoms_volume)):
        for k in range(q.shape[0]):
            m = lab == k
            assert abs(q[k] - density['charge'][m].sum() * vv) <= 1e-9 * max(1.0, abs(q[k]))
            assert abs(s_[k] - density['spin'][m].sum() * vv) <= 1e-9 * max(1.0, abs(s_[k]))
            assert abs(vol[k] - m.sum() * vv) <= 1e-9 * max(1.0, vol[k])
