"""CPU tests of the CHGCAR text path: the oracle's strtod restatement of
io/vasp.py:90-104 against the fixtures captured from the reference's own reader, and the header logic of
pybader_amd/io_vasp.py with the oracle standing in for the device parser."""
import os

import numpy as np
import pytest

import oracle

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
CASES = ['chgcar_py_12x11x14', 'chgcar_f90_16x16x16', 'chg_10col_9x11x13']


class OracleParserContext:
    """stands in for _lib.Context in io_vasp.read: same three calls, oracle arithmetic"""
    shape = None

    def set_grid(self, shape, dist_mat, T_grad):
        self.shape = tuple(shape)

    def parse_density_text(self, text, divisor):
        self.rho = oracle.parse_density_text(bytes(text), self.shape, divisor)
        return self.rho.size, 0

    def download_density(self):
        return self.rho.copy()


def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


@pytest.mark.parametrize('name', CASES)
def test_reader_with_oracle_parser_equals_reference(name, tmp_path, monkeypatch):
    from pybader_amd import io_vasp, utils
    monkeypatch.setattr(utils, 'remember_density', lambda ctx, d: None)
    g = load(name)
    path = tmp_path / 'CHGCAR'
    path.write_bytes(g['file_bytes'].tobytes())
    spin = 'spin' in g.files
    density, lattice, atoms, info = io_vasp.read(str(path), spin_flag=spin, ctx=OracleParserContext())
    assert np.array_equal(density['charge'], g['charge'])          # bit for bit
    if spin:
        assert np.array_equal(density['spin'], g['spin'])
    assert np.array_equal(lattice, g['lattice'])
    # Cartesian atoms go through np.linalg.inv + np.dot: numpy 1.26 (fixture) and 2.2 (here) differ in the last ulp
    assert np.allclose(atoms, g['atoms'], rtol=0, atol=1e-12)
    assert np.array_equal(info['element_nums'], g['element_nums'])
    assert info['spin_flag'] == spin and info['file_type'] == 'VASP'


def test_oracle_parser_edge_cases():
    # Fortran order, tabs / blank lines, signs, exponents, trailing tokens ignored, malformed token refused
    t = b'\t1 -2.5E+00\n\n  .5e1 4 5 6 7 8 augmentation'
    a = oracle.parse_density_text(t, (2, 2, 2), 1.0)
    assert a[1, 0, 0] == -2.5 and a[0, 1, 0] == 5.0 and a[1, 1, 1] == 8.0
    with pytest.raises(ValueError):
        oracle.parse_density_text(b'1 2 3', (2, 2, 1), 1.0)
    with pytest.raises(ValueError):
        oracle.parse_density_text(b'1 2 x3 4', (2, 2, 1), 1.0)
