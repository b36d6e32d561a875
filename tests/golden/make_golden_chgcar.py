#!/opt/conda/bin/python3.9
"""Golden fixtures for the CHGCAR text path, made by IMPORTING THE REFERENCE:

    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_chgcar.py

Small CHGCAR files are written with the reference's own writer (`pybader.io.vasp.write`, both number
formats) or by hand (CHG-style, 10 numbers per line), read back with the reference's reader
(`pybader.io.vasp.read`), and stored as data: the file's bytes plus the arrays the reference returned.
Nothing of the reference is copied.  Runs only in the build container (needs /root/reference)."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402,F401  (sets up the environment + numba shim the reference needs)
from pybader.io import vasp  # noqa: E402
from pybader.utils import nostdout  # noqa: E402

from pybader_amd import synth  # noqa: E402


def ref_read(path, spin):
    with nostdout():
        density, lattice, atoms, info = vasp.read(path, charge_flag=True, spin_flag=spin)
    return density, lattice, atoms, info


def case_written_by_reference(name, shape, lattice, fortran_format, spin):
    rho = synth.synth_density(shape, lattice, synth.ATOMS8, synth.BACKGROUND)
    density = {'charge': rho.copy()}
    if spin:
        rng = np.random.default_rng(7)
        # signed values only in the fixed-width Fortran format: with the E-format a minus sign widens the
        # line and the reference reader (which locates the spin block from equal line lengths) never finds it
        density['spin'] = (rng.random(shape) - (0.5 if fortran_format == 2 else 0.0)) * rho
    atoms = synth.atoms_cartesian(synth.ATOMS8, lattice)
    info = {'element_nums': np.array([len(atoms)]), 'elements': ['H'], 'charge_flag': True, 'spin_flag': spin,
            'fortran_format': fortran_format, 'buffer_size': 64, 'comment': 'golden\n'}
    d = tempfile.mkdtemp()
    with nostdout():
        vasp.write('t', atoms, lattice, {k: v.copy() for k, v in density.items()}, info, prefix=os.path.join(d, ''))
    path = os.path.join(d, 't-CHGCAR')
    return save(name, path, spin)


def case_chg_style(name, shape, lattice):
    """CHG-like: 10 numbers per line in %13.5E, lines of equal length as the reference reader requires."""
    rng = np.random.default_rng(11)
    vals = rng.lognormal(-3.0, 3.0, int(np.prod(shape)))
    vals[::17] = 0.0
    lines = ['golden chg', '   1.0']
    for row in lattice:
        lines.append(' %12.6f%12.6f%12.6f' % tuple(row))
    lines += ['   H', '     1', 'Direct', '  0.250000  0.250000  0.250000', '', ' %4d %4d %4d' % shape]
    body = []
    for i in range(0, vals.size, 10):
        body.append(''.join('%13.5E' % v for v in vals[i:i + 10]))
    d = tempfile.mkdtemp()
    path = os.path.join(d, 'CHG')
    with open(path, 'w') as f:
        f.write('\n'.join(lines + body) + '\n')
    return save(name, path, False)


def save(name, path, spin):
    density, lattice, atoms, info = ref_read(path, spin)
    raw = np.frombuffer(open(path, 'rb').read(), dtype=np.uint8)
    out = {'file_bytes': raw, 'lattice': lattice, 'atoms': atoms, 'charge': density['charge'],
           'element_nums': np.asarray(info['element_nums'])}
    if spin and 'spin' in density:
        out['spin'] = density['spin']
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, raw.size, 'bytes,', density['charge'].shape, 'spin' if 'spin' in out else '')


if __name__ == '__main__':
    case_written_by_reference('chgcar_py_12x11x14', (12, 11, 14), synth.TRICLINIC, 0, True)
    case_written_by_reference('chgcar_f90_16x16x16', (16, 16, 16), synth.CUBIC6, 2, True)
    case_chg_style('chg_10col_9x11x13', (9, 11, 13), synth.TRICLINIC)
