"""VASP CHGCAR / CHG reader with the density block parsed on the GPU -- SURVEY.md 8(f) rank 4.

Same call signature and return value as the reference's `pybader.io.vasp.read` (io/vasp.py:15-164):
`(density, lattice, atoms, file_info)` with `density['charge']` / `density['spin']` float64 `[x][y][z]`
arrays already divided by the cell volume, the lattice scaled, the atoms wrapped into the cell and
Cartesian.  Only the 1.3e8-number text block (2.4 GB at 512^3) is handled differently: it is handed to
`xb_parse_density_text` as raw bytes (memory mapped, no Python token objects) and converted on the device,
bit-identical to numpy's string -> float64.  The charge density stays resident for the following
`bader_calc` (the upload a `Bader` run would start with is skipped).

The reference's writer (`pybader.io.vasp.write`) is untouched: `file_info['write_function']` is that very
function when pybader is importable (None otherwise), so `Bader.write_volume` keeps working.
"""
import mmap
import os

import numpy as np

from . import _lib, utils
from .interface import distance_matrix, gradient_transform

try:                                    # the export path (-e) keeps using the reference's writer
    from pybader.io.vasp import write as _reference_write
except Exception:                       # noqa: BLE001  (pybader absent, or its numba stack not importable)
    _reference_write = None

__extensions__ = ['chgcar', '.vasp']
__args__ = ['charge_flag', 'spin_flag', 'buffer_size']


def _line(mm, pos):
    """(text of the line starting at pos without the newline, position after it)"""
    end = mm.find(b'\n', pos)
    if end < 0:
        end = len(mm)
    return mm[pos:end].decode('ascii', 'replace'), min(end + 1, len(mm))


def _block_bytes(mm, start, n_values):
    """an upper bound of the bytes holding n_values numbers from `start` on: whole lines of the width of the
    first one plus slack (the device parser reports when it found fewer numbers than voxels)"""
    first, nxt = _line(mm, start)
    per_line = max(1, len(first.split()))
    width = max(1, nxt - start)
    end = min(len(mm), start + (n_values // per_line + 3) * width)
    # never cut a number in two: with lines of uneven width the estimate may end inside the last token it needs,
    # and the token count would still come out right -- end the slice on white space (or the end of the file)
    while end < len(mm) and mm[end] not in b' \t\r\n\x0b\x0c':
        end += 1
    return end - start


def read(fn, charge_flag=True, spin_flag=False, buffer_size=64, ctx=None):
    """Read the charge and/or spin density of a VASP CHGCAR (augmentation charges are ignored).
    `buffer_size` is accepted for signature compatibility and unused."""
    ctx = ctx or _lib.default_context()
    prefix, filename = os.path.split(fn)
    prefix = os.path.join(prefix, '')
    density = {}
    with open(fn, 'rb') as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        try:
            _, pos = _line(mm, 0)                                  # comment
            text, pos = _line(mm, pos)
            scale = np.array(text.split(), dtype=np.float64)       # one factor, or one per lattice vector
            lattice = np.zeros((3, 3), dtype=np.float64)
            for i in range(3):
                text, pos = _line(mm, pos)
                lattice[i] = text.split()
            text, pos = _line(mm, pos)
            atom_types = text.split()
            try:                                                   # VASP 4: no line of element symbols
                atom_nums = np.array(atom_types, dtype=np.int64)
                atom_types = None
            except ValueError:
                text, pos = _line(mm, pos)
                atom_nums = np.array(text.split(), dtype=np.int64)
            n_atoms = int(atom_nums.sum())
            text, pos = _line(mm, pos)
            direct = text.lstrip().lower().startswith('d')
            atoms = np.zeros((n_atoms, 3), dtype=np.float64)
            for i in range(n_atoms):
                text, pos = _line(mm, pos)
                atoms[i] = text.split()[:3]
            if not direct:
                atoms = np.dot(atoms, np.linalg.inv(lattice))
            atoms %= 1                                             # wrapped into the cell
            _, pos = _line(mm, pos)                                # blank line
            grid_line_pos = pos
            text, pos = _line(mm, pos)
            grid = np.array(text.split(), dtype=np.int64)
            grid_line = mm[grid_line_pos:pos]
            n_values = int(np.prod(grid))
            if scale.shape[0] == 1:
                lattice *= scale[0]
            else:
                lattice *= scale[:3, None]
            atoms = np.dot(atoms, lattice)
            lattice_vol = np.dot(lattice[0], np.cross(lattice[1], lattice[2]))

            shape = tuple(int(g) for g in grid)
            vl = lattice / np.array(shape, dtype=np.float64)[:, None]
            ctx.set_grid(shape, distance_matrix(vl), gradient_transform(vl))
            view = np.frombuffer(mm, dtype=np.uint8)
            blocks = {}
            charge_bytes = _block_bytes(mm, pos, n_values)
            blocks['charge'] = (pos, charge_bytes)
            if spin_flag:
                # the spin block follows a second copy of the grid line (after the augmentation data)
                at = mm.find(b'\n' + grid_line, pos + charge_bytes // 2)
                if at < 0:
                    print(f"  No spin density in {fn}")
                    spin_flag = False
                else:
                    spin_pos = at + 1 + len(grid_line)
                    blocks['spin'] = (spin_pos, _block_bytes(mm, spin_pos, n_values))
            # charge last: it is the array the partitioning runs on and stays resident
            for key in [k for k in ('spin', 'charge') if k in blocks and (k != 'charge' or charge_flag)]:
                start, nbytes = blocks[key]
                try:
                    ctx.parse_density_text(view[start:start + nbytes], lattice_vol)
                except _lib.BaderHipError as err:
                    if err.code != _lib.XB_E_SHORT:                         # OOM, a malformed token, ...: not a sizing problem
                        raise
                    ctx.parse_density_text(view[start:], lattice_vol)      # odd line widths: take the rest of the file
                density[key] = ctx.download_density()
            if 'charge' in density:
                utils.remember_density(ctx, density['charge'])
            del view
        finally:
            mm.close()
    file_info = {
        'filename': filename,
        'prefix': prefix,
        'file_type': 'VASP',
        'buffer_size': buffer_size,
        'write_function': _reference_write,
        'element_nums': atom_nums,
        'charge_flag': charge_flag,
        'spin_flag': spin_flag,
        'voxel_offset': np.zeros(3),
    }
    if atom_types is not None:
        file_info['elements'] = atom_types
    return density, lattice, atoms, file_info
