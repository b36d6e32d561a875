#!/usr/bin/env python3
"""End to end on one MI355X: CHGCAR file -> per-atom Bader charges, with the time of every stage.

    python examples/chgcar_charges.py path/to/CHGCAR [--method neargrid] [--refine changed:2]

The file is read by pybader_amd.io_vasp.read (density block parsed on the GPU, charge density left
resident), the partitioning runs through pybader_amd.interface.Bader -- the same step methods, in the same
order, as pybader's `Bader.__call__` (interface.py:398-416)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('chgcar')
    ap.add_argument('--method', default='neargrid', choices=['neargrid', 'ongrid'])
    ap.add_argument('--refine', default='changed:2')
    args = ap.parse_args()
    from pybader_amd import io_vasp
    from pybader_amd.interface import Bader
    mode, iters = args.refine.split(':')
    t = [time.perf_counter()]
    density, lattice, atoms, info = io_vasp.read(args.chgcar)
    t.append(time.perf_counter())
    b = Bader(density, lattice, atoms, info, method=args.method, refine_mode=(mode, int(iters)))
    b.volumes_init()
    t.append(time.perf_counter())
    b.bader_calc()
    t.append(time.perf_counter())
    b.refine_volumes(b.bader_volumes)
    t.append(time.perf_counter())
    b.bader_to_atom_distance()
    b.sum_volumes()
    t.append(time.perf_counter())
    names = ['read (GPU text parse + download)', 'volumes_init', 'bader_calc', 'refine_volumes', 'atoms + sums']
    print(f"{info['filename']}: grid {density['charge'].shape}, {len(atoms)} atoms, {b.bader_maxima.shape[0]} Bader maxima")
    for n, a, c in zip(names, t[:-1], t[1:]):
        print(f'  {n:<34s} {1e3 * (c - a):9.2f} ms')
    print('  atom        charge        volume')
    for k, (q, v) in enumerate(zip(b.atoms_charge, b.atoms_volume)):
        print(f'  {k:4d}  {q:12.6f}  {v:12.6f}')
    total = float(np.sum(b.atoms_charge)) + float(getattr(b, 'vacuum_charge', 0.0))
    print(f'  sum of atomic charges {total:.6f} (integral of the density {float(density["charge"].sum() * b.voxel_volume):.6f})')
    return b


if __name__ == '__main__':
    main()
