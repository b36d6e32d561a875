#!/opt/conda/bin/python3.9
"""The REAL pybader.interface.Bader driven through the INTEGRATION.md binding (missing #5).

Build container only (needs /root/reference and the conda python3.9 + numba 0.54.1):

    /opt/conda/bin/python3.9 -W ignore tests/golden/check_real_class_binding.py

The reference cannot travel to the GPU box and there is no GPU here, so the real class can never meet libbader_hip.so.  What
this script checks is everything in between: it (1) runs the unpatched reference class and pickles it, (2) executes the
monkey-patch code block of INTEGRATION.md section 1 VERBATIM, with the CPU-oracle context of tests/oracle_context.py standing
in for the GPU context, runs the real class again -- `Bader.__call__`, `to_file` and all -- and (3) holds the two pickles
against each other slot by slot: type, dtype, shape, integer arrays bit for bit, floats to 1e-9.  Prints one JSON line;
exit code 1 on any difference."""
import json
import os
import pickle
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as mg          # noqa: E402  (environment + numba shim + reference import, nothing is generated)
import numpy as np                # noqa: E402

from pybader.interface import Bader          # noqa: E402
from pybader.utils import nostdout           # noqa: E402
from pybader_amd import synth                # noqa: E402

PROFILES = {'default': dict(), 'default_vacuum_spin': dict(vacuum_tol=0.03, spin_flag=True),
            'speed': dict(method='ongrid', refine_method='neargrid', refine_mode=('changed', 3), speed_flag=True),
            'all_inf': dict(refine_mode=('all', -1))}


def run(conf, tag):
    kw = mg.CASES['c40x48x56_tric']
    lattice = np.asarray(kw['lattice'], np.float64)
    rho = synth.synth_density(kw['shape'], lattice, synth.ATOMS8)
    spin = np.ascontiguousarray(rho[::-1] * 0.25)
    dest = os.path.join(mg.SCRATCH, 'bader.p')              # (same name for both runs: it is stored in the pickle)
    info = {'filename': 'synth', 'prefix': '', 'file_type': 'synthetic', 'write_function': None,
            'voxel_offset': np.zeros(3), 'out_dest': dest}
    b = Bader({'charge': rho, 'spin': spin}, lattice, synth.atoms_cartesian(synth.ATOMS8, lattice), info, threads=1, **conf)
    with nostdout():
        b()
    with open(dest, 'rb') as f:
        return pickle.load(f)


def slots(obj):
    out = {}
    for s in Bader.__slots__:
        try:
            out[s] = object.__getattribute__(obj, s)
        except AttributeError:
            pass
    return out


def compare(a, b, where, problems):
    if set(a) != set(b):
        problems.append(f'{where}: slots differ: {sorted(set(a) ^ set(b))}')
    for k in sorted(set(a) & set(b)):
        x, y = a[k], b[k]
        if type(x) is not type(y):
            problems.append(f'{where}.{k}: type {type(x).__name__} != {type(y).__name__}')
        elif isinstance(x, np.ndarray):
            if x.dtype != y.dtype or x.shape != y.shape:
                problems.append(f'{where}.{k}: {x.dtype}{x.shape} != {y.dtype}{y.shape}')
            elif x.dtype.kind in 'iub':
                if not np.array_equal(x, y):
                    problems.append(f'{where}.{k}: {int((x != y).sum())} integer entries differ')
            elif not np.allclose(x, y, rtol=1e-9, atol=1e-12):
                problems.append(f'{where}.{k}: floats differ by up to {float(np.abs(x - y).max())}')
        elif isinstance(x, dict):
            compare({f'[{q}]': v for q, v in x.items() if not callable(v)}, {f'[{q}]': v for q, v in y.items() if not callable(v)},
                    f'{where}.{k}', problems)
        elif isinstance(x, float):
            if abs(x - y) > 1e-9 * max(1.0, abs(x)):
                problems.append(f'{where}.{k}: {x} != {y}')
        elif x != y and not (x is None and y is None):
            problems.append(f'{where}.{k}: {x!r} != {y!r}')


def main():
    reference = {p: slots(run(conf, 'ref_' + p)) for p, conf in PROFILES.items()}
    # ---- the binding, verbatim from INTEGRATION.md section 1 (the monkey-patch block) ----
    with open(os.path.join(os.path.dirname(os.path.dirname(HERE)), 'INTEGRATION.md')) as f:
        text = f.read()
    blocks = re.findall(r'```python\n(.*?)```', text, re.S)
    patch = [b for b in blocks if b.startswith('import pybader.interface as I, pybader_amd.thread_handlers as T')]
    assert len(patch) == 1, 'the monkey-patch block of INTEGRATION.md was not found'
    from pybader_amd import _lib, thread_handlers
    from oracle_context import OracleContext
    ctx = OracleContext()
    _lib.default_context = lambda device=None: ctx         # the CPU oracle in place of the GPU context
    import oracle                                          # (xb_atom_assign is host code inside libbader_hip.so, which this
    _lib.atom_assign = lambda bm, at, lat: oracle.atom_assign(np.ascontiguousarray(bm, np.float64), np.ascontiguousarray(at, np.float64),
                                                              np.ascontiguousarray(lat, np.float64))   # interpreter cannot load)
    thread_handlers.VERBOSE = False
    exec(patch[0], {})
    import pybader.interface as I
    assert I.bader_calc is thread_handlers.bader_calc and I.Bader.__call__.__name__ == '_call_resident'
    problems, transfers = [], {}
    for p, conf in PROFILES.items():
        ctx.calls.clear()
        patched = slots(run(conf, 'amd_' + p))
        compare(reference[p], patched, p, problems)
        transfers[p] = {k: ctx.calls.count(k) for k in ('upload_density', 'upload_labels', 'download_labels')}
        if ctx.pinned_density is not None or ctx.resident_labels is not None:
            problems.append(f'{p}: a residency token outlived Bader.__call__')
        # (with spin_flag the spin density takes the device buffer for its two sums and the charge density comes back twice)
        if transfers[p]['upload_density'] > (4 if conf.get('spin_flag') else 1):
            problems.append(f'{p}: the density was uploaded {transfers[p]["upload_density"]} times inside resident()')
    print(json.dumps({'profiles': list(PROFILES), 'slots_compared': {p: len(v) for p, v in reference.items()},
                      'transfers_per_call': transfers, 'problems': problems}))
    return 1 if problems else 0


if __name__ == '__main__':
    sys.exit(main())
