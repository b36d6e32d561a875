"""Random soak of the one-GPU pipeline against the CPU oracle (test infrastructure; run by hand on a GPU box, and in a short
seeded form by tests/test_gpu_soak.py):

    python tests/soak_vs_oracle.py [cases] [seed] [--method neargrid|ongrid] [--odd | --oddbig | --big]

Each case: a random grid shape (whole 8^3 bricks, or anything from 10 to 60 with --odd: the routes for grids that are not
made of bricks), one of four lattices, optional noise / plateaus / vacuum tolerance, a random refinement mode.  The library's
assignment (map, maxima in basin order) must equal the oracle's -- the own-trajectory map of methods.neargrid's stepping
rule, or methods.ongrid's map -- and its refinement of that map the oracle's refinement, log and map.  Round 3: 1 800 cases
of this soak found the slow kernel applying the vacuum rule to labels nobody had written (fixed, regression test in
test_gpu_rough.py) and nothing else."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

HEX = np.array([[6.0, 0.0, 0.0], [-3.0, 5.196152422706632, 0.0], [0.0, 0.0, 7.0]])
ORTHO = np.array([[5.0, 0.0, 0.0], [0.0, 6.5, 0.0], [0.0, 0.0, 7.25]])


def run(cases, seed, method='neargrid', odd=False, verbose=False, big=False, oddbig=False):
    """-> descriptions of the cases whose result differs from the oracle's"""
    import oracle
    from pybader_amd import _lib, synth
    from pybader_amd.interface import distance_matrix, gradient_transform
    from rough_common import own_map, rank_labels
    lattices = [('cubic', synth.CUBIC6), ('triclinic', synth.TRICLINIC), ('hexagonal', HEX), ('orthorhombic', ORTHO)]
    rng = np.random.default_rng(seed)
    bad = []
    for k in range(cases):
        shape = tuple(int(rng.choice([16, 24, 32, 40, 48, 64, 72, 96])) for _ in range(3))
        if big:       # (--big: a few seconds of oracle time per case)
            shape = tuple(int(rng.choice([64, 96, 128, 160, 192])) for _ in range(3))
        elif oddbig:  # (--oddbig: grids the brick lattice does not divide, large enough for trapping regions to form)
            shape = tuple(int(rng.integers(41, 150)) for _ in range(3))
        elif odd:
            shape = tuple(int(rng.integers(10, 61)) for _ in range(3))
        elif rng.random() < 0.3:
            shape = (shape[0], shape[1], int(rng.choice([64, 128])))     # whole tiles in z: the tile-wise dilation
        lname, lat = lattices[int(rng.integers(4))]
        noise = float(rng.choice([0.0, 0.0, 1e-6, 1e-3, 3e-2]))
        quant = float(rng.choice([0.0, 0.0, 0.0, 1.0 / 32]))
        tol = [None, None, 0.02, 0.2][int(rng.integers(4))]
        mode, iters = [('changed', 2), ('all', 2), ('changed', -1), ('all', -1)][int(rng.integers(4))]
        vl = np.divide(lat, shape)
        dm, tg = distance_matrix(vl), gradient_transform(vl)
        ctx = _lib.Context(0)
        ctx.set_grid(shape, dm, tg)
        ctx.synth_density(lat, synth.ATOMS8, synth.BACKGROUND)
        rho = ctx.download_density()
        if noise:
            rho = rho + noise * np.random.default_rng(k).random(shape)
        if quant:
            rho = np.round(rho / quant) * quant
        rho = np.ascontiguousarray(rho)
        ctx.upload_density(rho)
        ctx.vacuum_assign(tol, 1.0)
        n = ctx.assign(method)
        got0 = ctx.download_labels(np.int64)
        gmax = np.ravel_multi_index(tuple(ctx.maxima().T), shape) if n else np.zeros(0, np.int64)
        log = ctx.refine(mode, iters)
        got = ctx.download_labels(np.int32)
        ctx.close()
        vol0 = np.zeros(shape, np.int32)
        vol0, _, _ = oracle.vacuum_assign(rho, vol0, float('nan') if tol is None else tol, rho, 1.0)
        if method == 'neargrid':
            lab, maxima = rank_labels(own_map(rho, vol0, dm, tg, main_ties=True))
        else:
            bmax, lab = oracle.bader_calc('ongrid', rho, vol0.copy(), dm, tg, 1)
            maxima = np.ravel_multi_index(tuple(np.asarray(bmax).T), shape) if len(bmax) else np.zeros(0, np.int64)
            lab = lab.astype(np.int64)
        v = lab.astype(np.int32).copy()
        olog = []
        oracle.refine('neargrid', (mode, iters), rho, v, dm, tg, 1, log=olog)
        parts = {'basins': n == len(maxima), 'maxima': np.array_equal(gmax, maxima), 'assignment': np.array_equal(got0, lab),
                 'log': [list(x) for x in log] == [list(x) for x in olog], 'refined map': np.array_equal(got, v)}
        what = f'case {k} of seed {seed}: {method} {shape} {lname} noise {noise} quantum {quant} vacuum_tol {tol} refine {mode}:{iters}, {n} basins'
        if not all(parts.values()):
            bad.append(what + ' -- differs in ' + ', '.join(p for p, ok in parts.items() if not ok))
        if verbose:
            print(('OK  ' if all(parts.values()) else 'BAD ') + what, flush=True)
    return bad


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    method = sys.argv[sys.argv.index('--method') + 1] if '--method' in sys.argv else 'neargrid'
    if '--method' in sys.argv:
        args.remove(method)
    failures = run(int(args[0]) if args else 20, int(args[1]) if len(args) > 1 else 1, method, '--odd' in sys.argv, verbose=True, big='--big' in sys.argv,
                   oddbig='--oddbig' in sys.argv)
    print('\n'.join(failures))
    print('bad cases:', len(failures))
    sys.exit(1 if failures else 0)
