"""Shared by tests/test_oracle_rough.py, tests/test_gpu_rough.py and tests/golden/make_rough_expected.py: the
non-smooth golden cases (reference-generated, tests/golden/make_golden.py ROUGH) and the comparison of this
library's order-independent pipeline with the reference's sequential result."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ROUGH_CASES = ['r40_noise005', 'r40_noise04', 'r64_noise04', 'r48_sig5', 'r48_sig5_noise', 'r32_quant8',
               'r40_vac_noise', 'r36_plateau_vac', 'r30_noise_sig4', 'r48_sig5_vac']
MODES = {'ng_changed_2': ('changed', 2), 'ng_changed_inf': ('changed', -1), 'ng_all_inf': ('all', -1)}


def load_rough(name):
    from pybader_amd import synth
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    kw = json.loads(str(g['rough_json']))
    rho = synth.rough_density(tuple(int(s) for s in g['shape']), g['lattice'], synth.ATOMS8, **kw)
    assert synth.sha256(rho) == str(g['rho_sha256']), 'rough density generator drifted'
    return g, rho


def vac_tol(g):
    t = float(g['vacuum_tol'])
    return None if np.isnan(t) else t


def rank_labels(F):
    """labels from an own-trajectory map (linear index of the maximum per voxel, -1 vacuum): rank of the smallest
    voxel index of each basin; returns (labels int64, maxima in label order)"""
    flat = F.reshape(-1)
    nv = flat >= 0
    maxima, first = np.unique(flat[nv], return_index=True)
    first = np.flatnonzero(nv)[first]
    order = np.argsort(first)
    rank = np.empty(maxima.shape[0], np.int64)
    rank[order] = np.arange(maxima.shape[0])
    lab = np.full(flat.shape, -1, np.int64)
    lab[nv] = rank[np.searchsorted(maxima, flat[nv])]
    return lab.reshape(F.shape), maxima[order]


def own_map(rho, vol0, dm, tg, main_ties):
    """own-trajectory map with the vacuum rule: a trajectory that ends on a vacuum maximum hands -1 to its start"""
    import oracle
    F = oracle.own_trajectory_map(rho, vol0, dm, tg, main_ties=main_ties)
    flat = F.reshape(-1).copy()
    ends_in_vac = (flat >= 0) & (vol0.reshape(-1)[np.maximum(flat, 0)] == -1)
    flat[ends_in_vac] = -1
    return flat.reshape(F.shape)


def pipeline_maps(g, rho):
    """what xb_assign(neargrid) must return for this case (oracle restatement) + the reference's maxima"""
    import oracle
    shape = rho.shape
    vol0 = np.zeros(shape, np.int32)
    tol = vac_tol(g)
    vol0, _, _ = oracle.vacuum_assign(rho, vol0, float('nan') if tol is None else tol, rho, 1.0)
    lab, maxima = rank_labels(own_map(rho, vol0, g['dist_mat'], g['T_grad'], main_ties=True))
    return {'assign': lab, 'maxima': maxima, 'vol0': vol0,
            'ref_maxima': np.ravel_multi_index(tuple(g['ng_bader_max'].T), shape)}


def basin_of(lab, maxima_lin):
    out = np.full(lab.shape, -1, np.int64)
    m = lab >= 0
    out[m] = maxima_lin[lab[m]]
    return out


def refined(g, rho, maps, tag):
    """the oracle's refinement (reference semantics) applied to this library's assignment"""
    import oracle
    v = maps['assign'].astype(np.int32).copy()
    log = []
    oracle.refine('neargrid', MODES[tag], rho, v, g['dist_mat'], g['T_grad'], 1, log=log)
    return v, log


def atoms_map(g, maps, final):
    """voxel -> atom map of `final` (labels in this library's numbering), the way Bader.bader_to_atom_distance builds it
    (interface.py:318-324, 479-484; thread_handlers.py:78-125): maxima -> fractional -> cartesian, nearest atom over the 27
    periodic images, then one lookup per voxel"""
    import oracle
    shape = final.shape
    vox = np.array(np.unravel_index(maps['maxima'], shape), dtype=np.int64).T
    cart = np.dot(np.ascontiguousarray(np.divide(np.add(vox, np.zeros(3)), shape)), g['lattice'])
    bader_atoms, _ = oracle.atom_assign(cart, g['atoms_cart'], g['lattice'])
    f = final.astype(np.int64)
    return np.where(f >= 0, bader_atoms[np.maximum(f, 0)], -1), bader_atoms


def atoms_deviation(g, tag, atoms_vol, rho):
    """north_star's own gate, independent of the basin numbering: voxels whose ATOM differs from the reference's
    atoms_volumes after refine mode `tag`, and whether every per-atom charge / volume agrees within 1e-6 (relative)"""
    ref = g[tag + '_atoms_volumes'].astype(np.int64)
    mine = np.asarray(atoms_vol).astype(np.int64)
    n = g['atoms_cart'].shape[0]
    vv = float(g['voxel_volume'])
    m = mine.reshape(-1) >= 0
    charge = np.bincount(mine.reshape(-1)[m], weights=rho.reshape(-1)[m], minlength=n) * vv
    volume = np.bincount(mine.reshape(-1)[m], minlength=n) * vv
    def close(a, b):
        return bool(np.all(np.abs(a - b) <= 1e-6 * np.maximum(np.abs(b), 1e-300)))
    return {'atoms_diff': int((ref != mine).sum()),
            'atoms_charge_volume_within_1e-6': close(charge, g[tag + '_atoms_charge']) and close(volume, g[tag + '_atoms_volume'])}


def compare(g, maps, tag, final, log, rho=None):
    """deviation of `final` (this library's labels after refine mode `tag`) from the reference's final map"""
    ref = g[tag].astype(np.int64)
    ref_b = basin_of(ref, maps['ref_maxima'])
    mine_b = basin_of(final.astype(np.int64), maps['maxima'])
    out = {'basin_diff': int((ref_b != mine_b).sum()), 'label_diff': int((ref != final).sum()),
           'log': [[int(a), int(b)] for a, b in log], 'ref_log': g[tag + '_log'].tolist()}
    if rho is not None:
        out.update(atoms_deviation(g, tag, atoms_map(g, maps, final)[0], rho))
    return out


def deviation(g, maps, tag, rho=None):
    if rho is None:
        _, rho = load_rough_cached(g)
    final, log = refined(g, rho, maps, tag)
    return compare(g, maps, tag, final, log, rho)


_cache = {}


def load_rough_cached(g):
    key = str(g['rho_sha256'])
    if key not in _cache:
        from pybader_amd import synth
        kw = json.loads(str(g['rough_json']))
        _cache[key] = (g, synth.rough_density(tuple(int(s) for s in g['shape']), g['lattice'], synth.ATOMS8, **kw))
    return _cache[key]


def expected():
    with open(os.path.join(GOLDEN, 'rough_expected.json')) as f:
        return json.load(f)
