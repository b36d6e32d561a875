"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against (a) golden vectors
captured from the reference and (b) the CPU oracle on seeded inputs.  Integer maps bit-exact;
charges/volumes to 1e-9 relative (north_star asks 1e-6)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, case_density, load_golden
from pybader_amd import _lib, synth
from pybader_amd.utils import dtype_calc

pytestmark = pytest.mark.gpu

FULL = ['c12_cubic', 'c64_cubic', 'c40x48x56_tric', 'c48_cubic_vac']
MODES = {'ng_changed_2': ('changed', 2), 'ng_changed_inf': ('changed', -1), 'ng_all_inf': ('all', -1),
         'ng_all_2': ('all', 2)}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope='module')
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def setup_case(ctx, name):
    g = load_golden(name)
    rho = case_density(g)
    ctx.set_grid(rho.shape, g['dist_mat'], g['T_grad'])
    ctx.upload_density(rho)
    return g, rho


def vac_tol(g):
    t = float(g['vacuum_tol'])
    return None if np.isnan(t) else t


@pytest.mark.parametrize('name', FULL)
def test_device_synth_is_bit_identical(ctx, name):
    g = load_golden(name)
    rho = case_density(g)
    ctx.set_grid(rho.shape, g['dist_mat'], g['T_grad'])
    ctx.synth_density(g['lattice'], g['atoms'], float(g['background']))
    assert np.array_equal(ctx.download_density(), rho)


@pytest.mark.parametrize('name', FULL)
def test_vacuum_assign(ctx, name):
    g, rho = setup_case(ctx, name)
    vc, vv = ctx.vacuum_assign(vac_tol(g), float(g['voxel_volume']))
    assert np.array_equal(ctx.download_labels(np.int8), g['ng_init'])
    np.testing.assert_allclose([vc, vv], [float(g['vacuum_charge']), float(g['vacuum_volume'])], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('name', FULL)
def test_neargrid_assign_is_the_post_refinement_map(ctx, name):
    """xb_assign(neargrid) == the reference's map after refinement (== its own-trajectory map F),
    including the basin numbering and the maxima list."""
    g, rho = setup_case(ctx, name)
    ctx.vacuum_assign(vac_tol(g), float(g['voxel_volume']))
    n = ctx.assign('neargrid')
    assert n == g['ng_bader_max'].shape[0]
    assert np.array_equal(ctx.maxima(), g['ng_bader_max'])
    lab = ctx.download_labels(g['ng_F'].dtype)
    assert np.array_equal(lab, g['ng_F'])
    for key in MODES:
        assert np.array_equal(lab, g[key]), key
    # pre-refinement map of the sequential reference differs only where its refinement changes it
    ndiff = int((lab != g['ng_main']).sum())
    assert ndiff == int(g['ng_changed_2_log'][0, 1])


@pytest.mark.parametrize('name', FULL)
def test_edge_find_on_reference_main_map(ctx, name):
    g, rho = setup_case(ctx, name)
    ctx.upload_labels(g['ng_main'])
    edges = ctx.edge_find()
    assert edges == int(g['ng_changed_2_log'][0, 0])
    assert np.array_equal(ctx.download_known(), g['ng_known0'])


@pytest.mark.parametrize('name', FULL)
@pytest.mark.parametrize('key', list(MODES))
def test_refine_from_reference_main_map(ctx, name, key):
    """Start from the reference's sequential (order dependent) main-pass map and refine on the GPU:
    per-iteration (edges, changed) counts and the final map must equal the reference's."""
    g, rho = setup_case(ctx, name)
    ctx.upload_labels(g['ng_main'])
    log = ctx.refine(*MODES[key])
    assert np.array_equal(np.array(log, np.int64).reshape(-1, 2), g[key + '_log'])
    assert np.array_equal(ctx.download_labels(g[key].dtype), g[key])
    if key == 'ng_changed_2':
        assert np.array_equal(ctx.download_known(), g['ng_changed_2_known_last'])


@pytest.mark.parametrize('name', FULL)
def test_full_pipeline_assign_refine_sums_atoms(ctx, name):
    g, rho = setup_case(ctx, name)
    vv = float(g['voxel_volume'])
    ctx.vacuum_assign(vac_tol(g), vv)
    n = ctx.assign('neargrid')
    log = ctx.refine('changed', 2)
    assert all(ch == 0 for _, ch in log)                # F is a fixed point of the refinement
    lab = ctx.download_labels(g['ng_changed_2'].dtype)
    assert np.array_equal(lab, g['ng_changed_2'])
    ch, vo = ctx.charge_sum(vv, n)
    np.testing.assert_allclose(ch, g['ng_bader_charge'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(vo, g['ng_bader_volume'], rtol=1e-9, atol=1e-12)
    atoms_cart = synth.atoms_cartesian(g['atoms'], g['lattice'])
    ba, bd = _lib.atom_assign(g['bader_maxima_cart'], atoms_cart, g['lattice'])
    assert np.array_equal(ba, g['ng_bader_atoms'])
    np.testing.assert_allclose(bd, g['ng_bader_distance'], rtol=1e-12, atol=1e-12)
    ctx.volume_assign(ba)
    assert np.array_equal(ctx.download_labels(g['ng_atoms_volumes'].dtype), g['ng_atoms_volumes'])
    ach, avo = ctx.charge_sum(vv, atoms_cart.shape[0])
    np.testing.assert_allclose(ach, g['ng_atoms_charge'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(avo, g['ng_atoms_volume'], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('name', FULL)
def test_ongrid_and_speed_profile(ctx, name):
    g, rho = setup_case(ctx, name)
    ctx.vacuum_assign(vac_tol(g), float(g['voxel_volume']))
    n = ctx.assign('ongrid')
    assert np.array_equal(ctx.maxima(), g['og_bader_max'])
    assert np.array_equal(ctx.download_labels(g['og_main'].dtype), g['og_main'])
    # BASELINE config 5: ongrid assign + neargrid edge refinement
    log = ctx.refine('changed', 2)
    assert np.array_equal(np.array(log, np.int64).reshape(-1, 2), g['og_ngrefine_changed_2_log'])
    assert np.array_equal(ctx.download_labels(g['og_main'].dtype), g['og_ngrefine_changed_2'])
    # speed profile (entry_points.py:340-345): refine the int8 atom map with ('changed', 3)
    ctx.upload_labels(g['og_atoms_volumes_pre'])
    ctx.refine('changed', 3)
    assert np.array_equal(ctx.download_labels(np.int8), g['og_atoms_volumes_speed'])


@pytest.mark.parametrize('name', ['c128_tric', 'c128_216atoms', 'c256_cubic', 'c320_tric', 'c512_cubic'])   # (c320_tric, round 5: the generic Grid instantiations above 128^3)
def test_large_golden_hashes(ctx, name):
    g = load_golden(name)
    shape = tuple(int(s) for s in g['shape'])
    ctx.set_grid(shape, g['dist_mat'], g['T_grad'])
    ctx.synth_density(g['lattice'], g['atoms'], float(g['background']))
    vv = float(g['voxel_volume'])
    ctx.vacuum_assign(None, vv)
    n = ctx.assign('neargrid')
    assert np.array_equal(ctx.maxima(), g['ng_bader_max'])
    dt = np.dtype(dtype_calc(-n))        # what the reference narrows its map to (int8 up to 63 basins, int16 for the 216 atoms)
    assert sha(ctx.download_labels(dt)) == str(g['ng_F_sha256'])
    log = ctx.refine('changed', 2)
    assert all(ch == 0 for _, ch in log)
    assert sha(ctx.download_labels(dt)) == str(g['ng_changed_2_sha256'])
    ch, vo = ctx.charge_sum(vv, n)
    np.testing.assert_allclose(ch, g['ng_bader_charge'], rtol=1e-9)
    np.testing.assert_allclose(vo, g['ng_bader_volume'], rtol=1e-9)
    ctx.vacuum_assign(None, vv)
    ctx.assign('ongrid')
    assert sha(ctx.download_labels(dt)) == str(g['og_main_sha256'])
    log = ctx.refine('changed', 2)
    assert np.array_equal(np.array(log, np.int64).reshape(-1, 2), g['og_ngrefine_changed_2_log'])
    assert sha(ctx.download_labels(dt)) == str(g['og_ngrefine_changed_2_sha256'])


def test_contexts_sharing_the_card_at_the_same_time():
    """Round 5: the chase of edge_check (k_ec_chase) passes work between its workgroups through mailboxes, and a workgroup may only
    assume that the OTHERS exist, not that they are running: four contexts (a stream and a host thread each) run config 5 at 256^3 at
    the same time, three times over -- their 1024-thread workgroups compete for the compute units, launches start piecemeal -- and
    every one must still give the reference's log and map.  (The first version of the sharing waited for a count that included
    workgroups which could not start before the waiting ones had left: tests/test_gpu_slabs.py caught it.)"""
    import threading
    g = load_golden('c256_cubic')
    shape = tuple(int(s) for s in g['shape'])
    errors, results = [], []

    def work(i):
        try:
            c = _lib.Context(0)
            c.set_grid(shape, g['dist_mat'], g['T_grad'])
            c.synth_density(g['lattice'], g['atoms'], float(g['background']))
            for _ in range(3):
                c.set_option(6, 1)
                c.vacuum_assign(None, float(g['voxel_volume']))
                c.assign('ongrid')
                log = c.refine('changed', 2)
                results.append((np.array(log, np.int64).reshape(-1, 2).tolist(), sha(c.download_labels(np.int8))))
            c.close()
        except Exception as e:   # noqa: BLE001 (reported below, in the test's thread)
            errors.append(repr(e))
    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(results) == 12
    for log, h in results:
        assert log == g['og_ngrefine_changed_2_log'].tolist()
        assert h == str(g['og_ngrefine_changed_2_sha256'])


def test_chase_when_mailboxes_are_full_or_closed(tmp_path):
    """The sharing of k_ec_chase must give the same decisions whatever happens to a shed entry: here the library is built with
    mailboxes of 64 slots (8192) that close after 40 idle rounds (32768) -- nine of ten sheds then find the target's mailbox full or
    closed and stay with the sender, workgroups leave early and late -- and runs config 5 at 256^3 and a sheared 128^3 in a child
    process (the library of THIS process is loaded already).  The child also says how many entries travelled: the paths were taken."""
    import subprocess
    import sys
    from pybader_amd import build
    lib = tmp_path / 'libbader_hip_smallbox.so'
    subprocess.check_call([build.hipcc()] + [f for f in build.FLAGS if f != '-Wall'] + ['-DEC_MB_CAP=64', '-DEC_LINGER=40', '-o', str(lib), build.SRC],
                          stderr=subprocess.DEVNULL)
    script = f"""
import sys, hashlib
import numpy as np
sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r}); sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
from conftest import load_golden
from pybader_amd import _lib
assert _lib.LIB_PATH == {str(lib)!r}
for name in ('c256_cubic', 'c128_tric'):
    g = load_golden(name)
    c = _lib.Context(0)
    c.set_grid(tuple(int(s) for s in g['shape']), g['dist_mat'], g['T_grad'])
    c.synth_density(g['lattice'], g['atoms'], float(g['background']))
    c.set_option(3, 4)
    c.vacuum_assign(None, float(g['voxel_volume']))
    c.assign('ongrid')
    log = c.refine('changed', 2)
    assert np.array_equal(np.array(log, np.int64).reshape(-1, 2), g['og_ngrefine_changed_2_log']), (name, log)
    assert hashlib.sha256(np.ascontiguousarray(c.download_labels(np.int8)).tobytes()).hexdigest() == str(g['og_ngrefine_changed_2_sha256']), name
    c.close()
print('ok')
"""
    out = subprocess.run([sys.executable, '-c', script], env=dict(os.environ, XB_LIBRARY=str(lib)), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith('ok'), out.stderr[-3000:]
    import re
    pairs = [(int(a), int(b)) for a, b in re.findall(r'edge_check sharing: (\d+) entries shed, (\d+) received', out.stderr)]
    assert pairs and any(shed > got > 0 for shed, got in pairs), pairs   # some sheds bounced off a full or closed mailbox, some travelled


# ---- seeded inputs against the CPU oracle ------------------------------------------------------
def random_case(seed, shape):
    rng = np.random.default_rng(seed)
    lat = synth.CUBIC6 * (0.8 + 0.4 * rng.random()) + 0.8 * (rng.random((3, 3)) - 0.5)
    na = int(rng.integers(3, 12))
    atoms = np.concatenate([rng.random((na, 3)), 0.3 + 0.3 * rng.random((na, 1)), 1 + 7 * rng.random((na, 1))], axis=1)
    rho = synth.synth_density(shape, lat, atoms)
    from pybader_amd.interface import distance_matrix, gradient_transform
    vl = np.divide(lat, shape)
    return rho, lat, atoms, distance_matrix(vl), gradient_transform(vl)


def rank_labels(F):
    """labels from the own-trajectory map: rank of the smallest voxel index of each basin"""
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


@pytest.mark.parametrize('seed,shape,tol', [(1, (24, 20, 28), None), (2, (33, 17, 19), None), (3, (30, 30, 30), 0.05),
                                            (4, (48, 40, 36), None), (5, (16, 64, 9), 0.04)])
def test_against_oracle_seeded(ctx, seed, shape, tol):
    import oracle
    rho, lat, atoms, dm, tg = random_case(seed, shape)
    ctx.set_grid(shape, dm, tg)
    ctx.upload_density(rho)
    vol0 = np.zeros(shape, np.int32)
    vol0, _, _ = oracle.vacuum_assign(rho, vol0, float('nan') if tol is None else tol, rho, 1.0)
    ctx.vacuum_assign(tol, 1.0)
    assert np.array_equal(ctx.download_labels(np.int32), vol0)
    # neargrid: own-trajectory map
    F = oracle.own_trajectory_map(rho, vol0, dm, tg, main_ties=True)
    want, maxima = rank_labels(F)
    n = ctx.assign('neargrid')
    got = ctx.download_labels(np.int64)
    assert n == maxima.shape[0]
    assert np.array_equal(np.ravel_multi_index(tuple(ctx.maxima().T), shape), maxima)
    assert np.array_equal(got, want)
    # the sequential reference path ends on the same map
    bmax, main = oracle.bader_calc('neargrid', rho, vol0, dm, tg, 1)
    v = main.copy()
    oracle.refine('neargrid', ('all', -1), rho, v, dm, tg, 1)
    assert np.array_equal(v.astype(np.int64), want), "oracle: sequential main+refine != own-trajectory map"
    # refinement kernels from the order-dependent sequential main map, all modes
    for mode in (('changed', 2), ('changed', -1), ('all', -1), ('all', 1), ('changed', 3)):
        v = main.copy()
        olog = []
        oracle.refine('neargrid', mode, rho, v, dm, tg, 1, log=olog)
        ctx.upload_labels(main)
        glog = ctx.refine(*mode)
        assert glog == [tuple(x) for x in olog], mode
        assert np.array_equal(ctx.download_labels(main.dtype), v), mode
    # ongrid + refinement from the ongrid map (many changed voxels: exercises edge_check)
    bmax, omain = oracle.bader_calc('ongrid', rho, vol0, dm, tg, 1)
    ctx.upload_labels(vol0)
    ctx.assign('ongrid')
    assert np.array_equal(ctx.maxima(), bmax)
    assert np.array_equal(ctx.download_labels(omain.dtype), omain)
    for mode in (('changed', 4), ('all', 3), ('changed', -1)):
        v = omain.copy()
        olog = []
        oracle.refine('neargrid', mode, rho, v, dm, tg, 1, log=olog)
        ctx.upload_labels(omain)
        glog = ctx.refine(*mode)
        assert glog == [tuple(x) for x in olog], mode
        assert np.array_equal(ctx.download_labels(omain.dtype), v), mode


@pytest.mark.parametrize('groups,qcap', [(1, 2), (2, 16), (64, 6000)])
def test_edge_check_queue_overflow_hand_over(ctx, groups, qcap):
    """The chase kernel's LDS queues, shrunk to a few entries, spill into the overflow list that seeds
    the next launch: logs, map and flags must still equal the reference's (golden 'changed' run)."""
    g, rho = setup_case(ctx, 'c64_cubic')
    ctx.set_option(4, groups)
    ctx.set_option(5, qcap)
    try:
        ctx.upload_labels(g['ng_main'])
        log = ctx.refine('changed', -1)
        assert np.array_equal(np.array(log, np.int64).reshape(-1, 2), g['ng_changed_inf_log'])
        assert np.array_equal(ctx.download_labels(g['ng_changed_inf'].dtype), g['ng_changed_inf'])
    finally:
        ctx.set_option(4, 256)
        ctx.set_option(5, 6000)


def test_atom_assign_matches_oracle(ctx):
    """k_atom_assign against the oracle, bit for bit (assigned atom and distance), incl. the reference's
    carried-over `pbc` vector (utils.py:199)."""
    import oracle
    rng = np.random.default_rng(3)
    lat = np.array([[6.0, 0.0, 0.0], [1.5, 5.5, 0.0], [0.7, 1.1, 6.2]])
    atoms = rng.random((7, 3)) @ lat
    bmax = rng.random((400, 3)) @ lat
    a, d = _lib.atom_assign(bmax, atoms, lat)
    a2, d2 = oracle.atom_assign(bmax, atoms, lat)
    assert np.array_equal(a, a2) and np.array_equal(d, d2)


def test_edge_check_kernel_level(ctx):
    """edge_check alone on a hand-made `known`: every voxel of a random subset flagged changed."""
    import oracle
    rho, lat, atoms, dm, tg = random_case(11, (20, 22, 18))
    vol0 = np.zeros(rho.shape, np.int32)
    _, main = oracle.bader_calc('ongrid', rho, vol0, dm, tg, 1)
    known = np.zeros(rho.shape, np.int8)
    oracle.edge_find(known, rho, main)
    rng = np.random.default_rng(5)
    known[(known == -1) & (rng.random(rho.shape) < 0.3)] = -2     # pretend these changed
    ctx.set_grid(rho.shape, dm, tg)
    ctx.upload_density(rho)
    ctx.upload_labels(main)
    ctx.upload_known(known)
    got = ctx.edge_check()
    k2 = known.copy()
    want = oracle.edge_check(k2, rho, main)
    assert got == want
    assert np.array_equal(ctx.download_known(), k2)


def test_python_mirror_api(ctx):
    """The reference-named Python layer (thread_handlers / interface) end to end on one golden case."""
    from pybader_amd import thread_handlers
    from pybader_amd.interface import Bader
    thread_handlers.VERBOSE = False
    g = load_golden('c64_cubic')
    rho = case_density(g)
    atoms_cart = synth.atoms_cartesian(g['atoms'], g['lattice'])
    b = Bader({'charge': rho}, g['lattice'], atoms_cart)
    assert np.array_equal(b.distance_matrix, g['dist_mat']) and np.array_equal(b.T_grad, g['T_grad'])
    b()
    assert b.bader_volumes.dtype == g['ng_changed_2'].dtype
    assert np.array_equal(b.bader_volumes, g['ng_changed_2'])
    assert np.array_equal(b.atoms_volumes, g['ng_atoms_volumes'])
    assert np.array_equal(b.bader_atoms, g['ng_bader_atoms'])
    np.testing.assert_allclose(b.bader_maxima, g['bader_maxima_cart'], rtol=1e-14)
    np.testing.assert_allclose(b.bader_charge, g['ng_bader_charge'], rtol=1e-9)
    np.testing.assert_allclose(b.atoms_charge, g['ng_atoms_charge'], rtol=1e-9)
    np.testing.assert_allclose(b.atoms_volume, g['ng_atoms_volume'], rtol=1e-9)
    # kernel-level plugin signatures
    from pybader_amd import methods, refinement
    vol = np.zeros(rho.shape, np.int32)
    vol, bmax, emax = methods.neargrid(rho, vol, np.zeros(3, np.int64), g['dist_mat'], g['T_grad'], np.zeros(1, np.int64))
    assert np.array_equal(vol - 1, g['ng_F']) and np.array_equal(bmax, g['ng_bader_max']) and emax.shape == (0, 3)
    known = np.zeros(rho.shape, np.int8)
    main = g['ng_main'].copy()
    assert refinement.edge_find(known, rho, main) == int(g['ng_changed_2_log'][0, 0])
    known, changed = refinement.neargrid(known, known.copy(), rho, main, np.zeros(3, np.int64), g['dist_mat'],
                                         g['T_grad'], np.zeros(1, np.int64))
    assert changed == int(g['ng_changed_2_log'][0, 1])
    # the reference's silent-return cases (thread_handlers.py:140-147)
    v = g['ng_main'].copy()
    assert thread_handlers.refine('ongrid', ('changed', 2), rho, v, g['dist_mat'], g['T_grad'], 1) is None
    assert thread_handlers.refine('neargrid', ('changed', 0), rho, v, g['dist_mat'], g['T_grad'], 1) is None
    assert np.array_equal(v, g['ng_main'])


@pytest.mark.parametrize('kw', [dict(), dict(vacuum_tol=1e-3), dict(method='ongrid'), dict(refine_mode=('all', 1)),
                                dict(refine_mode=('changed', 0)), dict(refine_mode=('changed', -1), vacuum_tol=1e-3)])
def test_bader_run_fused_equals_the_two_calls(ctx, kw):
    """Bader._run issues bader_calc + refine as ONE library call (thread_handlers.bader_calc_refine) -- every result
    slot must equal what the reference's two calls leave."""
    from pybader_amd import thread_handlers
    from pybader_amd.interface import Bader
    thread_handlers.VERBOSE = False
    g = load_golden('c48_cubic_vac' if 'vacuum_tol' in kw else 'c64_cubic')
    rho = case_density(g)
    atoms_cart = synth.atoms_cartesian(g['atoms'], g['lattice'])
    out = []
    for fused in (True, False):
        b = Bader({'charge': rho}, g['lattice'], atoms_cart, **kw)
        b.fused = fused
        b()
        out.append(b)
    a, b = out
    assert a.bader_volumes.dtype == b.bader_volumes.dtype and np.array_equal(a.bader_volumes, b.bader_volumes)
    assert np.array_equal(a.atoms_volumes, b.atoms_volumes) and np.array_equal(a.bader_atoms, b.bader_atoms)
    for slot in ('bader_maxima', 'bader_distance', 'atoms_surface_distance'):
        assert np.array_equal(getattr(a, slot), getattr(b, slot)), slot
    for slot in ('bader_charge', 'bader_volume', 'atoms_charge', 'atoms_volume'):      # segmented sums: the order of the additions is not fixed
        np.testing.assert_allclose(getattr(a, slot), getattr(b, slot), rtol=1e-9, err_msg=slot)
    assert (a.vacuum_charge, a.vacuum_volume) == (b.vacuum_charge, b.vacuum_volume)
    if not kw:
        assert np.array_equal(a.bader_volumes, g['ng_changed_2'])


@pytest.mark.parametrize('name', FULL)
def test_surface_distance(ctx, name):
    """thread_handlers.surface_distance on the reference's atom map."""
    g, rho = setup_case(ctx, name)
    atoms_cart = synth.atoms_cartesian(g['atoms'], g['lattice'])
    ctx.upload_labels(g['ng_atoms_volumes'])
    d2, edges = ctx.surface_distance(g['lattice'], atoms_cart)
    want = g['ng_atoms_surface_distance']
    got = np.where(np.isfinite(d2), np.sqrt(d2), 0.0)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


def test_speed_profile_and_extras_through_python_mirror(ctx):
    """speed profile (entry_points.py:340-345) end to end, volume_mask, vacuum sums with a separate reference."""
    from pybader_amd import thread_handlers, utils
    from pybader_amd.interface import Bader
    thread_handlers.VERBOSE = False
    g = load_golden('c64_cubic')
    rho = case_density(g)
    atoms_cart = synth.atoms_cartesian(g['atoms'], g['lattice'])
    b = Bader({'charge': rho}, g['lattice'], atoms_cart, method='ongrid', refine_method='neargrid',
              refine_mode=('changed', 3), speed_flag=True)
    b()
    assert not hasattr(b, 'bader_volumes')
    assert np.array_equal(b.bader_atoms, g['og_bader_atoms'])
    assert b.atoms_volumes.dtype == g['og_atoms_volumes_speed'].dtype
    assert np.array_equal(b.atoms_volumes, g['og_atoms_volumes_speed'])
    np.testing.assert_allclose(b.atoms_charge, g['og_atoms_charge_speed'], rtol=1e-9)
    np.testing.assert_allclose(b.atoms_volume, g['og_atoms_volume_speed'], rtol=1e-9)
    # default profile also fills the surface distances
    b2 = Bader({'charge': rho}, g['lattice'], atoms_cart)
    b2()
    np.testing.assert_allclose(b2.atoms_surface_distance, g['ng_atoms_surface_distance'], rtol=1e-12)
    # volume_mask
    m = utils.volume_mask(b2.bader_volumes, rho, 3)
    assert np.array_equal(m, np.where(b2.bader_volumes == 3, rho, 0.0))
    # vacuum decided on a reference density, charge summed on another (bader -ref)
    ref = rho * 0.5 + 0.01
    vol = np.zeros(rho.shape, np.int32)
    vol, vc, vv = utils.vacuum_assign(ref, vol, 0.03, rho, 0.25)
    mask = ref <= 0.03
    assert np.array_equal(vol == -1, mask) and mask.any()
    np.testing.assert_allclose(vc, rho[mask].sum() * 0.25, rtol=1e-9)
    np.testing.assert_allclose(vv, mask.sum() * 0.25, rtol=1e-12)
    utils.forget_density(_lib.default_context())


@pytest.mark.parametrize('kind,shape', [('noise', (24, 28, 20)), ('quantised', (32, 24, 24)), ('noise64', (40, 40, 40)),
                                        ('flat_vacuum', (24, 24, 24))])
def test_rough_densities_against_oracle(ctx, kind, shape):
    """Non-smooth inputs: hundreds of maxima (no trapping boxes), exact ties and plateaus (stay voxels,
    ongrid steps, non-monotone paths -> the window/slow-path logic), flat vacuum."""
    import oracle
    rho, lat, atoms, dm, tg = random_case(7, shape)
    rng = np.random.default_rng(99)
    tol = None
    if kind.startswith('noise'):
        rho = rho + 0.4 * rng.random(shape)
    elif kind == 'quantised':
        rho = np.round(rho * 8) / 8          # plateaus of exactly equal density
    else:
        rho = np.where(rho < 0.2, 0.0, rho)  # a flat zero region, declared vacuum
        tol = 0.0
    rho = np.ascontiguousarray(rho)
    ctx.set_grid(shape, dm, tg)
    ctx.upload_density(rho)
    vol0 = np.zeros(shape, np.int32)
    vol0, _, _ = oracle.vacuum_assign(rho, vol0, float('nan') if tol is None else tol, rho, 1.0)
    ctx.vacuum_assign(tol, 1.0)
    assert np.array_equal(ctx.download_labels(np.int32), vol0)
    F = oracle.own_trajectory_map(rho, vol0, dm, tg, main_ties=True)   # xb_assign follows methods.py:324's tie test
    # a trajectory that ends on a vacuum maximum hands -1 to its start voxel (refinement.py:286)
    flat = F.reshape(-1).copy()
    ends_in_vac = (flat >= 0) & (vol0.reshape(-1)[np.maximum(flat, 0)] == -1)
    flat[ends_in_vac] = -1
    want, maxima = rank_labels(flat.reshape(shape))
    n = ctx.assign('neargrid')
    assert n == maxima.shape[0]
    assert np.array_equal(np.ravel_multi_index(tuple(ctx.maxima().T), shape), maxima)
    assert np.array_equal(ctx.download_labels(np.int64), want)
    # refinement from the sequential reference main map, and ongrid
    bmax, main = oracle.bader_calc('neargrid', rho, vol0, dm, tg, 1)
    for mode in (('changed', 3), ('all', 2)):
        v = main.copy()
        olog = []
        oracle.refine('neargrid', mode, rho, v, dm, tg, 1, log=olog)
        ctx.upload_labels(main)
        assert ctx.refine(*mode) == [tuple(x) for x in olog], mode
        assert np.array_equal(ctx.download_labels(main.dtype), v), mode
    bmax, omain = oracle.bader_calc('ongrid', rho, vol0, dm, tg, 1)
    ctx.upload_labels(vol0)
    ctx.assign('ongrid')
    assert np.array_equal(ctx.maxima(), bmax)
    assert np.array_equal(ctx.download_labels(omain.dtype), omain)
    print('slow-path totals (assign, refine):', ctx.slow_path_stats())


def test_slow_path_was_exercised(ctx):
    """runs after the rough-density cases (same module-scoped context): the exact slow kernel must have
    seen work, otherwise that code would be untested"""
    a, r = ctx.slow_path_stats()
    assert a > 0, (a, r)


@pytest.mark.parametrize('key,cname,kw,mode', [
    ('c40x48x56_tric_volumes_0', 'c40x48x56_tric', {}, ('volumes', [0, 3])),
    ('c40x48x56_tric_atoms_-2', 'c40x48x56_tric', {}, ('atoms', [-2])),
    ('c48_cubic_vac_volumes_-2', 'c48_cubic_vac', {'vacuum_tol': 0.03}, ('volumes', [-2])),
])
def test_export_path_equals_the_reference(ctx, key, cname, kw, mode):
    """Bader.write_volume + the export loop (interface.py:417-436, 600-621) through the mirror: file names, comments,
    fortran_format and the masked charge / spin densities (utils.volume_mask on the GPU) equal what the reference
    handed to its writer (tests/golden/export_volumes.npz: sha256 of the arrays)."""
    import hashlib
    import json
    from pybader_amd import thread_handlers
    from pybader_amd.interface import Bader
    thread_handlers.VERBOSE = False
    want = json.loads(str(load_golden('export_volumes')[key]))
    g = load_golden(cname)
    rho = case_density(g)
    spin = np.ascontiguousarray(rho[::-1] * 0.25)
    calls = []

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()

    def capture(fname, atoms, lat, density, info, prefix='', **k):
        calls.append([fname, info['comment'], info['fortran_format'], sha(density['charge']), sha(density['spin']),
                      float(density['charge'].sum())])
    info = {'filename': 'synth', 'prefix': '', 'write_function': capture, 'voxel_offset': np.zeros(3)}
    b = Bader({'charge': rho, 'spin': spin}, g['lattice'], synth.atoms_cartesian(g['atoms'], g['lattice']), info, **kw)
    b.export_mode, b.fortran_format = mode, 2
    b()
    assert len(calls) == len(want)
    for got, ref in zip(calls, want):
        assert got[:5] == ref[:5], (got[:3], ref[:3])
        assert got[5] == pytest.approx(ref[5], rel=1e-12)


OUTPUT_SLOTS = ('_bader_maxima', 'bader_charge', 'bader_volume', 'bader_spin', 'bader_volumes', 'bader_atoms', 'bader_distance',
                'atoms_charge', 'atoms_volume', 'atoms_spin', 'atoms_volumes', 'atoms_surface_distance')


@pytest.mark.parametrize('profile', ['default', 'default_vacuum_spin', 'speed'])
def test_pickle_attribute_contract(profile):
    """SURVEY.md 8(b) last row: the slots the REAL pybader.interface.Bader pickles after __call__ -- recorded
    from a reference run, name / dtype / shape / contiguity (tests/golden/pickle_contract.json, make_golden.py
    run_contract_case) -- against what the drop-in leaves on its object for the same input and profile: every output slot
    the reference stores must be there with exactly that dtype and shape, and no output slot the reference does not store
    (`bader_volumes` is deleted in the speed profile, the spin sums only exist with spin_flag)."""
    import json
    from pybader_amd import thread_handlers
    from pybader_amd.interface import Bader
    thread_handlers.VERBOSE = False
    with open(os.path.join(GOLDEN, 'pickle_contract.json')) as f:
        contract = json.load(f)
    rec = contract['profiles'][profile]
    conf = {k: (tuple(v) if isinstance(v, list) else v) for k, v in rec['config'].items()}
    g = load_golden('c40x48x56_tric')
    assert list(g['shape']) == contract['shape']
    rho = case_density(g)
    spin = np.ascontiguousarray(rho[::-1] * 0.25)
    b = Bader({'charge': rho, 'spin': spin}, g['lattice'], synth.atoms_cartesian(g['atoms'], g['lattice']), **conf)
    b()
    assert b._bader_maxima.shape[0] == rec['n_maxima']
    for slot in OUTPUT_SLOTS:
        want = rec['slots_set'].get(slot)
        if want is None:
            assert not hasattr(b, slot), f'{slot}: the reference does not store it in the {profile} profile'
            continue
        got = getattr(b, slot)
        assert isinstance(got, np.ndarray), slot
        assert got.dtype.str == want['dtype'] and list(got.shape) == want['shape'], (slot, got.dtype.str, got.shape, want)
        assert got.flags.c_contiguous == want['c_contiguous'], slot
    for name in ('_vacuum_charge', '_vacuum_volume'):     # plain Python floats in the pickle
        want = rec['slots_set'][name]
        got = getattr(b, name[1:])
        assert isinstance(got, float) and want['type'] == 'float'
        assert abs(got - want['value']) <= 1e-9 * max(1.0, abs(want['value'])), (name, got, want['value'])


def test_label_array_stays_on_the_device_inside_resident():
    """inside `utils.resident(density)` the label array handed from bader_calc to refine to charge_sum to
    assign_to_atoms is recognised by identity and neither uploaded nor (when nothing changed) downloaded again; it is
    read-only while the device holds its twin; outside the block every call transfers as before.  Outputs identical."""
    from pybader_amd import thread_handlers, utils
    thread_handlers.VERBOSE = False
    g = load_golden('c64_cubic')
    rho = case_density(g)
    dm, tg = g['dist_mat'], g['T_grad']
    atoms_cart = synth.atoms_cartesian(g['atoms'], g['lattice'])
    ctx = _lib.default_context()
    counts = {'up': 0, 'down': 0, 'rho': 0}
    real_up, real_down, real_rho = ctx.upload_labels, ctx.download_labels, ctx.upload_density

    def up(labels):
        counts['up'] += 1
        return real_up(labels)

    def down(*a, **k):
        counts['down'] += 1
        return real_down(*a, **k)

    def up_rho(r):
        counts['rho'] += 1
        return real_rho(r)
    ctx.upload_labels, ctx.download_labels, ctx.upload_density = up, down, up_rho
    try:
        def flow():
            vol = np.zeros(rho.shape, np.int32)
            vol, vc, vv = utils.vacuum_assign(rho, vol, float('nan'), rho, 1.0)
            bmax, vol = thread_handlers.bader_calc('neargrid', rho, vol, dm, tg, 1)
            thread_handlers.refine('neargrid', ('changed', 2), rho, vol, dm, tg, 1)
            ch, vo = np.zeros(bmax.shape[0]), np.zeros(bmax.shape[0])
            utils.charge_sum(ch, vo, 1.0, rho, vol)
            frac = bmax / np.array(rho.shape)
            ba, bd, av = thread_handlers.assign_to_atoms(np.dot(frac, g['lattice']), atoms_cart, g['lattice'], vol, 1)
            ach, avo = np.zeros(atoms_cart.shape[0]), np.zeros(atoms_cart.shape[0])
            utils.charge_sum(ach, avo, 1.0, rho, av)
            return vol, ch, ba, av, ach
        bare = flow()
        n_bare = dict(counts)
        for k in counts:
            counts[k] = 0
        with utils.resident(rho):
            inside = flow()
            tracked = inside[3]
            assert not tracked.flags.writeable            # the atom map is the device's twin right now
            with pytest.raises(ValueError):
                tracked[0, 0, 0] = 5
        n_in = dict(counts)
        assert tracked.flags.writeable and ctx.resident_labels is None     # released with the block
        for a, b in zip(bare, inside):
            assert a.dtype == b.dtype
            if a.dtype.kind == 'f':        # segmented sums: the summation order differs from run to run
                np.testing.assert_allclose(a, b, rtol=1e-12)
            else:
                assert np.array_equal(a, b)
        assert np.array_equal(inside[0], g['ng_changed_2']) and np.array_equal(inside[3], g['ng_atoms_volumes'])
        # bare: density 5 times (vacuum, assign, refine, 2 sums), labels up 5 / down 2 (an unchanged refinement is not fetched); inside: density once, labels never
        # uploaded (vacuum_assign's zeros are the device's zeros), downloaded twice (the narrowed map, the atom map)
        assert n_bare == {'up': 5, 'down': 2, 'rho': 5}, n_bare
        assert n_in == {'up': 0, 'down': 2, 'rho': 1}, n_in
    finally:
        ctx.upload_labels, ctx.download_labels, ctx.upload_density = real_up, real_down, real_rho


@pytest.mark.parametrize('name,mode,iters', [('c64_cubic', 'changed', 2), ('c40x48x56_tric', 'all', -1), ('c48_cubic_vac', 'changed', 2),
                                              ('r48_sig5', 'changed', -1), ('r40_noise04', 'changed', 2), ('r32_quant8', 'all', -1),
                                              ('r64_noise04', 'changed', 2),      # 6 141 maxima: the bitmap numbering behind the gated call
                                              ('c12_cubic', 'changed', 2)])
def test_assign_refine_in_one_call_equals_the_two_calls(ctx, name, mode, iters):
    """xb_assign_refine (round 5: the refinement's first iteration queued behind the assignment, one host wait for both) against
    xb_assign followed by xb_refine: maxima, log, map and the state left behind (a second refine) must be the same -- on smooth
    grids (the deferred wait is taken), with a vacuum tolerance (not the fused combination: two calls inside), on 5-digit data
    with tie voxels, on noise with walkers for the exact slow path and on plateaus (the queued iteration is gated on the assignment's
    outcome and does nothing; the host part of the assignment takes over, then the refinement as an ordinary call) and on a grid
    too small for the brick pipeline."""
    from rough_common import load_rough
    if name.startswith('c'):
        g = load_golden(name)
        rho = case_density(g)
    else:
        g, rho = load_rough(name)
    tol = None if np.isnan(float(g['vacuum_tol'])) else float(g['vacuum_tol'])
    ctx.set_grid(rho.shape, g['dist_mat'], g['T_grad'])
    ctx.upload_density(rho)
    out = []
    for fused in (False, True):
        ctx.set_option(6, 1)
        ctx.vacuum_assign(tol, 1.0)
        if fused:
            n, log = ctx.assign_refine('neargrid', mode, iters)
        else:
            n = ctx.assign('neargrid')
            log = ctx.refine(mode, iters)
        lab = ctx.download_labels(np.int32)
        again = ctx.refine('all', 1)                    # what a later call finds: the flags, table and uniformity state left behind
        out.append((n, ctx.maxima(), log, lab, again, ctx.download_labels(np.int32)))
    a, b = out
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[2] == b[2]
    assert np.array_equal(a[3], b[3])
    assert a[4] == b[4] and np.array_equal(a[5], b[5])
