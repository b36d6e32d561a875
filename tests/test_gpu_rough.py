"""-m gpu: the HIP path on the NON-SMOOTH golden cases (reference-generated, tests/golden/make_golden.py ROUGH).

What is bit-exact there and what is not:
* every kernel that replaces an order-independent reference function is bit-exact on these inputs too: ongrid, edge_find,
  the refinement kernels (all modes) started from the reference's own sequential main map;
* xb_assign(neargrid) returns the own-trajectory map under methods.neargrid's stepping rule, basins numbered by
  their smallest voxel -- equal to the oracle's restatement of exactly that, bit for bit;
* against the reference's SEQUENTIAL result (main pass + refine) that pipeline deviates by the counts recorded in
  tests/golden/rough_expected.json (label permutations of the basin numbering, a handful of voxels on ties,
  vacuum voxels the reference's main pass relabels) -- asserted here so that the documentation cannot drift."""
import numpy as np
import pytest

from pybader_amd import _lib
from rough_common import MODES, ROUGH_CASES, compare, expected, load_rough, pipeline_maps, refined, vac_tol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def setup(ctx, name):
    g, rho = load_rough(name)
    ctx.set_grid(rho.shape, g['dist_mat'], g['T_grad'])
    ctx.upload_density(rho)
    return g, rho


@pytest.mark.parametrize('name', ROUGH_CASES)
def test_order_independent_kernels_equal_the_reference(ctx, name):
    g, rho = setup(ctx, name)
    ctx.vacuum_assign(vac_tol(g), float(g['voxel_volume']))
    assert np.array_equal(ctx.download_labels(np.int8), g['ng_init'])
    # ongrid: memoryless, order independent -> the reference's map and maxima order
    ctx.assign('ongrid')
    assert np.array_equal(ctx.maxima(), g['og_bader_max'])
    assert np.array_equal(ctx.download_labels(g['og_main'].dtype), g['og_main'])
    log = ctx.refine('changed', 2)
    assert np.array_equal(np.array(log, np.int64).reshape(-1, 2), g['og_ngrefine_changed_2_log'])
    assert np.array_equal(ctx.download_labels(g['og_main'].dtype), g['og_ngrefine_changed_2'])
    # refinement from the reference's sequential main map: logs and maps of every mode
    for tag, mode in MODES.items():
        if tag not in g.files:
            continue
        ctx.upload_labels(g['ng_main'])
        log = ctx.refine(*mode)
        assert np.array_equal(np.array(log, np.int64).reshape(-1, 2), g[tag + '_log']), tag
        assert np.array_equal(ctx.download_labels(g[tag].dtype), g[tag]), tag


@pytest.mark.parametrize('name', ROUGH_CASES)
def test_neargrid_pipeline_vs_oracle_and_recorded_deviation(ctx, name):
    g, rho = setup(ctx, name)
    exp = expected()[name]
    maps = pipeline_maps(g, rho)                       # oracle restatement of what xb_assign must return
    for tag, mode in MODES.items():
        if tag not in g.files:
            continue
        ctx.vacuum_assign(vac_tol(g), float(g['voxel_volume']))
        n = ctx.assign('neargrid')
        assert n == exp['n_maxima']
        assert np.array_equal(np.ravel_multi_index(tuple(ctx.maxima().T), rho.shape), maps['maxima'])
        assert np.array_equal(ctx.download_labels(np.int64), maps['assign'])
        log = ctx.refine(*mode)
        got = ctx.download_labels(np.int32)
        want, olog = refined(g, rho, maps, tag)        # the oracle's refinement of the same map
        assert [list(x) for x in log] == [list(x) for x in olog], tag
        assert np.array_equal(got, want), tag
        assert compare(g, maps, tag, got, log, rho) == exp[tag], tag   # deviation from the reference: as recorded


@pytest.mark.parametrize('name', ROUGH_CASES)
def test_atom_map_and_per_atom_charges_through_the_drop_in(name):
    """north_star's own parity gate on rough inputs: the voxel -> atom map and the per-atom charges /
    volumes of Bader.__call__ (interface.py:399-416) through the Python drop-in, against what the REFERENCE produced for the
    same density and refine mode (tests/golden/make_golden.py run_rough_case).  The atom map does not depend on the basin
    numbering: wherever the recorded partition equals the reference's (basin_diff == 0) it must be the reference's bit for
    bit, dtype included, and the sums agree to 1e-6; elsewhere the recorded voxel counts."""
    from pybader_amd import thread_handlers
    from pybader_amd.interface import Bader
    from rough_common import atoms_deviation
    thread_handlers.VERBOSE = False
    g, rho = load_rough(name)
    exp = expected()[name]
    for tag, mode in MODES.items():
        if tag not in g.files:
            continue
        # (distance_matrix / T_grad are host numpy in the reference and move in the last ulp with the numpy version on a
        # triclinic lattice -- DESIGN.md section 2: the kernels take them as data, so the captured matrices are fed)
        class Captured(Bader):
            distance_matrix = property(lambda self: g['dist_mat'])
            T_grad = property(lambda self: g['T_grad'])
        b = Captured({'charge': rho}, g['lattice'], g['atoms_cart'], vacuum_tol=vac_tol(g), refine_mode=mode)
        np.testing.assert_allclose(Bader.distance_matrix.fget(b), g['dist_mat'], rtol=4e-16)
        b()
        dev = atoms_deviation(g, tag, b.atoms_volumes, rho)
        assert dev == {k: exp[tag][k] for k in dev}, (tag, dev)
        assert b.atoms_volumes.dtype == g[tag + '_atoms_volumes'].dtype and b.bader_atoms.dtype == g[tag + '_bader_atoms'].dtype
        if exp[tag]['basin_diff'] == 0:
            assert np.array_equal(b.atoms_volumes, g[tag + '_atoms_volumes']), tag
            np.testing.assert_allclose(b.atoms_charge, g[tag + '_atoms_charge'], rtol=1e-6, atol=0)
            np.testing.assert_allclose(b.atoms_volume, g[tag + '_atoms_volume'], rtol=1e-6, atol=0)
            # the per-basin sums are the reference's too, permuted by the numbering: compare as multisets
            np.testing.assert_allclose(np.sort(b.bader_charge), np.sort(g[tag + '_bader_charge']), rtol=1e-6, atol=1e-12)


def test_slow_path_does_not_read_labels_nobody_wrote():
    """Found by a random soak against the oracle: without vacuum the assignment drops the deferred `labels := 0` (it writes
    every label and reads none) -- but the exact slow kernel applied the vacuum rule (`labels[maximum] == -1`,
    methods.py:449-452) unconditionally and read the label of a maximum nobody had written yet: whatever the memory held,
    e.g. the -1 of a closed context's vacuum.  Here the label array is poisoned with -1 on purpose; a noisy density sends
    a thousand trajectories to the slow kernel; the map must still be the oracle's own-trajectory map."""
    import oracle
    from pybader_amd import synth
    from pybader_amd.interface import distance_matrix, gradient_transform
    from rough_common import own_map, rank_labels
    shape = (96, 48, 16)
    vl = np.divide(synth.TRICLINIC, shape)
    dm, tg = distance_matrix(vl), gradient_transform(vl)
    ctx = _lib.Context(0)
    ctx.set_grid(shape, dm, tg)
    ctx.synth_density(synth.TRICLINIC, synth.ATOMS8, synth.BACKGROUND)
    rho = np.ascontiguousarray(ctx.download_density() + 1e-6 * np.random.default_rng(72).random(shape))
    ctx.upload_density(rho)
    ctx.upload_labels(np.full(shape, -1, np.int32))
    ctx.vacuum_assign(None, 1.0)
    n = ctx.assign('neargrid')
    got = ctx.download_labels(np.int64)
    assert ctx.slow_path_stats()[0] > 100          # the path under test ran
    ctx.close()
    lab, maxima = rank_labels(own_map(rho, np.zeros(shape, np.int32), dm, tg, main_ties=True))
    assert n == len(maxima) and np.array_equal(got, lab)


def test_every_tier_of_the_slow_path_gives_the_same_map():
    """Round 6: the exact slow kernel runs in tiers of 64 / 2048 / 32768 path voxels and -- instead of failing the call as rounds 1-5
    did -- a last one of 2^20.  With debug switch 64 the tiers hold 3 / 5 / 8 voxels, so that the walkers of a noisy density pass
    through every one of them, the last included; the assignment and the refinement must give what the default tiers give."""
    from pybader_amd import synth
    from pybader_amd.interface import distance_matrix, gradient_transform
    shape = (96, 48, 16)
    vl = np.divide(synth.TRICLINIC, shape)
    dm, tg = distance_matrix(vl), gradient_transform(vl)
    res = []
    for dbg in (0, 64):
        ctx = _lib.Context(0)
        ctx.set_option(3, dbg)
        ctx.set_grid(shape, dm, tg)
        ctx.synth_density(synth.TRICLINIC, synth.ATOMS8, synth.BACKGROUND)
        rho = np.ascontiguousarray(ctx.download_density() + 1e-6 * np.random.default_rng(72).random(shape))
        ctx.upload_density(rho)
        ctx.vacuum_assign(None, 1.0)
        n = ctx.assign('neargrid')
        pre = ctx.download_labels(np.int32)
        log = ctx.refine('all', 2)
        res.append((n, ctx.maxima(), pre, log, ctx.download_labels(np.int32), ctx.slow_path_stats()))
        ctx.close()
    a, b = res
    assert a[5][0] > 100 and a[5] == b[5]
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3] and np.array_equal(a[4], b[4])
