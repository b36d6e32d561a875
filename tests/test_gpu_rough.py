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
        assert compare(g, maps, tag, got, log) == exp[tag], tag   # deviation from the reference: as recorded
