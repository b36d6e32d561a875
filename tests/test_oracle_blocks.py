"""The oracle's restatement of the reference's threads > 1 path (oracle/bader_oracle_blocks.c: factor_3d block split,
methods.neargrid with the block-extension branches, volume_offset / volume_merge / array_merge / edge_assign,
threaded refinement) against vectors captured from the reference itself with its blocks merged in submission order
(tests/golden/make_golden.py threads_blocks).  SURVEY.md section 8 rows a8 / a9."""
import json

import numpy as np
import pytest

import oracle
from conftest import load_golden
from pybader_amd import synth
from rough_common import load_rough


def density(cname):
    if cname.startswith('r'):
        g, rho = load_rough(cname)
    else:
        from conftest import case_density
        g = load_golden(cname)
        rho = case_density(g)
    return g, rho


G = load_golden('threads_blocks')
CASES = [(c, t) for c, ts in json.loads(str(G['cases_json'])).items() for t in ts]


def test_factor_3d_and_block_table():
    tab = load_golden('tables')
    for i, want in enumerate(tab['factor_3d']):
        assert oracle.factor_3d(i + 1) == tuple(int(v) for v in want)
    # np.array_split pieces in np.ndindex order; the shape-sorted zip of thread_handlers.py:28-29
    idx, ln = oracle.block_table((40, 48, 56), 6)
    assert ln.prod(axis=1).sum() == 40 * 48 * 56 and idx.shape == (6, 3)
    f = oracle.factor_3d(6)
    assert [len(set(idx[:, j].tolist())) for j in range(3)] == list(f)        # shapes ascending: split == factor_3d
    idx2, ln2 = oracle.block_table((56, 48, 40), 6)
    assert [len(set(idx2[:, j].tolist())) for j in range(3)] == [f[2], f[1], f[0]]


@pytest.mark.parametrize('cname,threads', CASES)
def test_block_path_equals_the_reference(cname, threads):
    g, rho = density(cname)
    assert synth.sha256(rho) == str(G[cname + '_rho_sha256'])
    tol = float(g['vacuum_tol'])
    vol0 = np.zeros(rho.shape, np.int32)
    vol0, _, _ = oracle.vacuum_assign(rho, vol0, tol, rho, 1.0)
    key = f'{cname}_t{threads}'
    bmax, main = oracle.bader_calc('neargrid', rho, vol0, g['dist_mat'], g['T_grad'], threads=threads, workers=2)
    assert np.array_equal(bmax, G[key + '_max'])
    assert main.dtype == G[key + '_main'].dtype and np.array_equal(main, G[key + '_main'])
    v = main.copy()
    oracle.refine('neargrid', ('changed', 2), rho, v, g['dist_mat'], g['T_grad'], threads=threads, workers=2)
    assert np.array_equal(v, G[key + '_changed_2'])
    # the blocks' result does not depend on how many workers run them
    bmax1, main1 = oracle.bader_calc('neargrid', rho, vol0, g['dist_mat'], g['T_grad'], threads=threads, workers=1)
    assert np.array_equal(main1, main) and np.array_equal(bmax1, bmax)
