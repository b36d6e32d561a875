"""The Python drop-in layer (pybader_amd.interface / thread_handlers / utils) on the CPU: the GPU context replaced by the oracle-backed
stand-in of tests/oracle_context.py, the results held against the reference's own (golden fixtures).  Host logic only -- argument
order, in-place semantics, dtypes, residency tokens, the fused bader_calc_refine route against the reference's two calls."""
import numpy as np
import pytest

import oracle
from conftest import case_density, load_golden
from oracle_context import OracleContext
from pybader_amd import _lib, synth, thread_handlers
from pybader_amd.interface import Bader


@pytest.fixture
def cpu_context(monkeypatch):
    ctx = OracleContext()
    monkeypatch.setattr(_lib, 'default_context', lambda device=None: ctx)
    monkeypatch.setattr(_lib, 'atom_assign', lambda bm, at, lat: oracle.atom_assign(
        np.ascontiguousarray(bm, np.float64), np.ascontiguousarray(at, np.float64), np.ascontiguousarray(lat, np.float64)))
    monkeypatch.setattr(thread_handlers, 'VERBOSE', False)
    return ctx


@pytest.mark.parametrize('fused', [True, False])
@pytest.mark.parametrize('name', ['c12_cubic', 'c40x48x56_tric'])
def test_bader_run_on_the_oracle_context_equals_the_reference(cpu_context, name, fused):
    g = load_golden(name)
    rho = case_density(g)
    b = Bader({'charge': rho}, g['lattice'], synth.atoms_cartesian(g['atoms'], g['lattice']))
    b.fused = fused
    b()
    assert b.bader_volumes.dtype == g['ng_changed_2'].dtype and np.array_equal(b.bader_volumes, g['ng_changed_2'])
    assert np.array_equal(b.atoms_volumes, g['ng_atoms_volumes']) and np.array_equal(b.bader_atoms, g['ng_bader_atoms'])
    np.testing.assert_allclose(b.bader_maxima, g['bader_maxima_cart'], rtol=1e-14)
    np.testing.assert_allclose(b.bader_charge, g['ng_bader_charge'], rtol=1e-9)
    np.testing.assert_allclose(b.atoms_charge, g['ng_atoms_charge'], rtol=1e-9)
    # one label download for the fused pair (the pre-refinement map never comes to the host), two for the reference's two calls;
    # the density goes up once per __call__ (utils.resident)
    calls = cpu_context.calls
    assert calls.count('upload_density') == 1
    assert cpu_context.pinned_density is None and cpu_context.resident_labels is None      # no token outlives the call


def test_bader_calc_refine_falls_back_where_refine_returns_silently(cpu_context):
    g = load_golden('c12_cubic')
    rho = case_density(g)
    for kw in (dict(refine_method='ongrid'), dict(refine_mode=('changed', 0))):   # thread_handlers.py:140-147
        vol = np.zeros(rho.shape, np.int8)
        args = dict(method='neargrid', refine_method='neargrid', refine_mode=('changed', 2))
        args.update(kw)
        bmax, out = thread_handlers.bader_calc_refine(args['method'], args['refine_method'], args['refine_mode'], rho, vol,
                                                      g['dist_mat'], g['T_grad'], 1)
        bmax2, out2 = thread_handlers.bader_calc('neargrid', rho, np.zeros(rho.shape, np.int8), g['dist_mat'], g['T_grad'], 1)
        assert np.array_equal(bmax, bmax2) and np.array_equal(out, out2)
    with pytest.raises(AttributeError):
        thread_handlers.bader_calc_refine('nosuch', 'neargrid', ('changed', 2), rho, np.zeros(rho.shape, np.int8), g['dist_mat'], g['T_grad'], 1)
