"""A short seeded run of tests/soak_vs_oracle.py: random grids, lattices, noise, plateaus, vacuum and refinement modes through
the one-GPU pipeline against the CPU oracle -- assignment map, basin order, refinement log and refined map."""
import pytest

from soak_vs_oracle import run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('method,odd,seed', [('neargrid', False, 101), ('neargrid', True, 102), ('ongrid', False, 103), ('ongrid', True, 104)])
def test_random_cases_equal_the_oracle(method, odd, seed):
    failures = run(25, seed, method, odd)
    assert not failures, '\n'.join(failures)


def test_random_grids_the_brick_lattice_does_not_divide():
    """round 4: grids of 41..149 voxels per axis, any remainder modulo 8 (also 1: the bricks one voxel wide that are never
    certified) -- the brick pipeline with cut bricks against the oracle"""
    failures = run(8, 105, 'neargrid', oddbig=True)
    assert not failures, '\n'.join(failures)
