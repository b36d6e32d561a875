import os
import sys

import numpy as np
import pytest

# The slab tests move halo planes through torch views of the library's device arrays.  PyTorch-ROCm
# bundles its own HIP runtime: it must be loaded BEFORE libbader_hip.so (then both share it); the
# other order leaves torch without a usable device.  The product itself never imports torch.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


@pytest.fixture(scope='session')
def golden():
    return load_golden


def case_density(g):
    """Rebuild the case's density from its stored parameters and check the stored sha256."""
    from pybader_amd import synth
    rho = synth.synth_density(tuple(int(s) for s in g['shape']), g['lattice'], g['atoms'], float(g['background']))
    assert synth.sha256(rho) == str(g['rho_sha256']), 'synthetic density generator drifted'
    return rho
