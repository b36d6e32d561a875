"""Kernel-level refinement plugins with the reference's signatures (pybader/refinement.py)."""
import numpy as np

from . import _lib
from .utils import ensure_density

__contains__ = ['neargrid']                    # refinement.py:13


def _ctx_for(density, dist_mat=None, T_grad=None):
    ctx = _lib.default_context()
    if dist_mat is not None:
        ctx.set_grid(density.shape, dist_mat, T_grad)
    elif ctx.shape != tuple(density.shape):
        ctx.set_grid(density.shape, np.zeros(27), np.zeros(9))
    ensure_density(ctx, density)
    return ctx


def edge_find(known, density, volumes):
    """refinement.edge_find (refinement.py:326-405): fills `known` in place, returns the edge count.
    `known` must be fresh (all zero), which is how the reference always calls it
    (thread_handlers.py:149-150, 202-203, 254-255)."""
    if np.any(known):
        raise ValueError("edge_find expects a fresh (all-zero) known array")
    ctx = _ctx_for(density)
    ctx.upload_labels(volumes)
    edges = ctx.edge_find()
    known[...] = ctx.download_known()
    return edges


def edge_check(known, density, volumes):
    """refinement.edge_check (refinement.py:409-508) -> (checked, edges); `known` in place."""
    ctx = _ctx_for(density)
    ctx.upload_labels(volumes)
    ctx.upload_known(known)
    checked, edges = ctx.edge_check()
    known[...] = ctx.download_known()
    return checked, edges


def neargrid(known, rknown, density, volumes, idx, dist_mat, T_grad, i_c):
    """refinement.neargrid (refinement.py:17-322) -> (known, changed); `known` and `volumes` in place.
    `rknown` must be the snapshot of `known` (thread_handlers.py:164): the device uses one array
    for both, which is equivalent because traces only test rknown == 2."""
    if np.any(np.asarray(idx) != 0) or tuple(known.shape) != tuple(density.shape):
        raise ValueError("pybader_amd.refinement kernels take the whole grid (idx == 0)")
    ctx = _ctx_for(density, dist_mat, T_grad)
    ctx.upload_labels(volumes)
    ctx.upload_known(known)
    changed, escaped = ctx.refine_trace()
    assert escaped == 0
    known[...] = ctx.download_known()
    ctx.download_labels(out=volumes)
    return known, changed
