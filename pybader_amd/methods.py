"""Kernel-level plugins with the reference's signature (pybader/methods.py:3-5: "all functions in
this module should have the same arguments"):

    method(density, volumes, idx, dist_mat, T_grad, i_c) -> (volumes, bader_max, edge_max)

`volumes` must be the whole grid (idx == 0): the GPU library does its own decomposition (slabs
across GPUs), so the reference's per-thread sub-blocks never reach this layer.  Labels come back
1-based like the reference's kernels (thread_handlers.volume_offset makes them 0-based)."""
import numpy as np

from . import _lib
from .utils import ensure_density

__contains__ = ['ongrid', 'neargrid']          # methods.py:12


def _run(method, density, volumes, idx, dist_mat, T_grad):
    if np.any(np.asarray(idx) != 0) or tuple(volumes.shape) != tuple(density.shape):
        raise ValueError("pybader_amd.methods kernels take the whole grid (idx == 0)")
    ctx = _lib.default_context()
    ctx.set_grid(density.shape, dist_mat, T_grad)
    ensure_density(ctx, density)
    ctx.upload_labels(volumes)
    ctx.assign(method)
    bader_max = ctx.maxima()
    out = ctx.download_labels(np.int32)
    out[out >= 0] += 1                         # 1-based local labels (methods.py:209 vol_num = bader_num)
    volumes[...] = out
    return volumes, bader_max, np.zeros((0, 3), dtype=np.int64)


def ongrid(density, volumes, idx, dist_mat, T_grad, i_c):
    """methods.ongrid (methods.py:15-219)."""
    return _run('ongrid', density, volumes, idx, dist_mat, T_grad)


def neargrid(density, volumes, idx, dist_mat, T_grad, i_c):
    """methods.neargrid (methods.py:222-611).  Returns every voxel's own-trajectory basin -- the
    order-independent map the reference reaches after refinement."""
    return _run('neargrid', density, volumes, idx, dist_mat, T_grad)
