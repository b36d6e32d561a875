"""Test-only transports for the slab scheduler (pybader_amd/slab.py).  The product moves planes through its own
C ABI (pybader_amd/comm.py: RCCL, or TCP staging); the CPU tests also drive the scheduler over torch.distributed's
gloo backend (world_size > 1 on CPU), and the one-GPU emulation copies planes between contexts of the same device
through torch views of the library's arrays."""
import numpy as np


class TorchComm:
    """gloo group + a host backend whose `tensors()` are torch views of numpy arrays"""

    def __init__(self, dist):
        self.dist = dist
        self.rank = dist.get_rank()
        self.size = dist.get_world_size()
        self.transport = 'gloo'

    def allgather(self, obj):
        out = [None] * self.size
        self.dist.all_gather_object(out, obj)
        return out

    def sum(self, *vals):
        got = self.allgather([int(v) for v in vals])
        return [sum(g[i] for g in got) for i in range(len(vals))]

    def max_float(self, x):
        return max(self.allgather(float(x)))

    def exchange_planes(self, backend, which, sends, recvs):
        """moves whole planes of the (nx, plane) array between ranks"""
        if not sends and not recvs:
            return
        tensor = backend.tensors()[which]
        ops = [self.dist.P2POp(self.dist.irecv, tensor[xa:xb], peer) for peer, xa, xb in recvs]
        ops += [self.dist.P2POp(self.dist.isend, tensor[xa:xb], peer) for peer, xa, xb in sends]
        for w in self.dist.batch_isend_irecv(ops):
            w.wait()

    def barrier(self):
        self.dist.barrier()


class _DevArray:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr, 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}


def device_views(ctx, device_index=0):
    """zero-copy torch views (labels (nx, plane) int32, known int8, brick masks int32) of a libbader_hip context"""
    import torch
    lib, h = ctx.lib, ctx.h
    nx, plane = ctx.shape[0], int(lib.xb_plane_elems(h))
    dev = f'cuda:{device_index}'
    lab = torch.as_tensor(_DevArray(lib.xb_labels_ptr(h), (nx, plane), '<i4'), device=dev)
    kn = torch.as_tensor(_DevArray(lib.xb_known_ptr(h), (nx, plane), '|i1'), device=dev)
    ptr, n, _, _ = ctx.brick_masks() if all(s % 8 == 0 for s in ctx.shape) else (0, 0, 0, 0)
    masks = torch.as_tensor(_DevArray(ptr, (n,), '<i4'), device=dev) if n else None
    # the bricks' single-maximum voxels live three brick arrays behind the masks (csrc: list + nbr / list + 4 nbr)
    maxvox = torch.as_tensor(_DevArray(ptr + 3 * n * 4, (n,), '<i4'), device=dev) if n else None
    return lab, kn, masks, maxvox


def block_view(ctx, which, device_index=0):
    """zero-copy uint8 torch view of exchange block `which` of the device-driven slab step (csrc/slab_step.h)"""
    import torch
    ptr, total, _, _ = ctx.slab_block(which)
    return torch.as_tensor(_DevArray(ptr, (total,), '|u1'), device=f'cuda:{device_index}')
