"""Per-rank cost of the slab path WITHOUT an N-GPU node: N logical ranks (one context each) share one GPU, driven by
N threads through the product's SlabRunner, and a global lock lets exactly one rank's library call run on the card
at a time.  The time a rank spends inside its calls is then what a dedicated GPU would spend on them (kernels +
the call's own host waits); the collectives between the calls are device-to-device copies here and are reported
separately (on a node they are RCCL transfers over xGMI).

Projection printed per N:  sum over phases of the slowest rank's time in that phase (the ranks run in lockstep),
next to the one-GPU step of the same grid.  Measurement aid for DESIGN.md section 5 -- not a benchmark of a node.

    python tools/slab_emulation.py --size 512 --ranks 1 2 4 8 --halo 16 --steps 5
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

os.environ.setdefault('HSA_ENABLE_INTERRUPT', '0')   # N host threads wait on one card: completion signals by polling, not by interrupt
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')    # one hardware queue per logical rank's stream (the default of 4 makes streams share queues)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from pybader_amd import _lib, slab, synth                      # noqa: E402
from pybader_amd.interface import distance_matrix, gradient_transform   # noqa: E402

GPU = threading.Lock()
sys.setswitchinterval(2e-5)   # the ranks are threads here: a rank coming back from a library call must not wait 5 ms for the GIL


class LockedBackend(slab.GpuBackend):
    """every library call of this rank runs alone on the card; its wall time is booked to the rank"""

    # calls that never touch the card (host-side state of the context): no lock, no device sync, nothing booked
    HOST_ONLY = {'slab_supported', 'slab_block', 'slab_walk_layout', 'slab_walk_send', 'maxima', 'host_waits', 'set_option', 'memory_stats', 'box_stats'}

    def __init__(self, ctx, book):
        super().__init__(ctx, 0)
        self.book = book
        self.waits = {}

    def __getattr__(self, name):
        f = getattr(self.ctx, name)
        if not callable(f) or name == 'sync' or name in self.HOST_ONLY:
            return f

        def call(*a, **k):
            with GPU:
                self.ctx.sync()
                t0 = time.perf_counter()
                w0 = self.ctx.host_waits()
                r = f(*a, **k)
                w1 = self.ctx.host_waits()
                self.ctx.sync()
                dt = time.perf_counter() - t0
                self.book[name] = self.book.get(name, 0.0) + dt
                if w1 > w0:      # waits of the host for the card inside the library call itself (not the emulation's own syncs)
                    self.waits[name] = self.waits.get(name, 0) + (w1 - w0)
                if dt > 2e-3 and os.environ.get('XB_EMUL_DEBUG'):
                    print(f'slow call {name}: {1e3 * dt:.1f} ms args {[getattr(x, "shape", x) for x in a]}', file=sys.stderr, flush=True)
            return r
        return call


def run(n, shape, g, halo, margin, steps, warmup, mode, iters):
    from test_gpu_slabs import Shared, ThreadComm
    sh = Shared(n)
    books = [dict() for _ in range(n)]
    waits = [dict() for _ in range(n)]
    steps_dev = [0] * n
    walls = [0.0] * n

    def work(rank):
        try:
            ctx = _lib.Context(0)
            if os.environ.get('XB_EMUL_DEBUG'):
                ctx.set_option(3, 16)
            be = LockedBackend(ctx, {})
            comm = ThreadComm(sh, rank)
            runner = slab.SlabRunner(be, comm, shape, g['dist_mat'], g['T_grad'], halo=halo)
            if n > 1:
                runner.enable_table_window(margin)
            with GPU:
                ctx.synth_density(synth.CUBIC6, synth.ATOMS8, synth.BACKGROUND)
            for it in range(warmup + steps):
                if it == warmup:
                    be.book = books[rank]
                    be.waits = waits[rank]
                    comm.barrier()
                    t0 = time.perf_counter()
                be.vacuum_assign(None, 1.0)
                runner.assign('neargrid')
                runner.refine(mode, iters)
            comm.barrier()
            walls[rank] = time.perf_counter() - t0
            steps_dev[rank] = runner.n_device_steps
            ctx.close()
        except Exception as e:  # noqa: BLE001
            sh.errors.append(repr(e))
            sh.barrier.abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if sh.errors:
        raise RuntimeError(sh.errors)
    names = sorted({k for b in books for k in b})
    per_phase = {k: [1e3 * b.get(k, 0.0) / steps for b in books] for k in names}
    slowest = {k: max(v) for k, v in per_phase.items()}
    return {'ranks': n, 'halo': halo, 'projected_ms': round(sum(slowest.values()), 3),
            'busiest_rank_ms': round(max(sum(b.values()) for b in books) * 1e3 / steps, 3),
            'phases_ms_slowest_rank': {k: round(v, 3) for k, v in slowest.items()},
            'host_waits_per_step': max(sum(w.values()) for w in waits) / steps,
            'host_waits_by_call': {k: max(w.get(k, 0) for w in waits) / steps for k in sorted({k for w in waits for k in w})},
            'device_driven_steps': min(steps_dev),
            'serialised_wall_ms': round(1e3 * max(walls) / steps, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--ranks', type=int, nargs='+', default=[1, 2, 4, 8])
    ap.add_argument('--halo', type=int, default=16)
    ap.add_argument('--margin', type=int, default=None)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--mode', default='changed')
    ap.add_argument('--iters', type=int, default=2)
    a = ap.parse_args()
    shape = (a.size,) * 3
    vl = np.divide(synth.CUBIC6, shape)
    g = {'dist_mat': distance_matrix(vl), 'T_grad': gradient_transform(vl)}
    for n in a.ranks:
        out = run(n, shape, g, a.halo, a.margin, a.steps, a.warmup, a.mode, a.iters)
        out['grid'] = a.size
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
