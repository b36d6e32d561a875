#!/usr/bin/env python3
"""Time probes of the persistent trace (k_ng_trace_g) from a diagnostic build (-DXB_DEBUG_COUNT -> pybader_amd/libbader_hip_dbg.so,
never loaded by the product): per wave the cycles spent walking, waiting at the workgroup's barrier and loading the brick's
records, and when each workgroup finished (the tail).  The probes write PLAIN stores into a slot per wave: summary atomics on one
line (round 5's first version) stretched the end of the kernel by 0.5 ms and everything measured there was the probe.  GPU box only; `--build` only compiles (works without a GPU).

    python tools/trace_probe.py [--build] [size] [--opt KEY=VALUE ...]
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pybader_amd import build, _lib, synth                      # noqa: E402
from pybader_amd.interface import distance_matrix, gradient_transform   # noqa: E402

dbg = os.path.join(ROOT, 'pybader_amd', 'libbader_hip_dbg.so')
args = [a for a in sys.argv[1:] if not a.startswith('--')]
if '--build' in sys.argv or not os.path.exists(dbg):
    subprocess.check_call([build.hipcc()] + build.FLAGS + ['-DXB_DEBUG_COUNT', '-o', dbg, build.SRC])
    if '--build' in sys.argv:
        sys.exit(0)
_lib.LIB_PATH = dbg
size = int(args[0]) if args else 512
lib = _lib.load()
raw = ctypes.CDLL(dbg)
shape = (size,) * 3
vl = np.divide(synth.CUBIC6, shape)
ctx = _lib.Context(0)
for a in sys.argv[1:]:
    if a.startswith('--opt='):
        k, v = a[6:].split('=')
        ctx.set_option(int(k), int(v))
ctx.set_grid(shape, distance_matrix(vl), gradient_transform(vl))
ctx.synth_density(synth.CUBIC6, synth.ATOMS8, synth.BACKGROUND)
out = (ctypes.c_ulonglong * 65536)()
for rep in range(3):
    raw.xb_debug_counts(out, 1)
    ctx.vacuum_assign(None, 1.0)
    ctx.assign('neargrid')
    raw.xb_debug_counts(out, 0)
o = np.array(out[:], np.float64)
WAVES = 4                                     # XB_TRACE_WAVES (k_trace.h): waves of a workgroup of the group trace
w = o[64:64 + 6 * 8192].reshape(8192 // WAVES, WAVES, 6)      # the kernel records its first 2048 workgroups, a slot per wave
w = w[(w[:, :, 1] > 0).all(axis=1)]          # workgroups of which every wave reported
bricks = int(o[64 + 6 * 8192:64 + 6 * 8192 + 2048].sum())
walk, total, wait, load = (w[:, :, k].sum() for k in range(4))
start = w[:, :, 4].min()
ends = (w[:, :, 5].max(axis=1) - start) / 100.0          # wall_clock64: 100 MHz -> microseconds; a workgroup ends with its last wave
n_waves = w.shape[0] * WAVES
print(f'waves {n_waves} bricks {bricks}: cycles per wave {total / n_waves:.0f}; walking {walk / total:.3f}, barrier wait {wait / total:.3f}, '
      f'record load {load / total:.3f}, rest {1 - (walk + wait + load) / total:.3f}')
print(f'workgroups {ends.size}: finish times (us after the first start) min {ends.min():.0f} p10 {np.percentile(ends, 10):.0f} '
      f'median {np.median(ends):.0f} p90 {np.percentile(ends, 90):.0f} max {ends.max():.0f}')
ctx.close()
