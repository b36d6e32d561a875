#!/usr/bin/env python3
"""Time probes of the persistent trace (k_ng_trace_g) from a diagnostic build (-DXB_DEBUG_COUNT -> pybader_amd/libbader_hip_dbg.so,
never loaded by the product): per wave the cycles spent walking, waiting at the workgroup's barrier and loading the brick's
records, and when each workgroup finished (the tail).  GPU box only; `--build` only compiles (works without a GPU).

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
out = (ctypes.c_ulonglong * 4096)()
for rep in range(3):
    raw.xb_debug_counts(out, 1)
    ctx.vacuum_assign(None, 1.0)
    ctx.assign('neargrid')
    raw.xb_debug_counts(out, 0)
o = [int(v) for v in out]
walk, total, wait, load, waves, bricks = o[16], o[17], o[18], o[19], o[21], o[22]
start = (1 << 62) - o[20]
ends = np.array([v for v in o[64:64 + 1024] if v], np.float64)
ends = (ends - start) / 100.0          # wall_clock64: 100 MHz -> microseconds
print(f'waves {waves} bricks {bricks}: cycles per wave {total / waves:.0f}; walking {walk / total:.3f}, barrier wait {wait / total:.3f}, '
      f'record load {load / total:.3f}, rest {1 - (walk + wait + load) / total:.3f}')
print(f'workgroups {ends.size}: finish times (us after the first start) min {ends.min():.0f} p10 {np.percentile(ends, 10):.0f} '
      f'median {np.median(ends):.0f} p90 {np.percentile(ends, 90):.0f} max {ends.max():.0f}')
hist = o[2200:2264]
print('brick durations (log2 bins of 10 ns):', {f'{(1 << k) / 100:.1f}us': v for k, v in enumerate(hist) if v})
nslow = min(o[23], 400)
raw = np.array(o[2300:2300 + 4 * nslow], np.uint64).reshape(-1, 4)
print(f'clock64 ticks per wall_clock64 tick, all bricks: {o[26] / max(o[25], 1):.2f} (x 100 MHz = the shader clock if clock64 counts shader cycles)')
steps = (raw[:, 0] >> np.uint64(32)).astype(np.float64)
slow = raw.astype(np.float64)
slow[:, 0] = (raw[:, 0] & np.uint64(0xffffffff)).astype(np.float64)
print(f'all bricks: mean longest-eighth wave-steps {o[24] / max(bricks, 1):.1f}, mean duration {o[25] / max(bricks, 1) / 100:.1f} us -> {o[25] / max(o[24], 1) * 10:.0f} ns per wave-step of the longest eighth')
if nslow:
    n_walk = bricks
    order = np.argsort(-slow[:, 1])
    print(f'{o[23]} bricks took > 100 us; the slowest (list position / list length, duration us, pulled at us):')
    for k in order[:12]:
        print(f'   {int(slow[k, 0])} / {n_walk}  {slow[k, 1] / 100:.0f}  {(slow[k, 2] - start) / 100:.0f}  steps {int(steps[k])}  -> {slow[k, 1] * 10 / max(steps[k], 1):.0f} ns per step')
    t0 = (slow[:, 2] - start) / 100
    for a in range(0, 1800, 200):
        m = (t0 >= a) & (t0 < a + 200)
        if m.any():
            print(f'   slow bricks pulled in [{a}, {a + 200}) us: {int(m.sum())}, mean duration {slow[m, 1].mean() / 100:.0f} us, max {slow[m, 1].max() / 100:.0f}, mean steps {steps[m].mean():.0f}, clock64/wall {(slow[m, 3] / slow[m, 1]).mean():.2f}, ns per step {(slow[m, 1] * 10 / np.maximum(steps[m], 1)).mean():.0f}')
ctx.close()
