#!/usr/bin/env python3
"""Diagnostic build of the library (-DXB_DEBUG_COUNT, written next to the product as libbader_hip_dbg.so -- never loaded
by the product) + one 512^3 neargrid assignment; prints the device-side diagnostic counters.  GPU box only.

    python tools/debug_counts.py [size]
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
subprocess.check_call([build.hipcc()] + build.FLAGS + ['-DXB_DEBUG_COUNT', '-o', dbg, build.SRC])
_lib.LIB_PATH = dbg
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lib = _lib.load()
raw = ctypes.CDLL(dbg)
shape = (size,) * 3
vl = np.divide(synth.CUBIC6, shape)
ctx = _lib.Context(0)
ctx.set_grid(shape, distance_matrix(vl), gradient_transform(vl))
ctx.synth_density(synth.CUBIC6, synth.ATOMS8, synth.BACKGROUND)
out = (ctypes.c_ulonglong * 16)()
for mirror in (1, 0):
    ctx.set_option(2, 0 if mirror else 1)
    raw.xb_debug_counts(out, 1)
    ctx.vacuum_assign(None, 1.0)
    ctx.assign('neargrid')
    raw.xb_debug_counts(out, 0)
    print('mirror', mirror, 'wave-iterations', out[0], 'open x', out[1], 'open y', out[2], 'open z', out[3], 'maybe max', out[4],
          'exact test runs', out[5], [int(v) for v in out[6:12]])
