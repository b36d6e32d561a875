#!/usr/bin/env python3
"""How long are the walkers of the group trace, and what would a different dealing of a brick's 512 walkers to its eight
wave-loads save?  Diagnostic build (-DXB_DEBUG_COUNT -> pybader_amd/libbader_hip_dbg.so, never loaded by the
product): every lean walker notes its step count in `known`.  A wave-load lasts as long as its longest walker, so the cost of a
dealing is the sum over its eight loads of the largest count.  GPU box only; `--build` only compiles.

    python tools/walk_lengths.py [--build] [size] [--lattice cubic|triclinic] [--atoms 8|216]
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
lat = synth.TRICLINIC if '--lattice=triclinic' in sys.argv else synth.CUBIC6
lib = _lib.load()
raw = ctypes.CDLL(dbg)
shape = (size,) * 3
vl = np.divide(lat, shape)
ctx = _lib.Context(0)
ctx.set_grid(shape, distance_matrix(vl), gradient_transform(vl))
atoms = synth.ATOMS8
if '--atoms=216' in sys.argv:
    atoms = synth.atoms_grid(6) if hasattr(synth, 'atoms_grid') else synth.ATOMS8
ctx.synth_density(lat, atoms, synth.BACKGROUND)
ctx.vacuum_assign(None, 1.0)
raw.xb_debug_steps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
assert raw.xb_debug_steps(ctx.h, 1, None) == 0
ctx.assign('neargrid')
buf = np.zeros(shape, np.int8)
assert raw.xb_debug_steps(ctx.h, 1, buf.ctypes.data_as(ctypes.c_void_p)) == 0      # fetch, then clear for the retraces
steps = buf.astype(np.int32)
log = ctx.refine('changed', 1)
rbuf = np.zeros(shape, np.int8)
assert raw.xb_debug_steps(ctx.h, 0, rbuf.ctypes.data_as(ctypes.c_void_p)) == 0
r = rbuf.reshape(-1).astype(np.int32)
edges = np.flatnonzero(r)            # C order = the order of no list in particular; the edge list is tile ordered
rl = r[edges] - 1
print(f'retraces: {edges.size} edge voxels (log {log}), loop iterations: mean {rl.mean():.2f}, median {np.median(rl):.0f}, p90 {np.percentile(rl, 90):.0f}, '
      f'p99 {np.percentile(rl, 99):.0f}, max {rl.max()}; histogram 0..24+: {np.bincount(np.minimum(rl, 24)).tolist()}')
for name, order in (('C order, 64 consecutive edge voxels a wave', rl), ('sorted (bound)', -np.sort(-rl))):
    pad = (-len(order)) % 64
    w = np.concatenate([order, np.zeros(pad, order.dtype)]).reshape(-1, 64)
    print(f'  retraces, {name}: wave-iterations {w.max(1).sum()}, lanes {rl.sum() / max(1, w.max(1).sum()):.1f} of 64')
# would the assignment walker's step count of the same voxel predict the retrace's length?  (edge lists binned by that hint)
hint = steps.reshape(-1)[edges]
print(f'hint (assignment steps of the edge voxel): {np.mean(hint == 0) * 100:.1f}% without a walker; correlation with the retrace length {np.corrcoef(hint, rl)[0, 1]:.3f}')


def lanes_of(groups):
    it = 0
    for g_ in groups:
        if len(g_) == 0:
            continue
        pad = (-len(g_)) % 64
        w = np.concatenate([g_, np.zeros(pad, g_.dtype)]).reshape(-1, 64)
        it += w.max(1).sum()
    return rl.sum() / max(1, it), it


for bins in ([1, 4, 8, 16], [1, 3, 6, 10, 16, 24], [1, 2, 4, 6, 8, 12, 16, 24, 32]):
    cls = np.digitize(hint, bins)
    ln, it = lanes_of([rl[cls == k] for k in range(len(bins) + 1)])
    print(f'  retraces binned by the hint at {bins}: wave-iterations {it}, lanes {ln:.1f} of 64')
order = np.argsort(hint, kind='stable')
ln, it = lanes_of([rl[order]])
print(f'  retraces sorted by the hint: wave-iterations {it}, lanes {ln:.1f} of 64')
# tile order (4 x 8 x 64 tiles as the edge sweep lists them) instead of C order
n3 = size
ex, ey, ez = np.unravel_index(edges, shape)
tile = ((ex // 4) * (n3 // 8) + ey // 8) * (n3 // 64) + ez // 64
o2 = np.argsort(tile, kind='stable')
ln, it = lanes_of([rl[o2]])
print(f'  retraces in tile order: wave-iterations {it}, lanes {ln:.1f} of 64')
for bins in ([1, 4, 8, 16], [1, 2, 4, 6, 8, 12, 16, 24, 32]):
    cls = np.digitize(hint[o2], bins)
    ln, it = lanes_of([rl[o2][cls == k] for k in range(len(bins) + 1)])
    print(f'  tile order, binned by the hint at {bins}: wave-iterations {it}, lanes {ln:.1f} of 64')
nb = size // 8
B = steps.reshape(nb, 8, nb, 8, nb, 8).transpose(0, 2, 4, 1, 3, 5).reshape(-1, 8, 8, 8)
B = B[B.reshape(len(B), -1).max(1) > 0]                     # the walk-list bricks
n = len(B)
flat = B.reshape(n, 512)
print(f'{size}^3: {n} walk-list bricks, {flat.size} walkers, mean {flat.mean():.2f} steps, median {np.median(flat):.0f}, '
      f'p90 {np.percentile(flat, 90):.0f}, p99 {np.percentile(flat, 99):.0f}, max {flat.max()}')
print('histogram of steps:', np.bincount(np.minimum(flat.ravel(), 40))[:41].tolist())
total = flat.sum()


def report(name, cost):
    print(f'  {name:44s} wave-steps/brick {cost / n:7.2f}   lanes {total / (64.0 * cost) * 64:5.1f} of 64')


cubes = B.reshape(n, 2, 4, 2, 4, 2, 4).transpose(0, 1, 3, 5, 2, 4, 6).reshape(n, 8, 64)
report('4x4x4 eighths (until round 6)', cubes.max(2).sum())
slabs = B.reshape(n, 2, 4, 4, 2, 8).transpose(0, 1, 3, 2, 4, 5).reshape(n, 8, 64)   # k_trace.h brick_sub_voxel
report('4x2x8 eighths (what ships)', slabs.max(2).sum())
srt = -np.sort(-flat, axis=1)
report('sorted by the true length (bound)', srt[:, ::64].sum())
px = B.reshape(n, 8, 64).max(2).sum(1)
py = B.transpose(0, 2, 1, 3).reshape(n, 8, 64).max(2).sum(1)
pz = B.transpose(0, 3, 1, 2).reshape(n, 8, 64).max(2).sum(1)
report('x-planes', px.sum()); report('y-planes', py.sum()); report('z-planes', pz.sum())
report('best plane direction per brick (bound)', np.minimum(np.minimum(px, py), pz).sum())
# direction-sorted dealing: a brick's voxels ordered by s.x with s the sign pattern of the brick's mean length gradient
gx = (B[:, 4:].mean((1, 2, 3)) - B[:, :4].mean((1, 2, 3)))
gy = (B[:, :, 4:].mean((1, 2, 3)) - B[:, :, :4].mean((1, 2, 3)))
gz = (B[:, :, :, 4:].mean((1, 2, 3)) - B[:, :, :, :4].mean((1, 2, 3)))
X, Y, Z = np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing='ij')
cost = 0
for thr in (0.0, 0.5, 1.0):
    cost = 0
    sx, sy, sz = [np.where(np.abs(g) > thr, np.sign(g), 0) for g in (gx, gy, gz)]
    key = sx[:, None, None, None] * X + sy[:, None, None, None] * Y + sz[:, None, None, None] * Z
    order = np.argsort(-key.reshape(n, 512), axis=1, kind='stable')
    dealt = np.take_along_axis(flat, order, axis=1).reshape(n, 8, 64)
    report(f'sorted by s.x, s = sign of the length trend (> {thr})', dealt.max(2).sum())
for cap in (4, 6, 8, 12, 16):
    first = np.minimum(cubes.max(2), cap).sum()
    rest = np.maximum(flat - cap, 0)
    rs = -np.sort(-rest, axis=1)
    second = rs[:, ::64].sum()
    parked = (rest > 0).sum() / flat.size
    report(f'two phases, cap {cap} ({100 * parked:.0f}% parked)', first + second)
# steps of a wave-load: the distribution of the eighths' maxima
m = cubes.max(2).ravel()
print(f'eighth maxima: mean {m.mean():.1f}, p50 {np.median(m):.0f}, p90 {np.percentile(m, 90):.0f}, p99 {np.percentile(m, 99):.0f}, max {m.max()}')
np.save(os.path.join(ROOT, 'gpurun_out', f'walk_lengths_{size}.npy'), flat.astype(np.int8)[:4096])

# The tail of the persistent trace: 2 048 workgroups take the bricks of their XCD's eighth of the Morton-ordered list as they come;
# a brick costs about (the maxima of its eight eighths, four waves at a time) wave-steps.  How long is the tail, and would a
# "heavy bricks first" order shorten it?  The predictor a list kernel could afford: how many of a brick's 26 neighbours are walk
# bricks themselves (a walk ends on entering a certified brick).
import heapq
Bg = steps.reshape(nb, 8, nb, 8, nb, 8).transpose(0, 2, 4, 1, 3, 5)
Wm = Bg.reshape(nb, nb, nb, 512).max(3) > 0
e8 = Bg.reshape(nb, nb, nb, 2, 4, 4, 2, 8).transpose(0, 1, 2, 3, 5, 4, 6, 7).reshape(nb, nb, nb, 8, 64).max(4).astype(np.int64)
cost = np.sort(e8, axis=3)[..., ::-1]
cost = cost[..., 0] + cost[..., 4]          # four waves: the slowest of the first four eighths + the slowest of the rest (roughly)
nbrs = sum(np.roll(Wm, (a, b, c), (0, 1, 2)) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)).astype(np.int64) - Wm
cw, nw = cost[Wm], nbrs[Wm]
print(f'brick cost (wave-steps): mean {cw.mean():.1f}, p90 {np.percentile(cw, 90):.0f}, p99 {np.percentile(cw, 99):.0f}, max {cw.max()}; '
      f'correlation with the number of walk-brick neighbours {np.corrcoef(cw, nw)[0, 1]:.2f}')


def morton(ix, iy, iz):
    m = np.zeros_like(ix)
    for b in range(8):
        m |= ((ix >> b) & 1) << (3 * b + 2) | ((iy >> b) & 1) << (3 * b + 1) | ((iz >> b) & 1) << (3 * b)
    return m


ix, iy, iz = np.nonzero(Wm)
order = np.argsort(morton(ix, iy, iz), kind='stable')
c_m, n_m = cost[ix, iy, iz][order], nbrs[ix, iy, iz][order]


def makespan(c, workers):
    h = [0] * workers
    for x in c:
        heapq.heapreplace(h, h[0] + int(x))
    return max(h), sum(h) / workers


for name, seq in (('Morton order (what ships)', c_m),
                  ('heavy first by the neighbour count (>= 20), Morton within', np.concatenate([c_m[n_m >= 20], c_m[n_m < 20]])),
                  ('heavy first by the true cost (bound)', -np.sort(-c_m))):
    per = (len(seq) + 7) // 8
    spans = [makespan(seq[k * per:(k + 1) * per] if 'bound' not in name and 'heavy first by the n' not in name else seq[k::8], 256) for k in range(8)]
    print(f'  {name:62s} makespan {max(s[0] for s in spans)} wave-steps, mean load {np.mean([s[1] for s in spans]):.0f}')
