#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json: Mvoxels/s of neargrid assign + edge refinement
on a synthetic 512^3 grid (BASELINE config 3), data resident in HBM, on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 512] [--method neargrid]
                    [--refine changed:2] [--cpu-size 320] [--no-cpu]

A "step" is one full pass of the hot path over the grid: volumes_init (label reset) -> bader_calc
-> refine, exactly the call sequence of Bader.__call__ (interface.py:406-409).  N > 1: one rank per
GPU (no PyTorch in the ranks: RCCL through the library's C ABI), either launched by torch.distributed.run
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or -- `python bench.py --gpus N` as typed -- by this script itself, which then spawns N fresh
rank processes BEFORE anything touches the GPU, relays rank 0's JSON line and exits non-zero if any rank
failed.  The grid is cut into axis-0 slabs (strong scaling: the 512^3 grid is fixed, north_star: ">= 6x at
8 GPUs").  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
BYTES_ASSIGN = 12              # read rho f64 once + write int32 label once
BYTES_PATH = 25                # assign 12 + first refine sweep 13 (rho 8 + label 4 + known 1)


def cpu_baseline(size, method, mode, iters, lattice, atoms, background, full_size):
    """The CPU oracle (a C restatement of the reference, bit-identical to it on the golden vectors) timed on this host.
    Two legs: (i) one core, the reference's threads=1 path, on a bounded `size`^3 sample of the workload; (ii) every
    core of the box's CPU share, the reference's threads=N path -- factor_3d blocks, methods.neargrid per block,
    merge, threaded refinement (oracle/bader_oracle_blocks.c, OpenMP over the blocks) -- on the configuration's own
    `full_size`^3 grid.  Reported, never the target."""
    import oracle
    from pybader_amd.interface import distance_matrix, gradient_transform

    def run(n, threads):
        shape = (n,) * 3
        rho = oracle.synth_density(shape, lattice, atoms, background)
        vl = np.divide(lattice, shape)
        dm, tg = distance_matrix(vl), gradient_transform(vl)
        vol = np.zeros(shape, np.int32)
        t0 = time.perf_counter()
        vol, _, _ = oracle.vacuum_assign(rho, vol, float('nan'), rho, 1.0)
        bmax, main = oracle.bader_calc(method, rho, vol, dm, tg, threads, workers=threads)
        t1 = time.perf_counter()
        oracle.refine('neargrid', (mode, iters), rho, main, dm, tg, threads, workers=threads)
        t2 = time.perf_counter()
        return float(n) ** 3, t1 - t0, t2 - t1, (rho, dm, tg, main, bmax)

    legs = {}
    for n in (64, 256):        # configs 1 and 2 on the host, one core and the box's share of cores
        # (the oracle restates the reference's threaded block path for neargrid, the headline method, only)
        for threads in ((1, min(16, len(os.sched_getaffinity(0)))) if method == 'neargrid' else (1,)):
            nv, a, r, _ = run(n, threads)
            legs[f'{n}^3_{threads}core'] = {'value': nv / (a + r) / 1e6, 'assign_mvox_s': nv / a / 1e6, 'assign_s': a, 'refine_s': r}
    nvox, ta, tr, sample = run(size, 1)
    out = {'value': nvox / (ta + tr) / 1e6, 'unit': 'Mvoxels/s', 'cores': 1, 'kind': 'port', 'legs': legs,
           'sample': f'{size}^3 grid, same 8 atoms/cell, {method} assign + refine ({mode},{iters}), '
                     f'assign {ta:.2f}s + refine {tr:.2f}s, single thread C port of the numba path (threads=1)',
           'assign_mvox_s': nvox / ta / 1e6}
    cores = min(16, len(os.sched_getaffinity(0)))          # the box's CPU share for one GPU
    if method == 'neargrid' and cores > 1:
        nvox, ta, tr, _ = run(full_size, cores)
        out['single_core'] = {k: out[k] for k in ('value', 'cores', 'sample', 'assign_mvox_s')}
        out.update({'value': nvox / (ta + tr) / 1e6, 'cores': cores, 'assign_mvox_s': nvox / ta / 1e6,
                    'sample': f'{full_size}^3 grid (the configuration itself), {method} assign + refine ({mode},{iters}), '
                              f'assign {ta:.2f}s + refine {tr:.2f}s, C port of the reference\'s threads={cores} path '
                              f'(factor_3d blocks {oracle.factor_3d(cores)}, OpenMP over the blocks; edge_find / edge_check '
                              'sequential as in the reference)'})
    return out, sample


def source_hash():
    """sha256 over the kernel sources (pybader_amd/csrc + include): a PMC summary names the build it was taken from"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, 'pybader_amd', 'csrc', '*')) + glob.glob(os.path.join(ROOT, 'include', '*.h'))):
        with open(path, 'rb') as f:
            h.update(os.path.basename(path).encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def traffic_from_profile(kernel, method):
    """(bytes, 'file sha256[:16]') from the newest committed PMC summary of this command: FETCH_SIZE + WRITE_SIZE of `kernel`
    per launch, or of every kernel of ONE step (kernel None; the summary is of a one-step run) -- or (None, why) when there is
    no summary or it was taken from other kernel sources than the ones running now (its `# sources` line; ADVICE r2: the
    traffic must not silently describe an older build).  Lines look like `FETCH_SIZE  k_name<...>   launches=  1 KB=  1315370`."""
    import glob
    import hashlib
    import re
    def order(f):   # round number, then the round's final profile after its intermediate ones
        name = os.path.basename(f)
        m = re.match(r'r(\d+)', name)
        return (int(m.group(1)) if m else -1, '_final_' in name, name)
    # (the steady-state summary of a round -- 1 warm-up + 3 steps, first step dropped -- goes before its one-step, cold one)
    def order2(f):
        return order(f)[:2] + ('_steady_' in os.path.basename(f), os.path.basename(f))
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r*_pmc_fetch_write*_512_{method}.txt')), key=order2)
    for path in reversed(files):
        kb, launches = 0.0, 0
        with open(path) as f:
            text = f.read()
        src = re.search(r'^# sources (\w+)', text, re.M)
        name = f'{os.path.relpath(path, ROOT)} sha256:{hashlib.sha256(text.encode()).hexdigest()[:16]}'
        if not src or src.group(1) != source_hash():
            return None, f'{name} was taken from other kernel sources ({src.group(1) if src else "unrecorded"} != {source_hash()}): dropped'
        for line in text.splitlines():
            m = re.match(r'(FETCH_SIZE|WRITE_SIZE)\s+(\S+).*launches=\s*(\d+)\s+KB=\s*(\d+)', line)
            if not m:
                continue
            k = m.group(2).split('<')[0]
            if kernel is None:
                if k not in ('k_synth_density', 'k_fill', 'k_charge_sum_lds'):   # the generator, the one-off fill and the map check after the timed region are not part of a step
                    kb += float(m.group(4))
                    launches = 1
            elif k == kernel or k.startswith(kernel + '_'):   # (k_ng_trace: the group form k_ng_trace_g)
                kb += float(m.group(4))
                launches = max(launches, int(m.group(3)))
        if launches:
            return kb * 1024.0 / launches, name
    return None, 'no PMC summary committed for this command'


def user_legs(size, steps, warmup, mode, iters, headline_ns_per_voxel, stage_names):
    """Round 5: what users run, timed like the headline -- the same step (volumes_init + bader_calc + refine) on
    densities the 8-atom cubic cell flatters: a triclinic cell (the generic Grid instantiations: no mirror prefilter, full
    T_grad), a 216-atom cell (bricks with several maxima are never certified) and a smooth density with noise in its vacuum
    (thousands of spurious maxima).  Each leg reports its per-voxel time against the headline's and the certified fraction."""
    from pybader_amd import _lib, synth
    from pybader_amd.interface import distance_matrix, gradient_transform
    shape = (size,) * 3
    nv = float(size) ** 3
    legs = {}
    atoms216 = synth.atoms_jittered_grid(6, 5)
    cases = [('triclinic_8_atoms', synth.TRICLINIC, synth.ATOMS8, None,
              'triclinic cell [[6,0,0],[1.5,5.5,0],[0.7,1.1,6.2]], the headline\'s 8 atoms'),
             ('cubic_216_atoms', synth.CUBIC6, atoms216, None, '216 atoms (6 x 6 x 6, jittered) in the cubic cell'),
             ('noisy_vacuum', synth.CUBIC6, synth.ATOMS8, 2e-3,
              'the headline\'s density + uniform noise of amplitude 2e-3 wherever it is below 0.2 (tests/test_gpu_fullsize.py::test_noisy_vacuum_keeps_the_atoms_regions at full size)')]
    cases = [c + (None,) for c in cases]
    # ... and that density the way its users run it: with a vacuum tolerance that takes the noise out
    cases.append(('noisy_vacuum_with_tol', synth.CUBIC6, synth.ATOMS8, 2e-3,
                  'the noisy-vacuum density with vacuum_tol = 0.21 (every voxel the noise touches is vacuum)', 0.21))
    for name, lat, atoms, noise, what, vac_tol in cases:
        vl = np.divide(lat, shape)
        c = _lib.Context(0)
        c.set_grid(shape, distance_matrix(vl), gradient_transform(vl))
        c.synth_density(lat, atoms, synth.BACKGROUND)
        if noise:
            rho = c.download_density()
            nrng = np.random.default_rng(11)
            rho += np.where(rho < 0.2, noise * nrng.random(shape), 0.0)
            c.upload_density(rho)
            del rho
        vv = abs(np.linalg.det(lat)) / nv

        def step():
            c.set_option(6, 1)
            c.vacuum_assign(vac_tol, vv)
            n = c.assign('neargrid')
            return n, c.refine(mode, iters)
        for _ in range(max(1, warmup)):
            step()
        c.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            n, log = step()
        c.sync()
        dt = (time.perf_counter() - t0) / steps
        c.enable_timing(True)
        c.kernel_time_reset()
        for _ in range(2):
            step()
        c.sync()
        legs[name] = {'workload': f'{size}^3, {what}; neargrid assign + neargrid edge refinement {mode}:{iters}',
                      'value': nv / dt / 1e6, 'unit': 'Mvoxels/s', 'ms_per_step': dt * 1e3, 'steps': steps, 'basins': int(n),
                      'refine_log': log[:4], 'ns_per_voxel': dt / nv * 1e9, 'ns_per_voxel_headline': headline_ns_per_voxel,
                      'per_voxel_time_vs_headline': dt / nv * 1e9 / headline_ns_per_voxel,
                      'trapping_boxes': {'count': c.box_stats()[0], 'voxel_fraction': c.box_stats()[1] / nv},
                      'slow_path_trajectories(assign,refine)': list(c.slow_path_stats()),
                      'stage_ms_avg': {nm: (lambda t: t[0] / t[1] if t[1] else 0.0)(c.kernel_time(i)) for i, nm in enumerate(stage_names)}}
        c.close()
    return legs


def batch_leg(size, steps, mode, iters, lattice, atoms, background, jobs=2):
    """Round 5: the SAME step on `jobs` independent densities at once -- a context (own stream, own buffers) and a host thread per
    job, as a batch of CHGCARs would be run (ctypes releases the GIL inside the library).  The card overlaps one job's latency-bound
    stretches (the ~20 small launches of the region growth, the tails of the trace kernels, the host's wait) with the other job's
    kernels.  A throughput number for batches; the headline stays the single job, whose step time is what one `bader` run sees."""
    import threading
    from pybader_amd import _lib
    from pybader_amd.interface import distance_matrix, gradient_transform
    shape = (size,) * 3
    vl = np.divide(lattice, shape)
    vv = abs(np.linalg.det(lattice)) / float(np.prod(shape))
    ctxs = []
    for _ in range(jobs):
        c = _lib.Context(0)
        c.set_grid(shape, distance_matrix(vl), gradient_transform(vl))
        c.synth_density(lattice, atoms, background)
        ctxs.append(c)
    results = [None] * jobs
    errors = []

    def work(j, k):
        try:
            c = ctxs[j]
            for _ in range(k):
                c.set_option(6, 1)
                c.vacuum_assign(None, vv)
                results[j] = c.assign_refine('neargrid', mode, iters)
        except Exception as e:   # noqa: BLE001 (a thread's failure must fail the leg, not vanish)
            errors.append(repr(e))
    for j in range(jobs):
        work(j, 3)
        ctxs[j].sync()
    per_job = max(10, steps)   # (every job runs as many steps as the headline: a short leg measures the threads' start-up)
    threads = [threading.Thread(target=work, args=(j, per_job)) for j in range(jobs)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for c in ctxs:
        c.sync()
    dt = (time.perf_counter() - t0) / (per_job * jobs)
    for c in ctxs:
        c.close()
    if errors or any(r is None for r in results):
        raise SystemExit(f'batch leg failed: {errors}')
    if len({(int(r[0]), tuple(map(tuple, r[1]))) for r in results}) != 1:
        raise SystemExit(f'batch leg: the jobs disagree on the same density: {results}')
    nv = float(size) ** 3
    return {'workload': f'{jobs} independent {size}^3 densities in flight (a context, a stream and a host thread each), the headline\'s step on each',
            'jobs_in_flight': jobs, 'steps': per_job * jobs, 'ms_per_step': dt * 1e3, 'value': nv / dt / 1e6, 'unit': 'Mvoxels/s',
            'whole_path_frac_of_hbm_roofline': BYTES_PATH * nv / dt / 1e9 / HBM_PEAK_GBS,
            'basins': [int(r[0]) for r in results], 'refine_log': [list(map(list, r[1][:2])) for r in results]}


def dropin_leg(ctx, size=256, k=6, reps=3):
    """bader_calc + refine through pybader_amd.thread_handlers on a 216-atom density (host numpy arrays at the boundary)"""
    from pybader_amd import _lib, synth, thread_handlers, utils
    from pybader_amd.interface import distance_matrix, gradient_transform
    rng = np.random.default_rng(5)
    cells = np.stack(np.meshgrid(*(np.arange(k),) * 3, indexing='ij'), -1).reshape(-1, 3)
    frac = (cells + 0.5 + 0.18 * (rng.random(cells.shape) - 0.5)) / k
    atoms = np.concatenate([frac, 0.09 + 0.04 * rng.random((len(frac), 1)), 2.0 + 6.0 * rng.random((len(frac), 1))], 1)
    shape = (size,) * 3
    lattice = synth.CUBIC6
    vl = np.divide(lattice, shape)
    dm, tg = distance_matrix(vl), gradient_transform(vl)
    c2 = _lib.Context(0)
    c2.set_grid(shape, dm, tg)
    c2.synth_density(lattice, atoms, synth.BACKGROUND)
    rho = c2.download_density()
    c2.close()
    thread_handlers.VERBOSE = False

    def pairs(inside_resident):
        times, n, stats, out = [], 0, (0, 0), None
        for _ in range(reps):
            vol = np.zeros(shape, np.int32)
            t0 = time.perf_counter()
            vol, _, _ = utils.vacuum_assign(rho, vol, float('nan'), rho, 1.0)       # Bader.volumes_init (interface.py:449-469)
            bmax, vol = thread_handlers.bader_calc('neargrid', rho, vol, dm, tg, 1)
            thread_handlers.refine('neargrid', ('changed', 2), rho, vol, dm, tg, 1)
            times.append(time.perf_counter() - t0)
            n, stats, out = bmax.shape[0], _lib.default_context().box_stats(), (bmax, vol)
        return min(times), n, stats, out

    # (i) as the INTEGRATION.md patch runs them: inside `resident(density)` (Bader.__call__ is wrapped in it) -- the density
    # is uploaded once per __call__, the label array stays on the device between bader_calc and refine
    dctx = _lib.default_context()
    dctx.set_grid(shape, dm, tg)
    dctx.upload_density(rho)                 # (first touch: allocations, pinned staging buffers)
    with utils.resident(rho):
        t_up = time.perf_counter()
        utils.ensure_density(dctx, rho)
        dctx.sync()
        t_up = time.perf_counter() - t_up
        best, n, stats, (bmax_a, vol_a) = pairs(True)
        vol_a = vol_a.copy()
    # (ii) bare calls, nothing promised about the host arrays: density + labels travel on every call (round 2's figure)
    bare, _, _, (bmax_b, vol_b) = pairs(False)
    return {'workload': f'{size}^3 grid, {len(atoms)} atoms, volumes_init + thread_handlers.bader_calc + refine (changed,2) as Bader.__call__ issues them, host arrays at the '
                        'boundary, inside utils.resident(density) as the INTEGRATION.md patch of Bader.__call__ runs them: the '
                        'narrowed label download of bader_calc is in, the density upload is once per __call__ (reported beside)',
            'value': float(size) ** 3 / best / 1e6, 'unit': 'Mvoxels/s', 'ms_per_call_pair': best * 1e3, 'basins': int(n),
            'density_upload_ms_once_per_call': t_up * 1e3,
            'ms_per_call_pair_bare': bare * 1e3,
            'outputs_identical': bool(np.array_equal(vol_a, vol_b) and np.array_equal(bmax_a, bmax_b)),
            'trapping_boxes': {'count': int(stats[0]), 'voxel_fraction': stats[1] / float(size) ** 3}}


def spawn_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start N rank processes of this script (fresh interpreters
    that have not touched the GPU; this parent never does), wait, relay rank 0's output.  Returns the exit code."""
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = set(range(n))
    while alive:                         # a rank that dies leaves its peers hanging in a collective: end them
        time.sleep(0.2)
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0:
                print(f'bench.py: rank {r} exited with code {code}', file=sys.stderr)
                rc = rc or (code if code > 0 else 1)
                for q in alive:
                    procs[q].kill()
    out0 = procs[0].stdout.read().decode()   # one JSON line: far below the pipe buffer
    sys.stdout.write(out0)
    sys.stdout.flush()
    if rc == 0 and not any(line.startswith('{') for line in out0.splitlines()):
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--method', default='neargrid', choices=['neargrid', 'ongrid'])
    ap.add_argument('--refine', default='changed:2')
    ap.add_argument('--cpu-size', type=int, default=320)
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-config5', action='store_true', help='skip the ongrid configuration timed after the headline one')
    ap.add_argument('--no-dropin', action='store_true', help='skip the thread_handlers leg on the 216-atom 256^3 density')
    ap.add_argument('--no-stage-steps', action='store_true', help='skip the 3 extra steps that time the other stages (profiler runs: only warm-up + timed steps on the card)')
    ap.add_argument('--no-batch', action='store_true', help='skip the two-densities-in-flight throughput leg')
    ap.add_argument('--no-user-legs', action='store_true', help='skip the triclinic / 216-atom / noisy-vacuum legs at the headline size')
    ap.add_argument('--no-sustained', action='store_true', help='skip the leg that repeats the step for --sustain-s seconds after the timed region')
    ap.add_argument('--sustain-s', type=float, default=2.5)
    ap.add_argument('--no-odd', action='store_true', help='skip the leg on a grid that is not made of whole 8^3 bricks (500 x 504 x 420)')
    ap.add_argument('--halo', type=int, default=None,
                    help='label planes valid each side of a slab (default 16 for N > 1: a retrace stops when it enters a '
                         'trapping region; whatever still leaves the valid planes is finished by remote path queries)')
    ap.add_argument('--table-margin', type=int, default=None,
                    help='planes of gradient-field table each side of a slab (N > 1; default max(32, nx / 16))')
    args = ap.parse_args()
    mode, iters = args.refine.split(':')
    iters = int(iters)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1 and 'RANK' not in os.environ:
            raise SystemExit(spawn_ranks(args.gpus))   # nothing has touched the GPU in this process
        args.gpus = world

    from pybader_amd import _lib, comm as xcomm, slab, synth
    from pybader_amd.interface import distance_matrix, gradient_transform

    store = None
    dev_index = 0
    if world > 1:
        # one process per GPU; no PyTorch: host rendezvous over TCP (pybader_amd/comm.py), planes and counters over
        # RCCL through the library's own C ABI (csrc/comm.h)
        store = xcomm.SocketStore(rank, world)
        dev_index = local_rank % max(1, _lib.load().xb_device_count())   # == local_rank on a full node
        ctx = _lib.Context(dev_index)
        comm = xcomm.RcclComm(ctx, store)
        # A node with a GPU per rank must run on the device transport: the host-staged fallback is ~35x slower and would
        # pass for a valid scaling number.  Fail loudly instead; fewer GPUs than ranks (the one-GPU
        # rehearsal) may fall back.
        n_dev = _lib.load().xb_device_count()
        if comm.transport != 'rccl' and n_dev >= world and not os.environ.get('XB_ALLOW_STAGED'):
            print(f'bench.py: rank {rank}: {n_dev} GPUs for {world} ranks but the device transport is not in use: '
                  f'{comm.init_error}', file=sys.stderr, flush=True)
            if rank == 0:
                print(json.dumps({'metric': f'Mvoxels/s {args.method} assign+refine on {args.size}^3 grid', 'value': None,
                                  'n_gpus': world, 'error': 'device transport (RCCL) unavailable',
                                  'config': {'transport': comm.transport, 'transport_error': comm.init_error}}), flush=True)
            store.close()
            raise SystemExit(4)
    else:
        class _Solo:
            rank, size, transport = 0, 1, 'none'
            def barrier(self):
                pass
            def max_float(self, x):
                return x
        comm = _Solo()
        ctx = _lib.Context(0)

    shape = (args.size,) * 3
    lattice, atoms, background = synth.CUBIC6, synth.ATOMS8, synth.BACKGROUND
    vl = np.divide(lattice, shape)
    dm, tg = distance_matrix(vl), gradient_transform(vl)
    voxel_volume = abs(np.linalg.det(lattice)) / float(np.prod(shape))

    halo = args.halo if args.halo is not None else (16 if world > 1 else 8)
    runner = slab.SlabRunner(slab.GpuBackend(ctx, dev_index), comm, shape, dm, tg, halo=halo)
    windowed = runner.enable_table_window(args.table_margin) if world > 1 else False
    ctx.synth_density(lattice, atoms, background)      # inputs resident in HBM before timing starts
    # HIP events inside the timed region: the dominant kernel's timer alone (an event pair between dependent kernels costs the
    # stream ~10 us; seven nested stage timers were ~0.09 ms of a 3.5 ms step).  The other stages are timed in extra steps
    # after the timed region.
    dom_timer = 6 if args.method == 'neargrid' else 1       # xb_kernel_time: 6 the trace kernel, 1 the ongrid pointer pass
    ctx.enable_timing(only=[dom_timer])
    if 'XB_OPT_DBG' in os.environ:
        ctx.set_option(3, int(os.environ['XB_OPT_DBG']))
    for key in (1, 2, 17):                              # (A/B runs: regions, the cross-check bits, the kill schedule)
        if f'XB_OPT_{key}' in os.environ:
            ctx.set_option(key, int(os.environ[f'XB_OPT_{key}']))
    if 'XB_OPT_EC_GROUPS' in os.environ:
        ctx.set_option(4, int(os.environ['XB_OPT_EC_GROUPS']))

    def step():
        ctx.set_option(6, 1)                            # no table carried over from the previous step
        ctx.vacuum_assign(None, voxel_volume)           # Bader.volumes_init: labels := 0 (no vacuum)
        if os.environ.get('XB_TWO_CALLS'):              # (A/B: the two library calls of rounds 1-4, two host waits)
            n = runner.assign(args.method)              # Bader.bader_calc
            log = runner.refine(mode, iters)            # Bader.refine_volumes
            return n, log
        return runner.assign_refine(args.method, mode, iters)   # ... both as Bader.__call__ issues them back to back: one host wait (round 5)

    def fence():
        ctx.sync()
        comm.barrier()
        ctx.sync()

    for _ in range(args.warmup):
        step()
    ctx.kernel_time_reset()
    if runner.timing is not None:
        runner.timing.clear()
    fence()
    waits0 = ctx.host_waits()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_basins, log = step()
    fence()
    dt = comm.max_float(time.perf_counter() - t0)       # max over ranks
    host_waits = (ctx.host_waits() - waits0 - 2) / args.steps   # waits of this rank for its card inside library calls (less the fence's two)

    # The same step back to back for a few seconds (outside the timed region): the timed K steps above are tens of milliseconds, which
    # a once-a-second activity sampler never sees; this gives it something to see and shows what the step costs when sustained.
    sustained = None
    if not args.no_sustained:
        n_sus = max(args.steps, min(4000, int(args.sustain_s / max(dt / args.steps, 1e-5))))
        n_sus = int(comm.max_float(float(n_sus)))       # the same count on every rank
        fence()
        ts = time.perf_counter()
        for _ in range(n_sus):
            step()
        fence()
        dts = comm.max_float(time.perf_counter() - ts)
        sustained = {'steps': n_sus, 'seconds': dts, 'ms_per_step': dts / n_sus * 1e3}

    nvox = float(np.prod(shape))
    # N > 1: where a slab step spends its time, from two EXTRA steps with a device sync around every scheduler phase
    # (the timed steps above run without those syncs)
    phase_ms = None
    if world > 1:
        runner.timing = {}
        for _ in range(2):
            step()
        fence()
        phase_ms = {k: v / 2 * 1e3 for k, v in runner.timing.items()}
        runner.timing = None
    # outside the timed region: every voxel carries exactly one basin of the merged numbering, over all ranks together (a wrong
    # multi-rank map must be loud, not fast)
    _, vols = ctx.charge_sum(1.0, n_basins)
    vols = [int(v) for v in vols]
    if world > 1:
        vols = comm.sum(*vols)
    if sum(vols) != int(nvox) or (vols and min(vols) <= 0):
        raise SystemExit(f'map check failed: basin volumes {vols[:16]}... sum {sum(vols)} of {int(nvox)} voxels')
    ms_per_step = dt / args.steps * 1e3
    # HIP-event timings on the library's own stream (xb_kernel_time): the dominant kernel from the timed region itself, the
    # other stages from three extra steps with every timer on
    stage_names = ['assign_after_masks(walk_list+k_brick_records+k_ng_trace)', 'k_og_masks', 'edge_find', 'k_refine_trace',
                   'masks+trapping_regions', 'k_brick_masks', 'k_ng_trace', 'k_brick_records']   # (6: whichever trace kernel ran)
    dom_time = ctx.kernel_time(dom_timer)
    ctx.enable_timing(True)
    ctx.kernel_time_reset()
    for _ in range(0 if args.no_stage_steps else 3):
        step()
    fence()
    tm = {name: ctx.kernel_time(i) for i, name in enumerate(stage_names)}
    tm[stage_names[dom_timer]] = dom_time
    ctx.enable_timing(only=[dom_timer])
    avg = {k: (ms / n if n else 0.0) for k, (ms, n) in tm.items()}
    own_frac = (runner.x_range[1] - runner.x_range[0]) / shape[0]
    # roofline.achieved is the WHOLE PATH -- 25 algorithmic bytes per voxel (assign 12 + first refine sweep
    # 13) over the step time; the single kernels (HIP events around each launch alone) sit under roofline.kernels with the
    # stage's own algorithmic bytes (12 B per voxel this rank labels for an assignment kernel, 13 for the refinement's)
    step_s = dt / args.steps
    achieved = BYTES_PATH * nvox / step_s / 1e9
    kernels = {}
    for name, bytes_per_voxel in (('k_brick_masks', BYTES_ASSIGN), ('k_ng_trace', BYTES_ASSIGN), ('k_brick_records', BYTES_ASSIGN),
                                  ('k_og_masks', BYTES_ASSIGN), ('edge_find', 13), ('k_refine_trace', 13)):
        if avg[name] > 0:
            gbs = bytes_per_voxel * nvox * own_frac / (avg[name] * 1e-3) / 1e9
            kernels[name] = {'ms_avg': avg[name], 'launches': int(tm[name][1]), 'algorithmic_bytes_per_voxel': bytes_per_voxel,
                             'achieved': gbs, 'frac': gbs / HBM_PEAK_GBS}
    dom = max(kernels, key=lambda k: kernels[k]['ms_avg']) if kernels else None
    # HBM bytes of one step (FETCH_SIZE + WRITE_SIZE over every kernel of a one-step run) from the committed summary of the
    # rocprofv3 --pmc passes of this very command, dropped when that summary was taken from other kernel sources
    traffic, traffic_src = None, 'only recorded for the 512^3 one-GPU configuration'
    if args.size == 512 and world == 1:
        traffic, traffic_src = traffic_from_profile(None, args.method)
        if dom and dom != 'edge_find':
            kernels[dom]['traffic'], _ = traffic_from_profile({'k_ng_trace': 'k_ng_trace_g'}.get(dom, dom), args.method)

    out = {
        'metric': f'Mvoxels/s {args.method} assign+refine on {args.size}^3 grid',
        'value': nvox / (dt / args.steps) / 1e6,
        'unit': 'Mvoxels/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': ms_per_step,
        'higher_is_better': True,
        'scaling': 'strong',
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {'workload': f'{args.size}^3 synthetic 8-Gaussian-atom cubic cell (BASELINE config 3), '
                               f'{args.method} assign + neargrid edge refinement {mode}:{iters}, '
                               'density resident in HBM',
                   'grid': list(shape), 'method': args.method, 'refine_mode': [mode, iters],
                   'parallelism': f'{world} axis-0 slab(s), density replicated, halo {runner.halo}, transport {comm.transport}, '
                                  f'table window {"slab+-%d planes" % runner.table_margin if windowed else "whole grid"}',
                   'host_waits_per_step': host_waits, 'device_driven_slab_step': bool(runner.n_device_steps),
                   'basins': int(n_basins), 'refine_log': log,
                   'trapping_boxes': {'count': ctx.box_stats()[0], 'voxel_fraction': ctx.box_stats()[1] / nvox}},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                     'traffic': traffic,
                     'hbm_moved_gbs': (traffic / step_s / 1e9) if traffic else None,   # what the memory system actually moves per second (quoted traffic / measured step time)
                     'traffic_note': 'NOT measured in this run: quoted from the committed PMC summary ' + str(traffic_src) +
                                     ' -- HBM bytes of ONE steady-state step = (FETCH_SIZE + WRITE_SIZE) x 1024 summed over its kernels, '
                                     'separate rocprofv3 --pmc passes of this command with the same kernel sources (hash checked; other '
                                     'sources: null); 8 B/lane row loads, 32 B gathers and record stores, so the gfx950 x2 rule for '
                                     '16 B/lane streams does not apply',
                     'algorithmic_bytes_per_voxel': BYTES_PATH, 'ms_per_step': ms_per_step,
                     'dominant_kernel': dom, 'kernels': kernels,
                     'kernel_timing': f'HIP events on the library stream: {stage_names[dom_timer]} inside the timed region (the only timer on there); the other stages in 3 extra steps with every timer on',
                     'stage_ms_avg': avg},
    }

    out['sustained'] = sustained
    cpu_tasks = []     # every CPU-oracle leg runs after the last GPU leg: the card's work stays in one stretch of the run

    # BASELINE config 5 (ongrid assign + neargrid edge refinement, the divergent-path stress case) in the same run, so
    # that it is driver-timed too: same grid, same density, K steps, its own HIP-event stage times
    if world == 1 and args.method == 'neargrid' and not args.no_config5:
        def step5():
            ctx.set_option(6, 1)
            ctx.vacuum_assign(None, voxel_volume)
            n5 = runner.assign('ongrid')
            return n5, runner.refine(mode, iters)
        ctx.enable_timing(False)
        step5()
        fence()
        t5 = time.perf_counter()
        for _ in range(args.steps):
            n5, log5 = step5()
        fence()
        dt5 = (time.perf_counter() - t5) / args.steps
        ctx.enable_timing(True)               # (stage times: two extra steps, outside the timed loop)
        ctx.kernel_time_reset()
        for _ in range(2):
            step5()
        fence()
        tm5 = {name: ctx.kernel_time(i) for i, name in enumerate(
            ['-', 'k_og_masks', 'edge_find', 'k_refine_trace', 'regions+records_for_retraces', 'k_brick_masks'])}
        out['config5'] = {'workload': f'{args.size}^3 same density, ongrid assign + neargrid edge refinement {mode}:{iters}',
                          'value': nvox / dt5 / 1e6, 'unit': 'Mvoxels/s', 'ms_per_step': dt5 * 1e3, 'steps': args.steps,
                          'basins': int(n5), 'refine_log': log5,
                          'whole_path_frac_of_hbm_roofline': BYTES_PATH * nvox / dt5 / 1e9 / HBM_PEAK_GBS,
                          'stage_ms_avg': {k: (ms / n if n else 0.0) for k, (ms, n) in tm5.items() if k != '-'}}
        t5b, t5src = traffic_from_profile(None, 'ongrid') if args.size == 512 else (None, 'only recorded at 512^3')
        out['config5']['traffic'] = t5b
        out['config5']['traffic_note'] = 'HBM bytes of one step (FETCH_SIZE + WRITE_SIZE over its kernels): ' + str(t5src)
        def cpu_config5():
            # ongrid + refinement against the CPU oracle on the same bounded sample as the neargrid leg below
            import oracle
            shp = (args.cpu_size,) * 3
            rho_s = oracle.synth_density(shp, lattice, atoms, background)
            vls = np.divide(lattice, shp)
            dm_s, tg_s = distance_matrix(vls), gradient_transform(vls)
            t0 = time.perf_counter()
            bmax5, want5 = oracle.bader_calc('ongrid', rho_s, np.zeros(shp, np.int32), dm_s, tg_s, 1)
            oracle.refine('neargrid', (mode, iters), rho_s, want5, dm_s, tg_s, 1)
            cpu5 = time.perf_counter() - t0
            c5 = _lib.Context(0)
            c5.set_grid(shp, dm_s, tg_s)
            c5.upload_density(rho_s)
            c5.vacuum_assign(None, 1.0)
            c5.assign('ongrid')
            c5.refine(mode, iters)
            got5 = c5.download_labels(want5.dtype)
            out['config5']['cpu_baseline'] = {'value': float(args.cpu_size) ** 3 / cpu5 / 1e6, 'unit': 'Mvoxels/s', 'cores': 1, 'kind': 'port',
                                              'sample': f'{args.cpu_size}^3 grid, ongrid assign + refine ({mode},{iters}), single thread C port',
                                              'gpu_map_equals_cpu_map': bool(np.array_equal(got5, want5) and np.array_equal(c5.maxima(), bmax5))}
            c5.close()
        if not args.no_cpu:
            cpu_tasks.append(cpu_config5)

    # A grid the 8^3 brick lattice does not divide (round 4: VERDICT r3 #4 -- FFT grids like 60, 84, 108, 140, 180, 500 are
    # everyday CHGCAR sizes): the same pipeline, the bricks the grid cuts handled in place.  Same cell and atoms, K steps,
    # and the map of a half-size grid of the same kind against the CPU oracle.
    if world == 1 and args.method == 'neargrid' and not args.no_odd and args.size == 512:
        oshape = (500, 504, 420)
        # (the cell shrinks with the grid: voxels of the headline's size and shape, atoms at the same fractional positions)
        olat = lattice * (np.array(oshape, np.float64) / float(args.size))[:, None]
        ovl = np.divide(olat, oshape)
        odm, otg = distance_matrix(ovl), gradient_transform(ovl)
        co = _lib.Context(0)
        co.set_grid(oshape, odm, otg)
        co.synth_density(olat, atoms, background)
        ovv = abs(np.linalg.det(olat)) / float(np.prod(oshape))

        def step_odd():
            co.set_option(6, 1)
            co.vacuum_assign(None, ovv)
            n_o = co.assign('neargrid')
            return n_o, co.refine(mode, iters)
        for _ in range(max(1, args.warmup)):
            step_odd()
        co.sync()
        t_o = time.perf_counter()
        for _ in range(args.steps):
            n_o, log_o = step_odd()
        co.sync()
        dt_o = (time.perf_counter() - t_o) / args.steps
        co.enable_timing(True)                # (stage times: two extra steps, outside the timed loop)
        co.kernel_time_reset()
        for _ in range(2):
            step_odd()
        co.sync()
        onvox = float(np.prod(oshape))
        out['odd_grid'] = {'workload': f'{oshape[0]} x {oshape[1]} x {oshape[2]} grid (not whole 8^3 bricks in x and z), same cell and atoms, '
                                       f'neargrid assign + neargrid edge refinement {mode}:{iters}',
                           'value': onvox / dt_o / 1e6, 'unit': 'Mvoxels/s', 'ms_per_step': dt_o * 1e3, 'steps': args.steps,
                           'basins': int(n_o), 'refine_log': log_o,
                           'ns_per_voxel': dt_o / onvox * 1e9, 'ns_per_voxel_headline': step_s / nvox * 1e9,
                           'per_voxel_time_vs_headline': (dt_o / onvox) / (step_s / nvox),
                           'trapping_boxes': {'count': co.box_stats()[0], 'voxel_fraction': co.box_stats()[1] / onvox},
                           'slow_path_trajectories(assign,refine)': list(co.slow_path_stats()),
                           'stage_ms_avg': {name: (lambda t: t[0] / t[1] if t[1] else 0.0)(co.kernel_time(i)) for i, name in enumerate(stage_names)}}
        co.close()

        def cpu_odd():
            import oracle
            hshape = (250, 252, 210)
            rho_h = oracle.synth_density(hshape, olat, atoms, background)
            hvl = np.divide(olat, hshape)
            hdm, htg = distance_matrix(hvl), gradient_transform(hvl)
            t0 = time.perf_counter()
            bmax_h, want_h = oracle.bader_calc('neargrid', rho_h, np.zeros(hshape, np.int32), hdm, htg, 1)
            oracle.refine('neargrid', (mode, iters), rho_h, want_h, hdm, htg, 1)
            cpu_h = time.perf_counter() - t0
            ch = _lib.Context(0)
            ch.set_grid(hshape, hdm, htg)
            ch.upload_density(rho_h)
            ch.vacuum_assign(None, 1.0)
            ch.assign('neargrid')
            ch.refine(mode, iters)
            got_h = ch.download_labels(want_h.dtype)
            out['odd_grid']['cpu_baseline'] = {'value': float(np.prod(hshape)) / cpu_h / 1e6, 'unit': 'Mvoxels/s', 'cores': 1, 'kind': 'port',
                                               'sample': f'{hshape[0]} x {hshape[1]} x {hshape[2]} grid, neargrid assign + refine ({mode},{iters}), single thread C port',
                                               'gpu_map_equals_cpu_map': bool(np.array_equal(got_h, want_h) and np.array_equal(ch.maxima(), bmax_h))}
            ch.close()
        if not args.no_cpu:
            cpu_tasks.append(cpu_odd)

    if world == 1 and args.method == 'neargrid' and not args.no_batch and args.size <= 640:
        out['batch_two_in_flight'] = batch_leg(args.size, args.steps, mode, iters, lattice, atoms, background)
        out['batch_two_in_flight']['vs_single_job'] = out['batch_two_in_flight']['value'] / out['value']
    if world == 1 and args.method == 'neargrid' and not args.no_user_legs and args.size >= 256:
        out['user_legs'] = user_legs(args.size, min(args.steps, 10), args.warmup and 2, mode, iters, step_s / nvox * 1e9, stage_names)

    # The Python drop-in (pybader_amd.thread_handlers, the reference's call signatures) on a density the headline does not
    # flatter: 216 atoms on a 256^3 grid (more maxima than round 1's 64 seed cubes), host arrays in, host arrays out --
    # label upload / download and the density upload included (PCIe), which `value` above never includes.
    if world == 1 and args.method == 'neargrid' and not args.no_dropin:
        out['dropin_many_atoms'] = dropin_leg(ctx)

    out['config']['map_check'] = {'voxels_labelled': int(sum(vols)), 'smallest_basin': int(min(vols)) if vols else 0, 'largest_basin': int(max(vols)) if vols else 0}
    mem = ctx.memory_stats()
    out['config']['device_bytes_per_rank'] = {'total': mem[0], 'table': mem[1], 'scratch': mem[2], 'per_voxel_of_the_grid': mem[0] / nvox}
    out['config']['slow_path_trajectories(assign,refine)'] = list(ctx.slow_path_stats())
    out['config']['retraces_redone_from_rho'] = ctx.deferred_stats()
    out['config']['retrace_passes_with_walkers'] = runner.n_fallbacks   # passes in which some retrace left a slab's valid planes (0 on one GPU)
    if phase_ms is not None:
        out['config']['slab_phase_ms_avg'] = phase_ms
        out['config']['halo_bytes_sent_per_step'] = ctx.comm_bytes_sent() / (args.warmup + args.steps + 2) if comm.transport == 'rccl' else None
        # what every rank's RCCL communicator says about itself (ranks, rank, device, library version): the first run on a
        # real node proves with this that it saw N ranks on N devices
        info = ctx.comm_info() if comm.transport == 'rccl' else {'nccl_comm_count': -1, 'nccl_user_rank': rank, 'nccl_device': dev_index, 'rccl_version': -1}
        infos = comm.allgather([info['nccl_comm_count'], info['nccl_user_rank'], info['nccl_device'], info['rccl_version']])
        out['config']['rccl'] = {'comm_count_per_rank': [int(i[0]) for i in infos], 'user_rank_per_rank': [int(i[1]) for i in infos],
                                 'device_per_rank': [int(i[2]) for i in infos], 'version': int(infos[0][3])}
    for task in cpu_tasks:
        task()
    if rank == 0 and world == 1 and not args.no_cpu:
        cb, (rho_s, dm_s, tg_s, want, bmax) = cpu_baseline(args.cpu_size, args.method, mode, iters,
                                                           lattice, atoms, background, args.size)
        # the same sample through the GPU path doubles as an end-of-run parity check
        c2 = _lib.Context(0)
        c2.set_grid(rho_s.shape, dm_s, tg_s)
        c2.upload_density(rho_s)
        c2.vacuum_assign(None, 1.0)
        c2.assign(args.method)
        c2.refine(mode, iters)
        got = c2.download_labels(want.dtype)
        cb['gpu_map_equals_cpu_map'] = bool(np.array_equal(got, want) and np.array_equal(c2.maxima(), bmax))
        c2.close()
        out['cpu_baseline'] = cb
    elif rank == 0:
        out['cpu_baseline'] = None

    if rank == 0:
        print(json.dumps(out))
    if store is not None:
        comm.barrier()
        ctx.close()
        store.close()


if __name__ == '__main__':
    main()
