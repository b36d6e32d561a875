"""N>1 path on CPU: world_size-2/3 gloo groups run the product's slab scheduler (slab.py: slab
ranges, halo plan, plane exchange, maxima-table merge, iteration driver) over a host backend built on
the oracle; the assembled map must equal the single-rank result and the reference's golden map."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import load_golden
from pybader_amd import slab

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(n, args, tmp_path, port, halo=4, transport='gloo'):
    out = str(tmp_path / 'out.npz')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(n))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'slab_worker.py')] + args + [out, str(halo), transport],
                                      env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=600)
        assert p.returncode == 0, o.decode()[-3000:]
    return np.load(out)


def test_slab_ranges_and_halo_plan():
    assert slab.slab_ranges(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert slab.slab_ranges(8, 8) == [(k, k + 1) for k in range(8)]
    # whole bricks, a brick or more per rank: the slabs do not cut bricks
    assert slab.slab_ranges(512, 3) == [(0, 176), (176, 344), (344, 512)]
    assert slab.slab_ranges(64, 8) == [(8 * k, 8 * k + 8) for k in range(8)]
    assert slab.slab_ranges(40, 8)[0] == (0, 5)            # fewer bricks than ranks: plane by plane
    rngs = slab.slab_ranges(16, 4)
    for r in range(4):
        sends, recvs = slab.halo_plan(rngs, r, 2, 16)
        got = sorted(p for _, a, b in recvs for p in range(a, b))
        x0, x1 = rngs[r]
        assert got == sorted({(x0 - 2) % 16, (x0 - 1) % 16, x1 % 16, (x1 + 1) % 16})
        # every send of r is a recv of the peer
        for peer, a, b in sends:
            _, precv = slab.halo_plan(rngs, peer, 2, 16)
            assert (r, a, b) in precv
    # halo wider than a slab: planes come from several owners
    sends, recvs = slab.halo_plan(slab.slab_ranges(12, 6), 0, 3, 12)
    assert sorted(p for _, a, b in recvs for p in range(a, b)) == [2, 3, 4, 9, 10, 11]


def test_merge_maxima_tables():
    t = [(np.array([50, 7]), np.array([40, 3])), (np.array([7, 90]), np.array([1, 60]))]
    assert slab.merge_maxima_tables(t).tolist() == [7, 50, 90]


@pytest.mark.parametrize('n,case,method,mode,iters,port,halo', [
    (2, 'c12_cubic', 'neargrid', 'changed', 2, 29611, 4),
    (3, 'c40x48x56_tric', 'neargrid', 'all', -1, 29612, 4),
    (2, 'c48_cubic_vac', 'neargrid', 'changed', 2, 29613, 4),
    (2, 'c40x48x56_tric', 'ongrid', 'all', 3, 29614, 6),
    (3, 'c40x48x56_tric', 'ongrid', 'all', 3, 29615, 3),      # narrow halo: traces escape -> fallback
    (3, 'c40x48x56_tric', 'ongrid', 'changed', 3, 29616, 6),  # relabelled voxels: edge_check across slabs
])
def test_slabs_equal_single_rank_and_golden(n, case, method, mode, iters, port, halo, tmp_path):
    import oracle
    from conftest import case_density
    r = run_world(n, [case, method, mode, str(iters)], tmp_path, port, halo)
    if halo == 3:
        assert int(r['fallbacks']) > 0, "the narrow-halo case is meant to exercise the escape fallback"
    g = load_golden(case)
    if method == 'neargrid':
        assert np.array_equal(r['pre'], g['ng_F'].astype(np.int32))
        key = f"ng_{mode}_{'inf' if iters < 0 else iters}"
        assert np.array_equal(r['post'], g[key].astype(np.int32))
        assert np.array_equal(np.array(np.unravel_index(r['maxima'], g['ng_F'].shape)).T, g['ng_bader_max'])
    else:
        assert np.array_equal(r['pre'], g['og_main'].astype(np.int32))
        rho = case_density(g)
        v = g['og_main'].astype(np.int32)
        log = []
        oracle.refine('neargrid', (mode, iters), rho, v, g['dist_mat'], g['T_grad'], 1, log=log)
        assert np.array_equal(r['post'], v)
        assert np.array_equal(r['log'], np.array(log, np.int64).reshape(-1, 2))


def test_slabs_over_the_products_own_host_transport(tmp_path):
    """the same scheduler over pybader_amd.comm.SocketStore / HostComm (file rendezvous + TCP star), 3 ranks"""
    r = run_world(3, ['c40x48x56_tric', 'ongrid', 'all', '3'], tmp_path, 29621, 3, transport='tcp')
    assert int(r['fallbacks']) > 0
    g = load_golden('c40x48x56_tric')
    assert np.array_equal(r['pre'], g['og_main'].astype(np.int32))
    r2 = run_world(2, ['c12_cubic', 'neargrid', 'changed', '2'], tmp_path, 29622, 4, transport='tcp')
    g2 = load_golden('c12_cubic')
    assert np.array_equal(r2['post'], g2['ng_changed_2'].astype(np.int32))


def test_walker_rounds_bookkeeping_with_a_mock_backend():
    """SlabRunner._migrate_walkers on CPU: four ranks (threads) with a mock backend whose walkers need a given number of
    hops.  Every walker's result must reach the owner of its start voxel exactly once, walkers must only be handed to
    the rank that owns the plane they arrived on, and all ranks must leave the loop in the same round."""
    import threading

    shape, n = (32, 4, 4), 4
    nyz = shape[1] * shape[2]
    ranges = slab.slab_ranges(shape[0], n)
    barrier = threading.Barrier(n)
    slots = [None] * n

    class Comm:
        def __init__(self, rank):
            self.rank, self.size = rank, n

        def gather_rows(self, rows):
            slots[self.rank] = np.ascontiguousarray(rows, np.int64)
            barrier.wait()
            out = np.concatenate(list(slots))
            barrier.wait()
            return out

    def owner_of(voxel):
        return next(r for r, (a, b) in enumerate(ranges) if a <= voxel // nyz < b)

    class Backend:
        WALKER_WORDS = 10

        def __init__(self, rank):
            self.rank, self.applied, self.continued, self.rounds = rank, [], 0, 0
            x0, x1 = ranges[rank]
            # three walkers per rank, needing 0, 1 and 2 more hops; they arrive on the next slab's first plane
            rows = np.zeros((3, 10), np.int64)
            for k in range(3):
                v = (x0 * nyz + k) | ((100 + rank) << 32)                  # start voxel | label
                arrive = (x1 % shape[0]) * nyz
                rows[k] = [v, (x1 - 1) * nyz | (arrive << 32), 0, 0, 0, 0, 0, 0, 0, k]
            self.rows = rows

        def walkers(self):
            return self.rows, np.zeros(0, np.int64)

        def walkers_continue(self, rows):
            x0, x1 = ranges[self.rank]
            out, res = [], []
            for row in np.asarray(rows).reshape(-1, 10):
                plane = ((row[1] >> 32) & 0xffffffff) // nyz
                assert x0 <= plane < x1, 'a walker was handed to a rank that does not own its plane'
                self.continued += 1
                if row[9] > 0:                                             # walks on into the next slab
                    nxt = row.copy()
                    nxt[9] -= 1
                    nxt[1] = (x1 - 1) * nyz | (((x1 % shape[0]) * nyz) << 32)
                    out.append(nxt)
                else:                                                      # ends here: label 7 for everybody
                    res.append((row[0] & 0xffffffff) | (7 << 32))
            return (np.array(out, np.int64).reshape(-1, 10), np.array(res, np.int64))

        def walkers_apply(self, res):
            self.rounds += 1
            for r in np.asarray(res).reshape(-1):
                v = int(r & 0xffffffff)
                assert owner_of(v) == self.rank, 'a result pair reached a rank that does not own the voxel'
                self.applied.append(v)
            return len(res), 0

    backends = [Backend(r) for r in range(n)]
    got = [None] * n

    def work(rank):
        runner = slab.SlabRunner.__new__(slab.SlabRunner)
        runner.be, runner.comm, runner.shape = backends[rank], Comm(rank), shape
        runner.x_range = ranges[rank]
        got[rank] = runner._migrate_walkers(3)

    ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for r, be in enumerate(backends):
        x0 = ranges[r][0]
        assert sorted(be.applied) == [x0 * nyz + k for k in range(3)]      # each of its walkers came back exactly once
        assert got[r] == (3, 0)                                            # (changed here, still parked here)
    assert sum(be.continued for be in backends) == n * (1 + 2 + 3)         # a walker is carried on once per hop + once to end


@pytest.mark.parametrize('counts', [(0, 0, 0, 0), (1, 0, 3, 2), (32, 32, 1, 0), (0, 33, 0, 5), (1000, 7, 0, 250)])
def test_rccl_gather_rows_padding_logic_with_a_mock_context(counts):
    """RcclComm.gather_rows (the only per-step gather on a multi-GPU node) against a context whose comm_allgather is played
    by threads: the one-collective path for short contributions, the second collective for long ones, ragged and empty
    ranks, rank order of the result."""
    import threading
    from pybader_amd import comm as pcomm

    n, w = len(counts), 10
    barrier = threading.Barrier(n)
    slots = [None] * n
    calls = [0] * n

    class Ctx:
        def __init__(self, rank):
            self.rank = rank

        def comm_allgather(self, vals):
            slots[self.rank] = np.ascontiguousarray(vals, np.int64).reshape(-1).copy()
            calls[self.rank] += 1
            barrier.wait()
            assert len({a.size for a in slots}) == 1, 'ncclAllGather needs equal contributions'
            out = np.concatenate(slots)
            barrier.wait()
            return out

    rows = [np.arange(c * w, dtype=np.int64).reshape(c, w) + 1000000 * (r + 1) for r, c in enumerate(counts)]
    got = [None] * n

    def work(rank):
        rc = pcomm.RcclComm.__new__(pcomm.RcclComm)
        rc.ctx, rc.size, rc.rank, rc.device = Ctx(rank), n, rank, True
        got[rank] = rc.gather_rows(rows[rank])

    ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    want = np.concatenate(rows)
    for r in range(n):
        assert got[r].shape == want.shape and np.array_equal(got[r], want)
    assert len(set(calls)) == 1 and calls[0] == (1 if max(counts) <= pcomm.RcclComm.FAST_ROWS else 2)


@pytest.mark.parametrize('broken_rank', [None, 1])
def test_rccl_comm_start_up_is_unanimous_with_a_mock_context(broken_rank):
    """RcclComm.__init__ over the product's own SocketStore (threads as ranks) with a mock context: the unique id travels
    from rank 0, the collective self-test decides the transport, and ONE rank whose communicator fails takes every rank to
    the host-staged transport."""
    import threading
    import uuid
    from pybader_amd import _lib, comm as pcomm

    n = 3
    key = 'test_' + uuid.uuid4().hex
    barrier = threading.Barrier(n)
    slots = [None] * n
    seen_uid = [None] * n

    class Ctx:
        def __init__(self, rank):
            self.rank = rank

        def comm_unique_id(self):
            return bytes(range(128))

        def comm_init(self, rank, size, uid):
            seen_uid[rank] = bytes(uid)
            if rank == broken_rank:
                raise _lib.BaderHipError('mock: no communicator on this rank')

        def _gather(self, vals):
            slots[self.rank] = np.ascontiguousarray(vals, np.int64).reshape(-1).copy()
            barrier.wait()
            out = [a.copy() for a in slots]
            barrier.wait()
            return out

        def comm_allreduce(self, vals, op='sum'):
            return np.sum(self._gather(vals), axis=0).tolist()

        def comm_allgather(self, vals):
            return np.concatenate(self._gather(vals))

    out = [None] * n
    errors = []

    def work(rank):
        try:
            store = pcomm.SocketStore(rank, n, key=key, timeout=60.0)
            ctx = Ctx(rank)
            rc = pcomm.RcclComm(ctx, store)
            out[rank] = (rc.device, rc.transport, rc.sum(rank + 1, 10))
            store.barrier()
            store.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
            barrier.abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert all(u == bytes(range(128)) for u in seen_uid)
    if broken_rank is None:
        assert out == [(True, 'rccl', [6, 30])] * n
    else:   # nobody entered a device collective (the mock's would have waited for the broken rank for ever); sums go over the store
        assert out == [(False, 'host-staged-tcp', [6, 30])] * n


@pytest.mark.parametrize('n,nx,halo', [(2, 64, 8), (2, 64, 24), (3, 96, 8), (4, 40, 3), (8, 512, 16), (8, 64, 2), (5, 100, 16)])
def test_halo_plan_pairs_sends_with_receives_in_order(n, nx, halo):
    """ncclSend / ncclRecv between two ranks are matched in posting order: what rank r sends to p must be, run by run and
    in the same order, what p expects from r (two slabs exchange BOTH halos with the same peer)."""
    ranges = slab.slab_ranges(nx, n)
    plans = [slab.halo_plan(ranges, r, halo, nx) for r in range(n)]
    for r in range(n):
        sends, recvs = plans[r]
        assert all(p != r for p, _, _ in sends + recvs)
        for p in range(n):
            if p != r:
                assert [(a, b) for q, a, b in sends if q == p] == [(a, b) for q, a, b in plans[p][1] if q == r]
        got = sorted(x for _, a, b in recvs for x in range(a, b))
        a0, b0 = ranges[r]
        want = sorted(set((a0 - halo + k) % nx for k in range(halo)) | set((b0 + k) % nx for k in range(halo))) \
            if (b0 - a0) + 2 * halo < nx else [x for x in range(nx) if not a0 <= x < b0]
        assert got == [x for x in want if not a0 <= x < b0]


def test_socket_store_route_delivers_point_to_point():
    """SocketStore.route (the host-staged plane exchange): every rank addresses objects to some ranks, rank 0 forwards; each
    rank gets exactly what was addressed to it, keyed by source."""
    import threading
    import uuid
    from pybader_amd import comm as pcomm

    n = 4
    key = 'route_' + uuid.uuid4().hex
    got, errors = [None] * n, []

    def work(rank):
        try:
            store = pcomm.SocketStore(rank, n, key=key, timeout=60.0)
            out = {dst: np.full(3, 10 * rank + dst) for dst in range(n) if dst != rank and (rank + dst) % 2 == 1}
            got[rank] = store.route(out)
            assert store.route({}) == {}
            store.barrier()
            store.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for r in range(n):
        want = {src: 10 * src + r for src in range(n) if src != r and (src + r) % 2 == 1}
        assert sorted(got[r]) == sorted(want)
        assert all(np.array_equal(got[r][src], np.full(3, v)) for src, v in want.items())


def test_wire_codec_roundtrip_and_rejects_garbage():
    """nothing received over a socket is unpickled -- the store's frames are a tagged binary encoding of plain
    values and numpy arrays of a few dtypes; anything else does not encode, and malformed frames do not decode."""
    import struct
    from pybader_amd import comm as pcomm
    obj = {'a': [1, 2.5, None, True, (3, 'x', b'yy')], 3: np.arange(6, dtype=np.int32).reshape(2, 3), (1, 2): {0: np.zeros(0, np.int8)},
           'rows': np.arange(20, dtype=np.int64).reshape(-1, 10)}
    back = pcomm.decode(pcomm.encode(obj))
    assert back['a'] == obj['a'] and np.array_equal(back[3], obj[3]) and back[3].dtype == np.int32
    assert back[(1, 2)][0].shape == (0,) and np.array_equal(back['rows'], obj['rows'])
    good = pcomm.encode(obj)
    for bad in (b'', b'Z', b'i\x00', good + b'x', good[:-3], b'l' + struct.pack('<Q', 1 << 60),
                b'a\x03<f8\x01' + struct.pack('<q', 1 << 40), b'a\x03<c8\x01' + struct.pack('<q', 0), b's' + struct.pack('<Q', 99) + b'abc'):
        with pytest.raises(ValueError):      # (also a frame that ends inside a fixed-size field: struct.error is re-raised)
            pcomm.decode(bad)
    # a 0-d array keeps its shape; a crafted shape whose int64 product wraps is refused, not sliced
    z = pcomm.decode(pcomm.encode(np.array(5.0)))
    assert z.shape == () and float(z) == 5.0
    wrap = b'a\x03<i8\x02' + struct.pack('<2q', 1 << 62, 4)          # 2^64 elements: wraps to 0 in int64
    with pytest.raises(ValueError):
        pcomm.decode(wrap)
    for unsendable in (object(), {1: {2, 3}}, np.zeros(2, np.complex128), np.array(['a'], dtype=object)):
        with pytest.raises(TypeError):
            pcomm.encode(unsendable)


def test_store_authenticates_before_it_reads_anything_else(tmp_path):
    """A local stranger that finds rank 0's port is dropped on its (wrong) fixed-size hello: the pickle it sends along is
    never looked at, a giant length prefix allocates nothing, and the real rank 1 still gets in afterwards.  The rendezvous
    file is a 0600 file in a 0700 directory of this user."""
    import os
    import pickle
    import socket
    import stat
    import struct
    import threading
    from pybader_amd import comm as pcomm
    key = f'test_auth_{os.getpid()}'
    marker = tmp_path / 'pwned'
    out = {}

    def rank0():
        st = pcomm.SocketStore(0, 2, key=key, timeout=60.0)
        out['gathered'] = st.allgather({'rank': 0})
        st.close()

    t = threading.Thread(target=rank0)
    t.start()
    d = pcomm._rendezvous_dir()
    assert stat.S_IMODE(os.lstat(d).st_mode) == 0o700
    path = os.path.join(d, f'rdzv_{key}')
    for _ in range(200):
        if os.path.exists(path):
            break
        time.sleep(0.05)
    assert stat.S_IMODE(os.lstat(path).st_mode) == 0o600
    port = int(open(path).read().split()[0])

    class Bomb:
        def __reduce__(self):
            return (open, (str(marker), 'w'))

    for payload in (struct.pack('<I', 1) + b'0' * 32 + pickle.dumps(Bomb()),            # wrong token, pickle behind it
                    struct.pack('<Q', 1 << 62) + pickle.dumps({'rank': 1, 'token': 'x'}),   # round 2's frame layout, absurd length
                    b'short'):
        s = socket.create_connection(('127.0.0.1', port), timeout=5.0)
        try:
            s.sendall(payload)
            s.shutdown(socket.SHUT_WR)
            assert s.recv(64) == b''      # dropped without an answer
        except (ConnectionResetError, BrokenPipeError, OSError) as err:
            # closed with the rest of the stranger's bytes unread: the reset may arrive before the send / shutdown returns
            assert not isinstance(err, socket.timeout), 'the stranger was kept waiting'
        s.close()
    assert not marker.exists()
    st1 = pcomm.SocketStore(1, 2, key=key, timeout=60.0)
    got = st1.allgather({'rank': 1})
    st1.close()
    t.join()
    assert got == [{'rank': 0}, {'rank': 1}] == out['gathered']
    assert not marker.exists() and not os.path.exists(path)


def test_comm_watchdog_ends_the_process_when_a_collective_never_returns():
    """A device collective a peer never joins cannot be cancelled: the watchdog ends the PROCESS with a message
    and a non-zero code, so the launcher sees a failure instead of a hang."""
    import subprocess
    import sys
    code = ("import time, sys; sys.path.insert(0, %r)\n"
            "from pybader_amd.comm import Watchdog\n"
            "w = Watchdog(0.4)\n"
            "with w('ncclAllReduce (test)'):\n"
            "    time.sleep(0.05)\n"          # a collective that completes: nothing happens
            "with w('ncclCommInitRank (test)'):\n"
            "    time.sleep(30)\n") % ROOT
    r = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=25, text=True)
    assert r.returncode == 86 and 'ncclCommInitRank (test)' in r.stderr and 'did not complete' in r.stderr


def test_block_collectives_over_the_host_transport():
    """The exchange blocks of the device-driven slab step (csrc/slab_step.h) over the product's host-staged transport, with a
    mock backend whose blocks are numpy arrays: every rank's part reaches every rank, parts may differ in size, and block
    5's counters are summed into its second half."""
    import threading
    import uuid
    from pybader_amd import comm as pcomm

    n = 3
    key = 'blocks_' + uuid.uuid4().hex
    parts = [(0, 40), (40, 8), (48, 100)]          # (offset, bytes) per rank
    out, errors = [None] * n, []

    class Backend:
        def __init__(self, rank):
            self.blocks = {2: np.zeros(148, np.uint8), 5: np.zeros(128, np.uint8)}
            off, m = parts[rank]
            self.blocks[2][off:off + m] = rank + 1
            self.blocks[5][:64] = np.frombuffer((np.arange(8, dtype=np.int64) * (rank + 1)).tobytes(), np.uint8)

        def slab_block_copy(self, which, host, off, to_device):
            if to_device:
                self.blocks[which][off:off + host.size] = host
            else:
                host[:] = self.blocks[which][off:off + host.size]

    def work(rank):
        try:
            store = pcomm.SocketStore(rank, n, key=key, timeout=60.0)
            hc = pcomm.HostComm(store)
            be = Backend(rank)
            hc.allgather_block(be, 2, parts)
            hc.allreduce_block(be)
            out[rank] = (be.blocks[2].copy(), np.frombuffer(be.blocks[5].tobytes(), np.int64).copy())
            store.barrier()
            store.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    want = np.concatenate([np.full(m, r + 1, np.uint8) for r, (_, m) in enumerate(parts)])
    for r in range(n):
        assert np.array_equal(out[r][0], want)
        assert np.array_equal(out[r][1][:8], np.arange(8) * (r + 1)) and np.array_equal(out[r][1][8:], np.arange(8) * 6)


def test_device_driven_step_scheduling_with_a_mock_backend():
    """The scheduler's side of the device-driven step: call order (masks, blocks 0-3, trace, block 4, finish), the repeat
    when the region growth asks for its long schedule (status 1), the hand-over to the host-driven calls (status 2), the
    walker rounds (block 6 / 7 alternately, results from the second gather on) and how the next pass's rounds follow the
    rounds that carried walkers."""
    calls = []

    class Comm:
        rank, size = 0, 2

        def allgather(self, obj):
            return [obj, obj]

        def sum(self, *v):
            return [2 * int(x) for x in v]

        def allgather_block(self, be, which, parts):
            calls.append(('gather', which, tuple(parts[0])))

        def allreduce_block(self, be):
            calls.append(('reduce',))

        def exchange_planes(self, be, which, sends, recvs):
            calls.append(('planes', which))

        stream_ordered = True

    class Backend:
        finishes = [(0, 1), (8, 0)]
        counts = None

        def set_grid(self, *a):
            pass

        def set_table_window(self, m):
            calls.append(('window', m))

        def set_option(self, k, v):
            calls.append(('option', k, v))

        def slab_supported(self, n):
            return True

        def slab_block(self, which):
            return 0, 1000, 100 * which, 10

        def slab_assign_masks(self, rank, n):
            calls.append(('masks', rank, n))

        def slab_assign_trace(self):
            calls.append(('trace',))

        def slab_assign_finish(self):
            calls.append(('finish',))
            return self.finishes.pop(0)

        def maxima(self):
            return np.zeros((8, 3), np.int64)

        def slab_refine_pass(self):
            calls.append(('pass',))

        send = 0

        def slab_walk_send(self, n):     # (round 5: how many walkers of a part travel; the layout follows)
            calls.append(('send', int(n)))
            self.send = int(n)

        def slab_walk_layout(self):
            return [1000, 500, 100, 600, 400] if not self.send else [1000, 16 + 4 * self.send // 100, 100, 600, 400]

        def slab_walkers_round(self, src, last):
            calls.append(('round', src, bool(last)))

        def slab_refine_counts(self):
            return self.counts

    be = Backend()
    runner = slab.SlabRunner(be, Comm(), (64, 64, 64), np.zeros((3, 3, 3)), np.eye(3), halo=8)
    assert runner.enable_table_window() and calls == [('window', 8)]
    del calls[:]
    assert runner.assign('neargrid') == 8 and runner.n_device_steps == 1
    one = [('masks', 0, 2), ('gather', 0, (0, 10)), ('gather', 1, (100, 10)), ('gather', 2, (200, 10)), ('gather', 3, (300, 10)), ('trace',),
           ('gather', 4, (400, 10)), ('finish',)]
    assert calls == [('option', 24, 1)] + one + one
    # a pass whose walkers were carried on in rounds 0 and 1 (bytes of glo[7]), none left: three rounds now, two next time
    del calls[:]
    be.counts = (np.array([50, 0, 7, 0, 0, 0, 0, 0x0101]), np.array([100, 0, 14, 0, 0, 0, 0, 0x0201]))
    assert runner.refine('changed', 2) == [(100, 0), (0, 0)]
    rounds = [c for c in calls if c[0] in ('gather', 'round')]
    assert rounds == [('gather', 6, (0, 500)), ('round', 0, False), ('gather', 7, (0, 100)), ('gather', 7, (600, 400)), ('round', 1, False),
                      ('gather', 6, (0, 100)), ('gather', 6, (600, 400)), ('round', 0, False), ('gather', 7, (0, 100)), ('gather', 7, (600, 400)),
                      ('round', 1, True)]
    assert calls[0] == ('planes', 0) and calls[1] == ('pass',) and ('reduce',) in calls and runner._walker_rounds == 2
    # ... and what travels of a part in the NEXT pass follows this pass's summed export count (14 walkers on 2 ranks: the floor of
    # 2048), set on the backend before the layout is asked for again; a pass that lost walkers goes back to the capacity (0)
    assert [c for c in calls if c[0] == 'send'] == [('send', 0)] and runner._walk_send == 2048   # (no history yet: the capacity)
    del calls[:]
    be.counts = (np.array([50, 0, 4000, 0, 0, 0, 0, 0x01]), np.array([100, 0, 9000, 0, 0, 3, 0, 0x01]))
    runner._finish_escaped = lambda n: 0
    assert runner.refine('changed', 2)[0] == (100, 0)
    sends = [c for c in calls if c[0] == 'send']
    first_gather = next(c for c in calls if c[0] == 'gather')
    assert sends == [('send', 2048)] and first_gather == ('gather', 6, (0, 16 + 4 * 2048 // 100))
    assert runner._walk_send == 0      # glo[5] = 3 walkers lost or stuck: the capacity again
    # status 2: this density goes to the host-driven calls, for good
    be.finishes = [(0, 2)]
    assert runner._assign_device_step() is None


def test_label_wire_is_agreed_after_a_one_sided_scatter():
    """in the path-query fallback only the ranks with parked retraces scatter voxels.  The width a label halo
    travels in must still be the same on every rank before the next exchange (send / receive sizes): SlabRunner takes the
    maximum over the ranks after the queries.  Two ranks (threads), a mock backend: rank 1 has one parked retrace and its
    write widens its width (a label that did not fit), rank 0 has none and returns before the scatter."""
    import threading

    shape, n = (8, 2, 2), 2
    nyz = shape[1] * shape[2]
    ranges = slab.slab_ranges(shape[0], n)
    barrier = threading.Barrier(n)
    slots = [None] * n

    class Comm:
        def __init__(self, rank):
            self.rank, self.size = rank, n

        def allgather(self, obj):
            slots[self.rank] = obj
            barrier.wait()
            out = list(slots)
            barrier.wait()
            return out

        def sum(self, *vals):
            got = self.allgather([int(v) for v in vals])
            return [sum(g[i] for g in got) for i in range(len(vals))]

    class Backend:
        def __init__(self, rank):
            self.rank, self.wire, self.scattered = rank, 1, []

        def escaped_paths(self, max_len):
            if self.rank == 0:
                return np.zeros(0, np.int64), np.zeros(1, np.int64), np.zeros(0, np.int64), np.zeros(0, bool)
            # one path: starts on plane 4 (rank 1's), its second voxel lies on plane 1 (rank 0's) and is known == 2
            return (np.array([4 * nyz], np.int64), np.array([0, 2], np.int64), np.array([4 * nyz, 1 * nyz], np.int64),
                    np.array([False]))

        def gather_voxels(self, idx):
            return np.full(len(idx), 300, np.int32), np.full(len(idx), 2, np.int8)

        def scatter_voxels(self, idx, lab, kn):
            self.scattered.append((np.asarray(idx).tolist(), np.asarray(lab).tolist()))
            self.wire = max(self.wire, 2)        # what the library does when a written label does not fit the width

        def label_wire(self, widen_to=0):
            self.wire = max(self.wire, widen_to)
            return self.wire

    backends = [Backend(r) for r in range(n)]
    moved = [None] * n

    def work(rank):
        runner = slab.SlabRunner.__new__(slab.SlabRunner)
        runner.be, runner.comm, runner.shape, runner.halo = backends[rank], Comm(rank), shape, 1
        runner.ranges = ranges
        moved[rank] = runner._resolve_escaped()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert backends[0].scattered == [] and len(backends[1].scattered) == 1       # only one rank wrote
    assert [be.wire for be in backends] == [2, 2]                                 # ... and both agree on the width
