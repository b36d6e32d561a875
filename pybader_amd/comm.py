"""Transport of the slab scheduler (pybader_amd/slab.py), one process per GPU -- no PyTorch.

* `SocketStore`: the host side.  Ranks of one node find each other through a rendezvous file keyed by
  MASTER_PORT and the launcher's pid (torch.distributed.run keeps MASTER_PORT for its own store, so the
  port itself is not ours to bind), then talk to rank 0 over TCP on 127.0.0.1: all-gather of small pickled
  objects (maxima tables, seeds, path queries), barrier.
* `RcclComm`: the device side through the C ABI (`xb_comm_*`, csrc/comm.h): RCCL send/recv of halo planes
  over xGMI, all-reduce of the iteration counters, broadcast of the brick masks.  A start-up self-test
  (collectives, then a ring of real planes) decides -- unanimously -- between 'rccl' and 'host-staged-tcp'
  (planes staged through host memory and the store: slower, same result).

The reference has no counterpart: its thread blocks share one address space (thread_handlers.py:28-58)."""
import os
import pickle
import secrets
import socket
import struct
import tempfile
import time

import numpy as np

from . import _lib


def _send_msg(sock, payload):
    sock.sendall(struct.pack('<Q', len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray(n)
    view, got = memoryview(buf), 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError('peer closed the connection')
        got += k
    return bytes(buf)


def _recv_msg(sock):
    (n,) = struct.unpack('<Q', _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class SocketStore:
    """Host-side collectives of `size` ranks on one node, star through rank 0."""

    def __init__(self, rank=None, size=None, key=None, timeout=300.0):
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else int(rank)
        self.size = int(os.environ.get('WORLD_SIZE', '1')) if size is None else int(size)
        self.timeout = timeout
        self.peers, self.sock, self._path = [], None, None
        if self.size == 1:
            return
        if key is None:
            key = f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', '')}_{os.getppid()}"
        path = os.path.join(tempfile.gettempdir(), f'pybader_amd_rdzv_{key}')
        deadline = time.monotonic() + timeout
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(('127.0.0.1', 0))
            srv.listen(self.size)
            srv.settimeout(timeout)
            token = secrets.token_hex(16)
            tmp = f'{path}.{os.getpid()}.tmp'
            with open(tmp, 'w') as f:
                f.write(f'{srv.getsockname()[1]} {token}\n')
            os.replace(tmp, path)            # atomic: a reader sees the old file or the new one
            self._path = path
            slots = [None] * self.size
            while any(s is None for s in slots[1:]):
                conn, _ = srv.accept()
                conn.settimeout(timeout)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                hello = pickle.loads(_recv_msg(conn))
                if hello.get('token') != token or not (0 < hello.get('rank', -1) < self.size):
                    conn.close()             # a stranger, or a rank of an earlier run that read a stale file
                    continue
                slots[hello['rank']] = conn
                _send_msg(conn, b'ok')
            srv.close()
            self.peers = slots
        else:
            while True:                      # the file may be missing or stale (an earlier run): retry until rank 0 answers
                try:
                    with open(path) as f:
                        port, token = f.read().split()
                    s = socket.create_connection(('127.0.0.1', int(port)), timeout=5.0)
                    s.settimeout(timeout)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    _send_msg(s, pickle.dumps({'rank': self.rank, 'token': token}))
                    if _recv_msg(s) == b'ok':
                        self.sock = s
                        break
                    s.close()
                except (OSError, ValueError, ConnectionError):
                    pass
                if time.monotonic() > deadline:
                    raise TimeoutError(f'rank {self.rank}: no rendezvous with rank 0 through {path}')
                time.sleep(0.05)

    def allgather(self, obj):
        if self.size == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [pickle.loads(_recv_msg(c)) for c in self.peers[1:]]
            blob = pickle.dumps(out, protocol=pickle.HIGHEST_PROTOCOL)
            for c in self.peers[1:]:
                _send_msg(c, blob)
            return out
        _send_msg(self.sock, pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL))
        return pickle.loads(_recv_msg(self.sock))

    def barrier(self):
        self.allgather(None)

    def route(self, out):
        """point-to-point through rank 0: `out` maps a destination rank to an object; returns {source rank: object} of
        what the others addressed to this rank (every rank calls it, possibly with an empty dict)"""
        if self.size == 1:
            return {}
        if self.rank == 0:
            boxes = [dict() for _ in range(self.size)]
            for dst, obj in out.items():
                boxes[dst][0] = obj
            for src in range(1, self.size):
                for dst, obj in pickle.loads(_recv_msg(self.peers[src])).items():
                    boxes[dst][src] = obj
            for dst in range(1, self.size):
                _send_msg(self.peers[dst], pickle.dumps(boxes[dst], protocol=pickle.HIGHEST_PROTOCOL))
            return boxes[0]
        _send_msg(self.sock, pickle.dumps(out, protocol=pickle.HIGHEST_PROTOCOL))
        return pickle.loads(_recv_msg(self.sock))

    def close(self):
        for c in self.peers[1:] if self.peers else []:
            c.close()
        if self.sock is not None:
            self.sock.close()
        if self._path:
            try:
                os.unlink(self._path)
            except OSError:
                pass
        self.peers, self.sock, self._path = [], None, None


class _stdout_to_stderr:
    """librccl greets on stdout when a communicator is made; bench.py owes its caller exactly one JSON line there"""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


class HostComm:
    """The scheduler's collectives on host objects only (the CPU tests drive a host backend with it; RcclComm
    builds on it).  `backend.planes(which)` must then be a writable (nx, plane) numpy view."""

    def __init__(self, store):
        self.store = store
        self.rank, self.size = store.rank, store.size
        self.transport = 'tcp-host'

    def allgather(self, obj):
        return self.store.allgather(obj)

    def sum(self, *vals):
        got = self.allgather([int(v) for v in vals])
        return [sum(g[i] for g in got) for i in range(len(vals))]

    def max_float(self, x):
        return max(self.allgather(float(x)))

    def barrier(self):
        self.store.barrier()

    def gather_rows(self, rows):
        """every rank's (n_r, w) int64 rows, concatenated in rank order"""
        rows = np.ascontiguousarray(rows, np.int64)
        return np.concatenate(self.allgather(rows)) if self.size > 1 else rows

    def _staged(self, sends, recvs, fetch, store_back):
        """planes through the store: every rank hands rank 0 what it sends, addressed; rank 0 forwards each rank its share"""
        out = {}
        for peer, xa, xb in sends:
            out.setdefault(peer, {})[(xa, xb)] = fetch(xa, xb)
        if hasattr(self.store, 'route'):
            inbox = self.store.route(out)
        else:       # (a store without routing: everybody gets everything)
            everything = self.allgather(out)
            inbox = {src: everything[src].get(self.rank, {}) for src in range(self.size)}
        for peer, xa, xb in recvs:
            store_back(xa, xb, inbox[peer][(xa, xb)])

    def exchange_planes(self, backend, which, sends, recvs):
        arr = backend.planes(which)
        self._staged(sends, recvs, lambda a, b: arr[a:b].copy(), lambda a, b, v: arr.__setitem__(slice(a, b), v))

    def share_brick_masks(self, backend, chunks):
        raise NotImplementedError('host backends build the whole table themselves')


class RcclComm(HostComm):
    """Device transport through libbader_hip.so's xb_comm_* (RCCL), host objects through the store."""

    def __init__(self, ctx, store):
        super().__init__(store)
        self.ctx = ctx
        self.device = False
        self.transport = 'none' if self.size == 1 else 'host-staged-tcp'
        if self.size == 1:
            return
        with _stdout_to_stderr():
            uid = ctx.comm_unique_id() if self.rank == 0 else None
        uid = self.allgather(uid)[0]
        ok = True
        try:
            with _stdout_to_stderr():
                ctx.comm_init(self.rank, self.size, uid)
        except _lib.BaderHipError as err:
            ok = False
            self.init_error = str(err)
        # first vote (over the store): a rank without a communicator must keep the others out of the device collectives,
        # which would wait for it for ever
        if all(self.allgather(bool(ok))):
            try:
                n = self.size
                ok = ctx.comm_allreduce([self.rank + 1, 1]) == [n * (n + 1) // 2, n]
                ok = ok and ctx.comm_allgather([self.rank, 7 * self.rank]).reshape(n, 2).tolist() == [[r, 7 * r] for r in range(n)]
            except _lib.BaderHipError as err:
                ok = False
                self.init_error = str(err)
        else:
            ok = False
        self.device = all(self.allgather(bool(ok)))     # unanimous: every rank takes the same transport
        if self.device:
            self.transport = 'rccl'

    def selftest_planes(self, backend, ranges):
        """second stage of the self-test, on the arrays that will really travel: every rank's first owned label
        plane goes round the ring; any wrong payload on any rank switches all ranks to the staged transport.
        Called once, right after the grid was set (the labels' contents do not matter yet)."""
        if not self.device:
            return
        ctx, nxt, prv = self.ctx, (self.rank + 1) % self.size, (self.rank - 1) % self.size
        mine, theirs = ranges[self.rank][0], ranges[prv][0]
        plane = np.full(ctx.plane_elems(), self.rank + 1, np.int32)
        ok = True
        try:
            ctx.copy_planes(0, True, plane, mine, mine + 1)
            ctx.comm_exchange_planes(0, [(nxt, mine, mine + 1)], [(prv, theirs, theirs + 1)])
            got = np.empty_like(plane)
            ctx.copy_planes(0, False, got, theirs, theirs + 1)
            ok = bool((got == prv + 1).all())
        except _lib.BaderHipError:
            ok = False
        if not all(self.allgather(ok)):
            self.device = False
            self.transport = 'host-staged-tcp'

    def sum(self, *vals):
        if self.device:
            return self.ctx.comm_allreduce([int(v) for v in vals])
        return super().sum(*vals)

    FAST_ROWS = 32     # contributions up to this many rows travel in ONE collective (count + rows, padded)

    def gather_rows(self, rows):
        rows = np.ascontiguousarray(rows, np.int64)
        if not self.device:
            return super().gather_rows(rows)
        n, w = rows.shape
        first = np.zeros((self.FAST_ROWS + 1, w), np.int64)
        first[0, 0] = n
        first[1:1 + min(n, self.FAST_ROWS)] = rows[:self.FAST_ROWS]
        got = self.ctx.comm_allgather(first.reshape(-1)).reshape(self.size, self.FAST_ROWS + 1, w)
        counts = got[:, 0, 0]
        cap = int(counts.max())
        if cap <= self.FAST_ROWS:       # the tie flags, the maxima rows, the last walker rounds
            return np.concatenate([got[r, 1:1 + int(counts[r])] for r in range(self.size)])
        padded = np.zeros((cap, w), np.int64)
        padded[:n] = rows
        got = self.ctx.comm_allgather(padded.reshape(-1)).reshape(self.size, cap, w)
        return np.concatenate([got[r, :int(counts[r])] for r in range(self.size)])

    def exchange_planes(self, backend, which, sends, recvs):
        if not sends and not recvs:
            return
        ctx = self.ctx
        if self.device:
            ctx.comm_exchange_planes(which, sends, recvs)
            return
        dt = np.int32 if which == 0 else np.int8
        pe = ctx.plane_elems()

        def fetch(a, b):
            buf = np.empty((b - a) * pe, dt)
            ctx.copy_planes(which, False, buf, a, b)
            return buf

        self._staged(sends, recvs, fetch, lambda a, b, v: ctx.copy_planes(which, True, np.ascontiguousarray(v, dt), a, b))

    def share_brick_masks(self, backend, chunks):
        ctx = self.ctx
        if self.device:
            ctx.comm_share_brick_masks([c[0] for c in chunks], [c[1] for c in chunks])
            return
        first, count = chunks[self.rank]
        parts = self.allgather(ctx.brick_masks_copy(None, first, count))
        for r, (f, n) in enumerate(chunks):
            if r != self.rank and n:
                ctx.brick_masks_copy(parts[r], f, n)

    def close(self):
        self.store.close()
