"""Transport of the slab scheduler (pybader_amd/slab.py), one process per GPU -- no PyTorch.

* `SocketStore`: the host side.  Ranks of one node find each other through a rendezvous file keyed by
  MASTER_PORT and the launcher's pid (torch.distributed.run keeps MASTER_PORT for its own store, so the
  port itself is not ours to bind) -- a 0600 file in a 0700 per-user directory -- authenticate with a raw fixed-size
  token frame (both ways, hmac), then talk to rank 0 over TCP on 127.0.0.1: all-gather of small objects (maxima tables,
  seeds, path queries) in a tagged binary encoding (nothing received is ever unpickled), barrier.
* `RcclComm`: the device side through the C ABI (`xb_comm_*`, csrc/comm.h): RCCL send/recv of halo planes
  over xGMI, all-reduce of the iteration counters, broadcast of the brick masks.  A start-up self-test
  (collectives, then a ring of real planes) decides -- unanimously -- between 'rccl' and 'host-staged-tcp'
  (planes staged through host memory and the store: slower, same result).

The reference has no counterpart: its thread blocks share one address space (thread_handlers.py:28-58)."""
import hmac
import os
import secrets
import socket
import stat
import struct
import tempfile
import threading
import time

import numpy as np

from . import _lib

# ---- wire format ---------------------------------------------------------------------------------------------------
# Nothing that arrives over a socket is ever unpickled (rank 0 used to pickle.loads the hello frame of any
# local process that found its port).  The scheduler's messages are small trees of None / bool / int / float / str /
# bytes / list / tuple / dict and numpy arrays of a few plain dtypes: encoded with a tagged binary codec that can
# only ever produce those.  Frames are length-prefixed and capped; the hello is a fixed-size raw frame checked with
# hmac.compare_digest before anything else is read from the peer.
MAX_FRAME = 1 << 36            # planes of a 1024^3 halo are < 2^31 bytes; a stranger cannot make us allocate more than this
HELLO_BYTES = 4 + 32           # rank (uint32) + 32 token characters
_DTYPES = {'|i1', '<i2', '<i4', '<i8', '|u1', '<u2', '<u4', '<u8', '<f4', '<f8', '|b1'}


def _enc(obj, out):
    if obj is None:
        out.append(b'N')
    elif isinstance(obj, (bool, np.bool_)):
        out.append(b'T' if obj else b'F')
    elif isinstance(obj, (int, np.integer)):
        out.append(b'i' + struct.pack('<q', int(obj)))
    elif isinstance(obj, (float, np.floating)):
        out.append(b'f' + struct.pack('<d', float(obj)))
    elif isinstance(obj, str):
        raw = obj.encode('utf-8')
        out.append(b's' + struct.pack('<Q', len(raw)) + raw)
    elif isinstance(obj, (bytes, bytearray)):
        out.append(b'b' + struct.pack('<Q', len(obj)) + bytes(obj))
    elif isinstance(obj, np.ndarray):
        a = np.asarray(obj, order='C')     # (ascontiguousarray would turn a 0-d array into shape (1,))
        if a.dtype.str not in _DTYPES:
            raise TypeError(f'comm: dtype {a.dtype} does not travel')
        ds = a.dtype.str.encode()
        out.append(b'a' + struct.pack('<B', len(ds)) + ds + struct.pack('<B', a.ndim) + struct.pack(f'<{a.ndim}q', *a.shape))
        out.append(a.tobytes())
    elif isinstance(obj, (list, tuple)):
        out.append((b'l' if isinstance(obj, list) else b't') + struct.pack('<Q', len(obj)))
        for x in obj:
            _enc(x, out)
    elif isinstance(obj, dict):
        out.append(b'd' + struct.pack('<Q', len(obj)))
        for k, v in obj.items():
            _enc(k, out)
            _enc(v, out)
    else:
        raise TypeError(f'comm: {type(obj).__name__} does not travel')


def encode(obj):
    out = []
    _enc(obj, out)
    return b''.join(out)


def _dec(buf, pos):
    tag = buf[pos:pos + 1]
    pos += 1
    if tag == b'N':
        return None, pos
    if tag == b'T':
        return True, pos
    if tag == b'F':
        return False, pos
    if tag == b'i':
        return struct.unpack_from('<q', buf, pos)[0], pos + 8
    if tag == b'f':
        return struct.unpack_from('<d', buf, pos)[0], pos + 8
    if tag in (b's', b'b'):
        (n,) = struct.unpack_from('<Q', buf, pos)
        pos += 8
        if n > len(buf) - pos:
            raise ValueError('comm: truncated frame')
        raw = bytes(buf[pos:pos + n])
        return (raw.decode('utf-8') if tag == b's' else raw), pos + n
    if tag == b'a':
        (dl,) = struct.unpack_from('<B', buf, pos)
        ds = bytes(buf[pos + 1:pos + 1 + dl]).decode()
        pos += 1 + dl
        if ds not in _DTYPES:
            raise ValueError('comm: unexpected dtype in a frame')
        (nd,) = struct.unpack_from('<B', buf, pos)
        shape = struct.unpack_from(f'<{nd}q', buf, pos + 1)
        pos += 1 + 8 * nd
        if any(d < 0 for d in shape):
            raise ValueError('comm: bad shape in a frame')
        dt = np.dtype(ds)
        count = 1
        for d in shape:                    # Python ints: a crafted shape cannot wrap the product
            count *= int(d)
        n = count * dt.itemsize
        if n < 0 or n > len(buf) - pos:
            raise ValueError('comm: truncated frame')
        return np.frombuffer(buf, dt, count=n // dt.itemsize, offset=pos).reshape(shape).copy(), pos + n
    if tag in (b'l', b't'):
        (n,) = struct.unpack_from('<Q', buf, pos)
        pos += 8
        if n > len(buf) - pos:
            raise ValueError('comm: truncated frame')
        items = []
        for _ in range(n):
            x, pos = _dec(buf, pos)
            items.append(x)
        return (items if tag == b'l' else tuple(items)), pos
    if tag == b'd':
        (n,) = struct.unpack_from('<Q', buf, pos)
        pos += 8
        if n > len(buf) - pos:
            raise ValueError('comm: truncated frame')
        d = {}
        for _ in range(n):
            k, pos = _dec(buf, pos)
            v, pos = _dec(buf, pos)
            d[k] = v
        return d, pos
    raise ValueError('comm: unknown tag in a frame')


def decode(buf):
    try:
        obj, pos = _dec(memoryview(buf), 0)
    except struct.error as e:              # a frame that ends inside a fixed-size field
        raise ValueError(f'comm: truncated frame ({e})') from None
    if pos != len(buf):
        raise ValueError('comm: trailing bytes in a frame')
    return obj


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


def _recv_msg(sock, limit=MAX_FRAME):
    (n,) = struct.unpack('<Q', _recv_exact(sock, 8))
    if n > limit:
        raise ConnectionError(f'frame of {n} bytes exceeds the limit {limit}')
    return _recv_exact(sock, n)


def _rendezvous_dir():
    """A directory only this user can enter (0700, owned by us, not a symlink): the rendezvous file in it names the port
    and the token, so whoever can write there can impersonate rank 0."""
    base = os.environ.get('XDG_RUNTIME_DIR')
    if not (base and os.path.isdir(base) and os.access(base, os.W_OK)):
        base = tempfile.gettempdir()
    path = os.path.join(base, f'pybader_amd-{os.geteuid()}')
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.geteuid() or (st.st_mode & 0o077):
        raise PermissionError(f'{path} is not a private directory of this user: refusing to rendezvous through it')
    return path


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
        key = ''.join(ch if ch.isalnum() or ch in '-_.' else '_' for ch in str(key))
        path = os.path.join(_rendezvous_dir(), f'rdzv_{key}')
        deadline = time.monotonic() + timeout
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(('127.0.0.1', 0))
            srv.listen(self.size)
            srv.settimeout(timeout)
            token = secrets.token_hex(16)
            try:
                os.unlink(path)              # a stale file of an earlier run with the same key
            except FileNotFoundError:
                pass
            # written under a private name (O_EXCL | O_NOFOLLOW, 0600) and moved into place: a reader sees the whole line or
            # no file, never an empty one
            tmp = f'{path}.{os.getpid()}.{secrets.token_hex(4)}.tmp'
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, 'O_NOFOLLOW', 0), 0o600)
            with os.fdopen(fd, 'w') as f:
                f.write(f'{srv.getsockname()[1]} {token}\n')
            os.replace(tmp, path)
            self._path = path
            slots = [None] * self.size
            while any(s is None for s in slots[1:]):
                left = deadline - time.monotonic()     # one deadline for the whole rendezvous, as on the clients' side
                if left <= 0:
                    srv.close()
                    raise TimeoutError(f'rank 0: {sum(s is None for s in slots[1:])} of {self.size - 1} ranks did not join through {path}')
                srv.settimeout(left)
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                conn.settimeout(min(timeout, 2.0))     # (a silent stranger holds the loop for two seconds at most)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                try:                         # fixed-size raw hello, authenticated before anything is decoded
                    hello = _recv_exact(conn, HELLO_BYTES)
                    (r,) = struct.unpack('<I', hello[:4])
                    good = hmac.compare_digest(hello[4:], token.encode()) and 0 < r < self.size and slots[r] is None
                except (OSError, ConnectionError):
                    good = False
                if not good:
                    conn.close()             # a stranger, or a rank of an earlier run that read a stale file
                    continue
                conn.settimeout(timeout)
                slots[r] = conn
                # mutual: the peer checks that the server it found knows the token too
                conn.sendall(hmac.new(token.encode(), b'rank0:' + hello[:4], 'sha256').digest())
            srv.close()
            self.peers = slots
        else:
            while True:                      # the file may be missing or stale (an earlier run): retry until rank 0 answers
                try:
                    fd = os.open(path, os.O_RDONLY | getattr(os, 'O_NOFOLLOW', 0))
                    with os.fdopen(fd) as f:
                        st = os.fstat(f.fileno())
                        if st.st_uid != os.geteuid() or (st.st_mode & 0o077):
                            raise PermissionError(f'{path} is not a private file of this user')
                        port, token = f.read().split()
                    s = socket.create_connection(('127.0.0.1', int(port)), timeout=5.0)
                    s.settimeout(timeout)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    me = struct.pack('<I', self.rank)
                    s.sendall(me + token.encode())
                    if hmac.compare_digest(_recv_exact(s, 32), hmac.new(token.encode(), b'rank0:' + me, 'sha256').digest()):
                        self.sock = s
                        break
                    s.close()
                except PermissionError:
                    raise
                except (OSError, ValueError, ConnectionError):
                    pass
                if time.monotonic() > deadline:
                    raise TimeoutError(f'rank {self.rank}: no rendezvous with rank 0 through {path}')
                time.sleep(0.05)

    def allgather(self, obj):
        if self.size == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [decode(_recv_msg(c)) for c in self.peers[1:]]
            blob = encode(out)
            for c in self.peers[1:]:
                _send_msg(c, blob)
            return out
        _send_msg(self.sock, encode(obj))
        return decode(_recv_msg(self.sock))

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
                for dst, obj in decode(_recv_msg(self.peers[src])).items():
                    if not (isinstance(dst, int) and 0 <= dst < self.size):
                        raise ValueError('comm: bad destination in a routed frame')
                    boxes[dst][src] = obj
            for dst in range(1, self.size):
                _send_msg(self.peers[dst], encode(boxes[dst]))
            return boxes[0]
        _send_msg(self.sock, encode(out))
        return decode(_recv_msg(self.sock))

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


class Watchdog:
    """A device collective that a peer never joins does not return and cannot be cancelled: a rank whose ncclCommInitRank
    failed, or that died, would leave the others waiting for ever.  Every device collective of RcclComm runs
    under this deadline; when it passes the PROCESS exits non-zero with a message, so the launcher ends the job loudly
    instead of hanging it.  ctypes releases the GIL during the library calls, so the timer thread does fire."""

    def __init__(self, seconds=None):
        self.seconds = float(os.environ.get('XB_COMM_TIMEOUT', '180')) if seconds is None else float(seconds)
        self._deadline, self._what = None, ''
        self._lock = threading.Lock()
        self._thread = None

    def _run(self):
        while True:
            time.sleep(0.5)
            with self._lock:
                d, what = self._deadline, self._what
            if d is not None and time.monotonic() > d:
                import sys
                print(f'pybader_amd.comm: device collective `{what}` did not complete within {self.seconds:.0f} s '
                      '(a peer is missing or stuck); exiting', file=sys.stderr, flush=True)
                os._exit(86)

    def __call__(self, what):
        self._what_next = what
        return self

    def __enter__(self):
        if self._thread is None:
            self._thread = threading.Thread(target=self._run, name='xb-comm-watchdog', daemon=True)
            self._thread.start()
        with self._lock:
            self._deadline, self._what = time.monotonic() + self.seconds, getattr(self, '_what_next', '')

    def __exit__(self, *exc):
        with self._lock:
            self._deadline = None
        return False


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

    # ---- exchange blocks of the device-driven slab step (csrc/slab_step.h), staged through the host ----
    def allgather_block(self, backend, which, parts):
        """every rank's part (offset, bytes) of block `which` to every rank"""
        off, n = parts[self.rank]
        mine = np.empty(n, np.uint8)
        backend.slab_block_copy(which, mine, off, False)
        got = self.allgather(mine)
        for r, (o, m) in enumerate(parts):
            if r != self.rank and m:
                backend.slab_block_copy(which, np.ascontiguousarray(got[r], np.uint8), o, True)

    def allreduce_block(self, backend):
        """block 5: summed[0..8) := sum over the ranks of local[0..8) (int64)"""
        mine = np.empty(64, np.uint8)
        backend.slab_block_copy(5, mine, 0, False)
        total = np.sum([np.frombuffer(np.ascontiguousarray(g, np.uint8).tobytes(), np.int64) for g in self.allgather(mine)], axis=0)
        backend.slab_block_copy(5, np.frombuffer(total.astype(np.int64).tobytes(), np.uint8).copy(), 64, True)


class RcclComm(HostComm):
    """Device transport through libbader_hip.so's xb_comm_* (RCCL), host objects through the store."""

    def __init__(self, ctx, store):
        super().__init__(store)
        self.ctx = ctx
        self.device = False
        self.init_error = None        # why the device transport is not in use (bench.py prints it and fails on a full node)
        self.watchdog = Watchdog()
        self.transport = 'none' if self.size == 1 else 'host-staged-tcp'
        if self.size == 1:
            return
        ok = True
        uid = None
        try:
            with _stdout_to_stderr():
                uid = ctx.comm_unique_id() if self.rank == 0 else None
        except _lib.BaderHipError as err:
            self.init_error = str(err)
        uid = self.allgather(uid)[0]
        if uid is None:
            ok = False
            self.init_error = self.init_error or 'rank 0 could not make an RCCL unique id'
        else:
            try:
                with _stdout_to_stderr(), self.watchdog('ncclCommInitRank'):
                    ctx.comm_init(self.rank, self.size, uid)
            except _lib.BaderHipError as err:
                ok = False
                self.init_error = str(err)
        # first vote (over the store): a rank without a communicator must keep the others out of the device collectives,
        # which would wait for it for ever
        if all(self.allgather(bool(ok))):
            try:
                n = self.size
                with self.watchdog('self-test collectives'):
                    ok = ctx.comm_allreduce([self.rank + 1, 1]) == [n * (n + 1) // 2, n]
                    ok = ok and ctx.comm_allgather([self.rank, 7 * self.rank]).reshape(n, 2).tolist() == [[r, 7 * r] for r in range(n)]
                if not ok:
                    self.init_error = 'the RCCL self-test collectives returned wrong values'
            except _lib.BaderHipError as err:
                ok = False
                self.init_error = str(err)
        else:
            ok = False
            self.init_error = self.init_error or 'a peer rank has no RCCL communicator'
        votes = self.allgather(bool(ok))
        self.device = all(votes)     # unanimous: every rank takes the same transport
        if self.device:
            self.transport = 'rccl'
        elif self.init_error is None:
            self.init_error = f'ranks {[r for r, v in enumerate(votes) if not v]} failed the RCCL self-test'

    watchdog = None

    def _guard(self, what):
        import contextlib
        return self.watchdog(what) if self.watchdog is not None else contextlib.nullcontext()

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
            with self.watchdog('self-test plane ring'):
                ctx.comm_exchange_planes(0, [(nxt, mine, mine + 1)], [(prv, theirs, theirs + 1)])
            got = np.empty_like(plane)
            ctx.copy_planes(0, False, got, theirs, theirs + 1)
            ok = bool((got == prv + 1).all())
            if not ok:
                self.init_error = 'the RCCL plane ring delivered a wrong payload'
        except _lib.BaderHipError as err:
            ok = False
            self.init_error = str(err)
        votes = self.allgather(ok)
        if not all(votes):
            self.device = False
            self.transport = 'host-staged-tcp'
            self.init_error = self.init_error or f'ranks {[r for r, v in enumerate(votes) if not v]} failed the RCCL plane ring'

    def sum(self, *vals):
        if self.device:
            with self._guard('allreduce'):
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
        with self._guard('allgather'):
            got = self.ctx.comm_allgather(first.reshape(-1)).reshape(self.size, self.FAST_ROWS + 1, w)
        counts = got[:, 0, 0]
        cap = int(counts.max())
        if cap <= self.FAST_ROWS:       # the tie flags, the maxima rows, the last walker rounds
            return np.concatenate([got[r, 1:1 + int(counts[r])] for r in range(self.size)])
        padded = np.zeros((cap, w), np.int64)
        padded[:n] = rows
        with self._guard('allgather'):
            got = self.ctx.comm_allgather(padded.reshape(-1)).reshape(self.size, cap, w)
        return np.concatenate([got[r, :int(counts[r])] for r in range(self.size)])

    def exchange_planes(self, backend, which, sends, recvs):
        if not sends and not recvs:
            return
        ctx = self.ctx
        if self.device:
            with self._guard('plane exchange'):
                ctx.comm_exchange_planes(which, sends, recvs)
            return
        dt = np.int32 if which == 0 else np.int8
        pe = ctx.plane_elems()

        def fetch(a, b):
            buf = np.empty((b - a) * pe, dt)
            ctx.copy_planes(which, False, buf, a, b)
            return buf

        self._staged(sends, recvs, fetch, lambda a, b, v: ctx.copy_planes(which, True, np.ascontiguousarray(v, dt), a, b))

    @property
    def stream_ordered(self):
        """the device collectives are ordered on the context's stream: nothing has to wait for the card before them"""
        return self.device

    def allgather_block(self, backend, which, parts):
        if not self.device:
            return super().allgather_block(backend, which, parts)
        with self._guard('block allgather'):     # (the first collective on a communicator connects lazily: it can block on the host)
            self.ctx.comm_allgather_block(which, [p[0] for p in parts], [p[1] for p in parts])

    def allreduce_block(self, backend):
        if not self.device:
            return super().allreduce_block(backend)
        with self._guard('block allreduce'):
            self.ctx.comm_allreduce_block()

    def share_brick_masks(self, backend, chunks):
        ctx = self.ctx
        if self.device:
            with self._guard('brick mask broadcast'):
                ctx.comm_share_brick_masks([c[0] for c in chunks], [c[1] for c in chunks])
            return
        first, count = chunks[self.rank]
        parts = self.allgather(ctx.brick_masks_copy(None, first, count))
        for r, (f, n) in enumerate(chunks):
            if r != self.rank and n:
                ctx.brick_masks_copy(parts[r], f, n)

    def close(self):
        self.store.close()
