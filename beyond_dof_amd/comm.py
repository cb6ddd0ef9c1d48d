"""Process group of the reconstruction loops: one process per GPU.

The reference uses mpi4py COMM_WORLD (Allreduce / Barrier on host float64 arrays, cnn_propagator/fullfield.py:124-134,
348-351) and degrades to a single fake rank when MPI is absent (cnn_propagator/pseudo.py:27-33).  Here:

  RcclComm    the product path.  Data plane: RCCL over xGMI on device float32 buffers, behind the C ABI of libbdof.so
              (bdof_comm_*, include/bdof.h) — no torch in the process.  Control plane (rendezvous, barriers, a few
              scalars): a unix-domain socket star on rank 0 (SocketGroup), one node as in BASELINE's 8-GPU configuration.
  TorchComm   torch.distributed, backend "gloo" — multi-rank rehearsals on CPU or with several ranks on one GPU (RCCL
              refuses two ranks on one device); synchronous.
  PseudoComm  size-1 stand-in, the counterpart of cnn_propagator/pseudo.py:Mpi.

Device collectives share one interface, stream-ordered against a bdof ctx:
  t = comm.start_allreduce(ctx, buf, lo, hi)            SUM of the floats [lo, hi) of `buf`, in place
  t = comm.start_reduce_scatter(ctx, buf, lo, per)      SUM; rank r ends with its part [lo + r*per, lo + (r+1)*per)
  t = comm.start_allgather(ctx, buf, lo, per)           every rank's part [lo + r*per, ...) to all
  comm.wait(ctx, t)                                     the ctx stream waits for that collective
"""
import ctypes
import json
import os
import socket
import stat
import struct
import time

import numpy as np


class PseudoComm(object):
    """size-1 stand-in, the counterpart of cnn_propagator/pseudo.py:Mpi."""
    size = 1
    rank = 0
    local_rank = 0
    backend = 'none'
    sharded = False

    def attach(self, ctx):
        pass

    def Barrier(self):
        pass

    def start_allreduce(self, ctx, buf, lo, hi):
        return None

    def start_reduce_scatter(self, ctx, buf, lo, per):
        return None

    def start_allgather(self, ctx, buf, lo, per):
        return None

    def wait(self, ctx, ticket):
        pass

    def allreduce_sum_device(self, ctx, buf):
        return buf

    def bcast_device(self, ctx, buf, root=0):
        return buf

    def allreduce_sum_host(self, arr):
        return arr

    def allreduce_max_host(self, arr):
        return arr

    def bcast_host(self, arr, root=0):
        return arr

    def close(self):
        pass


# ------------------------------------------------------------------------------------------------------------------
# control plane of the native path
# ------------------------------------------------------------------------------------------------------------------
def _send_msg(sock, data):
    sock.sendall(struct.pack('<Q', len(data)))
    sock.sendall(data)


def _recv_exact(sock, n):
    chunks, got = [], 0
    while got < n:
        b = sock.recv(min(n - got, 1 << 22))
        if not b:
            raise ConnectionError('peer closed the rendezvous socket')
        chunks.append(b)
        got += len(b)
    return b''.join(chunks)


def _recv_msg(sock):
    (n,) = struct.unpack('<Q', _recv_exact(sock, 8))
    return _recv_exact(sock, n)


def _encode(obj):
    """Wire form of the few things the control plane carries — None, bytes, numpy arrays, JSON-able scalars / lists / dicts.
    No pickle: a peer's message is parsed, never executed."""
    if obj is None:
        return b'N'
    if isinstance(obj, (bytes, bytearray)):
        return b'B' + bytes(obj)
    if isinstance(obj, np.ndarray):
        a = np.ascontiguousarray(obj)
        if a.dtype.hasobject:
            raise TypeError('object arrays are not sent over the rendezvous socket')
        head = json.dumps({'dtype': a.dtype.str, 'shape': list(a.shape)}).encode()
        return b'A' + struct.pack('<I', len(head)) + head + a.tobytes()
    if isinstance(obj, (np.integer, np.floating)):
        obj = obj.item()
    return b'J' + json.dumps(obj).encode()


def _decode(data):
    tag, body = data[:1], data[1:]
    if tag == b'N':
        return None
    if tag == b'B':
        return bytes(body)
    if tag == b'A':
        (n,) = struct.unpack('<I', body[:4])
        head = json.loads(body[4:4 + n].decode())
        return np.frombuffer(body[4 + n:], dtype=np.dtype(head['dtype'])).reshape(head['shape']).copy()
    if tag == b'J':
        return json.loads(body.decode())
    if tag == b'L':
        items, off = [], 0
        while off < len(body):
            (n,) = struct.unpack('<Q', body[off:off + 8])
            items.append(_decode(body[off + 8:off + 8 + n]))
            off += 8 + n
        return items
    raise ValueError('malformed rendezvous message (tag {!r})'.format(tag))


def _encode_list(items):
    return b'L' + b''.join(struct.pack('<Q', len(e)) + e for e in items)


def rendezvous_dir():
    """A directory only this user can enter (0700, owned by us): $XDG_RUNTIME_DIR when the session has one, else
    /tmp/bdof-<uid>.  A socket in plain /tmp could be pre-created or bound by another local user."""
    base = os.environ.get('XDG_RUNTIME_DIR')
    if base and os.path.isdir(base) and os.stat(base).st_uid == os.getuid():
        d = os.path.join(base, 'bdof')
    else:
        d = '/tmp/bdof-{}'.format(os.getuid())
    try:
        os.makedirs(d, mode=0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError('rendezvous directory {} is not a private directory of uid {} (mode {:o}, owner {})'.format(
            d, os.getuid(), st.st_mode & 0o777, st.st_uid))
    return d


def rendezvous_path():
    """One node, one job per (MASTER_ADDR, MASTER_PORT) — the launcher's own key for the job (torch.distributed.run
    exports both; bench.py's self-launch passes a socket in a directory of its own through BDOF_RDZV)."""
    p = os.environ.get('BDOF_RDZV')
    if p:
        return p
    return os.path.join(rendezvous_dir(), 'rdzv_{}_{}.sock'.format(os.environ.get('MASTER_ADDR', '127.0.0.1'),
                                                                   os.environ.get('MASTER_PORT', '29500')))


def _peer_uid(sock):
    """uid of the process at the other end of a unix-domain socket (SO_PEERCRED)."""
    cred = sock.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize('3i'))
    return struct.unpack('3i', cred)[1]


class SocketGroup(object):
    """Star of unix-domain stream sockets on rank 0: rendezvous + small host-side collectives (None / bytes / numpy arrays /
    JSON-able values, see _encode).  Rank 0 binds the path (a stale file of a dead job is replaced), the others connect with
    retries; both ends check that the peer runs under the same uid (SO_PEERCRED)."""

    def __init__(self, rank, size, path=None, timeout=None):
        self.rank, self.size = int(rank), int(size)
        self.path = path or rendezvous_path()
        timeout = float(timeout if timeout is not None else os.environ.get('BDOF_RDZV_TIMEOUT', '600'))
        self.peers, self.sock = {}, None
        if self.size == 1:
            return
        if self.rank == 0:
            try:
                os.unlink(self.path)
            except OSError:
                pass
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(self.path)
            srv.listen(self.size)
            srv.settimeout(timeout)
            try:
                while len(self.peers) < self.size - 1:
                    conn, _ = srv.accept()
                    conn.settimeout(timeout)
                    peer_uid = _peer_uid(conn)              # read before the socket is closed (getsockopt on a closed one: EBADF)
                    if peer_uid != os.getuid():
                        conn.close()
                        raise RuntimeError('rendezvous {}: a process of uid {} connected'.format(self.path, peer_uid))
                    r, n = [int(v) for v in _decode(_recv_msg(conn))]
                    if n != self.size or not (0 < r < self.size) or r in self.peers:
                        conn.close()
                        raise RuntimeError('rendezvous {}: unexpected peer (rank {}, world {}), this job has world {}'.format(
                            self.path, r, n, self.size))
                    self.peers[r] = conn
            except socket.timeout:
                raise RuntimeError('rendezvous {}: only {} of {} ranks arrived within {} s'.format(
                    self.path, len(self.peers) + 1, self.size, timeout))
            finally:
                srv.close()
                try:
                    os.unlink(self.path)
                except OSError:
                    pass
            for conn in self.peers.values():
                _send_msg(conn, b'ok')
        else:
            deadline = time.time() + timeout
            while True:
                s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    s.connect(self.path)
                    s.settimeout(timeout)
                    if _peer_uid(s) != os.getuid():
                        raise RuntimeError('rendezvous {}: the listener runs under another uid'.format(self.path))
                    _send_msg(s, _encode([self.rank, self.size]))
                    if _recv_msg(s) == b'ok':        # a dying stale listener would close instead
                        self.sock = s
                        break
                except RuntimeError:
                    s.close()
                    raise
                except (OSError, ConnectionError):
                    pass
                s.close()
                if time.time() > deadline:
                    raise RuntimeError('rendezvous {}: rank 0 did not answer within {} s'.format(self.path, timeout))
                time.sleep(0.05)

    def allgather(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank."""
        if self.size == 1:
            return [obj]
        if self.rank == 0:
            mine = _encode(obj)
            enc = [mine] + [_recv_msg(self.peers[r]) for r in range(1, self.size)]
            blob = _encode_list(enc)
            for r in range(1, self.size):
                _send_msg(self.peers[r], blob)
            return _decode(blob)
        _send_msg(self.sock, _encode(obj))
        return _decode(_recv_msg(self.sock))

    def bcast(self, obj, root=0):
        return self.allgather(obj if self.rank == root else None)[root]

    def barrier(self):
        self.allgather(None)

    def close(self):
        for c in list(self.peers.values()) + ([self.sock] if self.sock else []):
            try:
                c.close()
            except OSError:
                pass
        self.peers, self.sock = {}, None


class RcclComm(object):
    """The product's multi-GPU communicator: RCCL through libbdof.so, no torch."""
    backend = 'rccl'
    sharded = True

    def __init__(self, rank=None, size=None, local_rank=None, path=None):
        env = os.environ
        self.rank = int(env.get('RANK', '0') if rank is None else rank)
        self.size = int(env.get('WORLD_SIZE', '1') if size is None else size)
        self.local_rank = int(env.get('LOCAL_RANK', self.rank) if local_rank is None else local_rank)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # the pool's driver only supports dmabuf IPC
        self.group = SocketGroup(self.rank, self.size, path)
        self._h = None
        self._lib = None
        self._device = None
        self.always_reduce = bool(env.get('BDOF_FORCE_COMM'))   # run the collectives even with one rank (tests)

    # ---- binding to a device ctx ---------------------------------------------------------------------------------
    def attach(self, ctx):
        """Create the RCCL communicator on the ctx's device (collective: every rank calls it at the same point)."""
        if self._h is not None and self._device == ctx.device:
            return
        if self._h is not None:
            self._destroy()
        lib = ctx.lib
        where = [tuple(w) for w in self.group.allgather([socket.gethostname(), int(ctx.device)])]
        if len(set(where)) != len(where) and not os.environ.get('BDOF_RCCL_LIB'):       # a stand-in library (tests) may share a GPU
            raise RuntimeError('RCCL needs one device per rank, got {} — rehearse several ranks on one GPU with '
                               'BDOF_COMM_BACKEND=gloo'.format(where))
        uid = ctypes.create_string_buffer(128)
        if self.rank == 0:
            rc = lib.bdof_comm_unique_id(uid, 128)
            if rc != 0:
                raise RuntimeError('bdof_comm_unique_id failed ({}): {}'.format(rc, (lib.bdof_comm_last_error(None) or b'').decode()))
        raw = self.group.bcast(uid.raw if self.rank == 0 else None)
        h = ctypes.c_void_p()
        rc = lib.bdof_comm_create(ctypes.byref(h), int(ctx.device), self.size, self.rank, raw, 128)
        if rc != 0:
            raise RuntimeError('bdof_comm_create failed ({}): {}'.format(rc, (lib.bdof_comm_last_error(None) or b'').decode()))
        self._h, self._lib, self._device = h.value, lib, ctx.device
        # one line per rank on stderr: a wrong rank -> GPU map (two ranks on one device's neighbour, a masked device list) shows
        # here and nowhere else
        bus = ctypes.create_string_buffer(64)
        lib.bdof_device_pci_bus_id(int(ctx.device), bus, 64)
        import sys
        print('[bdof] rank {}/{} (local {}) pid {}: device {} pci {} rccl communicator of {} ranks ready'.format(
            self.rank, self.size, self.local_rank, os.getpid(), ctx.device, bus.value.decode() or '?', lib.bdof_comm_size(self._h)),
            file=sys.stderr, flush=True)

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError('libbdof collective failed ({}): {}'.format(rc, (self._lib.bdof_comm_last_error(self._h) or b'').decode()))

    def _ptr(self, buf, lo):
        from . import _lib
        return _lib._ptr(buf) + 4 * int(lo)

    # ---- device collectives --------------------------------------------------------------------------------------
    def start_allreduce(self, ctx, buf, lo, hi):
        t = ctypes.c_int(-1)
        self._check(self._lib.bdof_allreduce_grad(self._h, ctx.handle, self._ptr(buf, lo), int(hi) - int(lo), ctypes.byref(t)))
        return t.value

    def start_reduce_scatter(self, ctx, buf, lo, per):
        t = ctypes.c_int(-1)
        self._check(self._lib.bdof_reduce_scatter_grad(self._h, ctx.handle, self._ptr(buf, lo), int(per), ctypes.byref(t)))
        return t.value

    def start_allgather(self, ctx, buf, lo, per):
        t = ctypes.c_int(-1)
        self._check(self._lib.bdof_allgather_volume(self._h, ctx.handle, self._ptr(buf, lo), int(per), ctypes.byref(t)))
        return t.value

    def wait(self, ctx, ticket):
        if ticket is not None:
            self._check(self._lib.bdof_comm_wait(self._h, ctx.handle, int(ticket)))

    def allreduce_sum_device(self, ctx, buf):
        """Whole-buffer SUM (cnn_propagator/fullfield.py:350); the ctx stream continues behind it."""
        self.wait(ctx, self.start_allreduce(ctx, buf, 0, buf.nbytes // 4))
        return buf

    def bcast_device(self, ctx, buf, root=0):
        t = ctypes.c_int(-1)
        self._check(self._lib.bdof_bcast_volume(self._h, ctx.handle, self._ptr(buf, 0), buf.nbytes // 4, int(root), ctypes.byref(t)))
        self.wait(ctx, t.value)
        return buf

    # ---- host-side (control plane) -------------------------------------------------------------------------------
    def Barrier(self):
        self.group.barrier()

    def allreduce_sum_host(self, arr):
        return np.sum(self.group.allgather(np.asarray(arr)), axis=0)

    def allreduce_max_host(self, arr):
        return np.max(self.group.allgather(np.asarray(arr)), axis=0)

    def bcast_host(self, arr, root=0):
        return self.group.bcast(np.asarray(arr), root)

    def _destroy(self):
        if self._h is not None:
            self._lib.bdof_comm_destroy(self._h)
            self._h = None

    def close(self):
        self._destroy()
        self.group.close()


class TorchComm(object):
    """torch.distributed group (gloo): rehearsal backend.  Collectives are synchronous: the ctx stream is drained first,
    the result is complete on return, tickets are None.  reduce-scatter / all-gather are expressed through all_reduce
    (gloo has neither for device tensors): same results as RcclComm's, so FullfieldSolver's sharded step can be checked
    with two ranks on one GPU or on CPU tensors."""
    sharded = True

    def __init__(self, backend=None):
        from . import _lib
        if _lib._lib is not None and not _lib.TORCH_FIRST and _lib._lib.bdof_device_count() > 0:
            raise RuntimeError('libbdof.so was loaded before torch: the process would hold two HIP runtimes and torch could '
                               'not see the GPU.  Import torch first, or set BDOF_PRELOAD_TORCH=1 (BDOF_COMM_BACKEND=gloo does it).')
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.dist = dist
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group(backend=backend or 'gloo')
        self.size = dist.get_world_size()
        self.rank = dist.get_rank()
        self.local_rank = int(os.environ.get('LOCAL_RANK', self.rank))
        self.backend = dist.get_backend()
        self.always_reduce = bool(os.environ.get('BDOF_FORCE_COMM'))

    def attach(self, ctx):
        pass

    def Barrier(self):
        self.dist.barrier()

    def close(self):
        if self.dist.is_initialized():
            self.dist.destroy_process_group()

    def as_tensor(self, buf):
        """Zero-copy torch view of a DeviceBuffer (or pass a tensor through)."""
        if isinstance(buf, self.torch.Tensor):
            return buf.view(-1)
        t = self.torch.as_tensor(buf, device='cuda:{}'.format(self.torch.cuda.current_device()))
        return (self.torch.view_as_real(t) if t.is_complex() else t).view(-1)          # counts are in floats

    def _sync(self, ctx, t):
        if ctx is not None:
            ctx.sync()
        if t.is_cuda:
            self.torch.cuda.synchronize()

    def start_allreduce(self, ctx, buf, lo, hi):
        t = self.as_tensor(buf)
        self._sync(ctx, t)
        self.dist.all_reduce(t[int(lo):int(hi)], op=self.dist.ReduceOp.SUM)
        self._sync(None, t)
        return None

    def start_reduce_scatter(self, ctx, buf, lo, per):
        return self.start_allreduce(ctx, buf, lo, int(lo) + self.size * int(per))

    def start_allgather(self, ctx, buf, lo, per):
        t = self.as_tensor(buf)
        self._sync(ctx, t)
        lo, per = int(lo), int(per)
        seg = t[lo:lo + self.size * per]
        if self.rank > 0:
            seg[:self.rank * per].zero_()
        if self.rank < self.size - 1:
            seg[(self.rank + 1) * per:].zero_()
        self.dist.all_reduce(seg, op=self.dist.ReduceOp.SUM)
        self._sync(None, t)
        return None

    def wait(self, ctx, ticket):
        pass

    def allreduce_sum_device(self, ctx, buf):
        t = self.as_tensor(buf)
        self.start_allreduce(ctx, t, 0, t.numel())
        return buf

    def bcast_device(self, ctx, buf, root=0):
        t = self.as_tensor(buf)
        self._sync(ctx, t)
        self.dist.broadcast(t, src=root)
        self._sync(None, t)
        return buf

    def allreduce_sum_host(self, arr):
        t = self.torch.from_numpy(np.ascontiguousarray(arr))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.numpy()

    def allreduce_max_host(self, arr):
        t = self.torch.from_numpy(np.ascontiguousarray(arr))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.numpy()

    def bcast_host(self, arr, root=0):
        t = self.torch.from_numpy(np.ascontiguousarray(arr))
        self.dist.broadcast(t, src=root)
        return t.numpy()


def comm_backend():
    """'rccl' (native, default) or 'gloo' (torch.distributed rehearsal), from BDOF_COMM_BACKEND."""
    b = os.environ.get('BDOF_COMM_BACKEND', 'rccl').lower()
    if b == 'nccl':
        b = 'rccl'
    if b not in ('rccl', 'gloo'):
        raise ValueError("BDOF_COMM_BACKEND must be 'rccl' or 'gloo'")
    return b


def get_comm(backend=None):
    """The job's communicator: RcclComm when launched as several ranks (RANK / WORLD_SIZE / MASTER_* in the environment, as
    `python -m torch.distributed.run` or bench.py's self-launch export them), else the single-rank fallback."""
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        backend = backend or comm_backend()
        return RcclComm() if backend == 'rccl' else TorchComm(backend)
    if os.environ.get('BDOF_FORCE_COMM') and (backend or comm_backend()) == 'rccl':
        return RcclComm(rank=0, size=1, local_rank=int(os.environ.get('LOCAL_RANK', '0')))      # one rank through RCCL (tests, profiles)
    return PseudoComm()


def minibatch_schedule(n_theta, size, minibatch_size, rng=None, shuffle=True):
    """Index lists of one epoch (cnn_propagator/fullfield.py:196-203): shuffle, pad to a multiple of
    size*minibatch with the first indices (the evident intent of the malformed np.concatenate at :200,
    SURVEY quirk Q7), split, sort each chunk.  Rank r then takes chunk[r*mb:(r+1)*mb] (fullfield.py:343)."""
    ind = np.arange(n_theta)
    if shuffle:
        (rng or np.random).shuffle(ind)
    n_tot = size * minibatch_size
    if n_theta % n_tot > 0:
        ind = np.concatenate([ind, ind[:n_tot - n_theta % n_tot]])
    return [np.sort(ind[i:i + n_tot]) for i in range(0, len(ind), n_tot)]
