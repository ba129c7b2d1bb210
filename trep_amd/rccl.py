"""Torch-free multi-GPU plumbing: one process per GPU, batch slices, one RCCL all-gather.

The launcher only has to start N processes with RANK / LOCAL_RANK / WORLD_SIZE in the environment (what
``python -m torch.distributed.run`` does; the ranks themselves never import torch).  ``Communicator.from_env()``
then builds the RCCL communicator of ``csrc/comm.hip``:

* rank 0 asks the library for a 128-byte unique id and publishes it in a small file whose name is derived from the
  launcher's pid and the rendezvous port (both identical on every rank of one launch and different between
  launches); the other ranks poll for the file.  Plain file I/O -- nothing here touches the GPU;
* every rank calls ``tg_comm_create(device=LOCAL_RANK, world, rank, id)`` (ncclCommInitRank over xGMI).

Collectives offered: ``all_gather_rows`` (row blocks of device or host arrays, ragged by at most the shard
imbalance), ``max`` / ``sum`` of host scalars, ``barrier``.  The rollouts themselves need none of them.
"""
import contextlib
import ctypes
import os
import sys
import time

import numpy as np

from . import _lib
from .distributed import shard_bounds  # noqa: F401  (re-exported: the shard arithmetic is shared with the gloo tests)

ID_BYTES = 128
SUM, MAX, MIN = 0, 1, 2


def _rendezvous_path():
    key = os.environ.get("TREPAMD_RUN_KEY")
    if not key:
        key = "%d_%s_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"))
    base = os.environ.get("TREPAMD_RENDEZVOUS_DIR") or ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
    return os.path.join(base, "trepamd_rccl_%s.id" % key)


def exchange_unique_id(rank, world, make_id, timeout=300.0):
    """Rank 0 publishes make_id() (bytes); every rank returns the same bytes.  File based, atomic rename."""
    path = _rendezvous_path()
    if rank == 0:
        blob = bytes(make_id())
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as fh:
            fh.write(blob)
        os.replace(tmp, path)
        return blob, path
    t0 = time.time()
    try:        # a file left behind by an earlier launch with the same key is older than this process: never take it
        born = os.stat("/proc/%d" % os.getpid()).st_ctime - 30.0
    except OSError:
        born = 0.0
    while True:
        try:
            if os.stat(path).st_mtime >= born:
                with open(path, "rb") as fh:
                    blob = fh.read()
                if len(blob) == ID_BYTES:
                    return blob, path
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError("rank %d: no RCCL id at %s after %.0f s" % (rank, path, timeout))
        time.sleep(0.01)


@contextlib.contextmanager
def _stdout_to_stderr():
    """RCCL prints a version banner on the C-level stdout when the first communicator comes up; programs that print
    one JSON line on stdout (bench.py) must not see it there."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(saved, 1)
        os.close(saved)


class Communicator(object):
    """RCCL communicator of this process (world size 1 works too: the collectives are then local copies)."""

    def __init__(self, device, world, rank, unique_id):
        self.L = _lib.lib()
        self.device, self.world, self.rank = int(device), int(world), int(rank)
        buf = (ctypes.c_uint8 * ID_BYTES).from_buffer_copy(unique_id)
        with _stdout_to_stderr():
            self._h = self.L.tg_comm_create(self.device, self.world, self.rank, ctypes.cast(buf, ctypes.c_void_p))
            if self._h:     # the banner is printed (buffered) during the first collective at the latest
                self.L.tg_comm_barrier(self._h)
        if not self._h:
            raise _lib.LibraryError(self.L.tg_last_error().decode())
        self._bufs = {}

    @staticmethod
    def new_unique_id():
        L = _lib.lib()
        buf = (ctypes.c_uint8 * ID_BYTES)()
        with _stdout_to_stderr():
            _lib.check(L.tg_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
        return bytes(buf)

    @classmethod
    def from_env(cls):
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
        blob, path = exchange_unique_id(rank, world, cls.new_unique_id)
        comm = cls(local_rank, world, rank, blob)
        comm.barrier()                      # everyone has read the id: rank 0 may remove the file
        if rank == 0:
            try:
                os.remove(path)
            except OSError:
                pass
        return comm

    def info(self):
        """(ranks, rank, device) as RCCL itself reports them (ncclCommCount / ncclCommUserRank)."""
        out = np.zeros(3, dtype=np.int32)
        _lib.check(self.L.tg_comm_info(self._h, out.ctypes.data_as(_lib._c_ip)))
        return int(out[0]), int(out[1]), int(out[2])

    def wait_stream(self, hip_stream=None):
        """Order the communicator's stream after everything enqueued so far on `hip_stream` (None: the NULL stream; a
        batch's stream: ``mvi.stream``) -- an event, no host synchronisation."""
        _lib.check(self.L.tg_comm_wait_stream(self._h, hip_stream))

    def close(self):
        if self._h:
            for p in self._bufs.values():
                self.L.tg_device_free(self.device, p[0])
            self._bufs = {}
            self.L.tg_comm_destroy(self._h)
            self._h = None

    # -- scalars ---------------------------------------------------------------------------------------------
    def all_reduce(self, values, op=SUM):
        v = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
        _lib.check(self.L.tg_comm_all_reduce_host(self._h, v.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), v.size, op))
        return v

    def max(self, value):
        return float(self.all_reduce([value], MAX)[0])

    def sum(self, value):
        return float(self.all_reduce([value], SUM)[0])

    def barrier(self):
        _lib.check(self.L.tg_comm_barrier(self._h))

    # -- the data-path collective ------------------------------------------------------------------------------
    def _buffer(self, name, nbytes):
        cur = self._bufs.get(name)
        if cur is None or cur[1] < nbytes:
            if cur is not None:
                self.L.tg_device_free(self.device, cur[0])
            p = self.L.tg_device_alloc(self.device, max(nbytes, 8))
            if not p:
                raise _lib.LibraryError(self.L.tg_last_error().decode())
            cur = (p, nbytes)
            self._bufs[name] = cur
        return cur[0]

    def all_gather_device(self, send_ptr, recv_ptr, bytes_per_rank, synchronize=True, after=False):
        """recv [world][bytes_per_rank] <- send [bytes_per_rank] of every rank (device pointers).  `after`: the HIP stream
        that produces `send` (a batch's ``stream``; None for the NULL stream) -- the collective is ordered after it by an
        event; False: the caller has ordered it already (tg_comm_wait_stream) or synchronised."""
        if after is not False:
            self.wait_stream(after)
        _lib.check(self.L.tg_comm_all_gather(self._h, send_ptr, recv_ptr, bytes_per_rank))
        if synchronize:
            _lib.check(self.L.tg_comm_synchronize(self._h))

    def all_gather_rows(self, local, total_rows=None):
        """Concatenation in rank order of every rank's row block `local` [n_local][...] (host array).  Blocks made by
        shard_bounds differ by at most one row; shorter ones are padded for the fixed-size collective and trimmed."""
        local = np.ascontiguousarray(local, dtype=np.float64)
        counts = self.all_reduce(np.eye(self.world)[self.rank] * local.shape[0], SUM).astype(np.int64)
        n_max = int(counts.max())
        width = int(np.prod(local.shape[1:], dtype=np.int64))
        row_bytes = width * 8
        if n_max == 0 or row_bytes == 0:
            return np.zeros((int(counts.sum()),) + local.shape[1:])
        padded = np.zeros((n_max, width))
        padded[:local.shape[0]] = local.reshape(local.shape[0], width)
        send = self._buffer("send", n_max * row_bytes)
        recv = self._buffer("recv", self.world * n_max * row_bytes)
        _lib.check(self.L.tg_memcpy_h2d(self.device, send, padded.ctypes.data, n_max * row_bytes))
        self.all_gather_device(send, recv, n_max * row_bytes)
        out = np.zeros((self.world, n_max, width))
        _lib.check(self.L.tg_memcpy_d2h(self.device, out.ctypes.data, recv, self.world * n_max * row_bytes))
        parts = [out[r, :counts[r]] for r in range(self.world)]
        res = np.concatenate(parts, 0).reshape((-1,) + local.shape[1:])
        if total_rows is not None and res.shape[0] != total_rows:
            raise ValueError("gathered %d rows, expected %d" % (res.shape[0], total_rows))
        return res
