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

The collective LOGIC -- who pads what to the largest shard, what is trimmed, the order of the concatenation, the
scalar reductions -- lives in ``RowCollective`` and is written against two transport primitives only
(``_all_reduce_host`` and ``_all_gather_block``).  ``Communicator`` implements them on RCCL (``tg_comm_*``);
``FileTransport`` implements them with files in a shared directory (no GPU, no library: the world-2 / world-3 CPU
tests drive the product's pad / trim / concatenate code through it, ``tests/test_distributed_cpu.py``).
"""
import contextlib
import ctypes
import os
import sys
import time

import numpy as np

from .distributed import shard_bounds, padded_rows  # noqa: F401  (re-exported: the shard arithmetic is shared with the tests)

ID_BYTES = 128
SUM, MAX, MIN = 0, 1, 2


def _L():
    """The ctypes binding, imported on first use: the file transport and the shard arithmetic need no native library."""
    from . import _lib
    return _lib


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


class RowCollective(object):
    """The collectives of the path in terms of two transport primitives.  A transport provides ``world``, ``rank`` and

    * ``_all_reduce_host(v, op)``: element-wise SUM / MAX / MIN over the ranks of a float64 host vector, in place;
    * ``_all_gather_block(block)``: ``block`` [n_max][width] float64 (the same shape on every rank) -> [world][n_max][width].
    """
    world = 1
    rank = 0

    def _all_reduce_host(self, v, op):
        raise NotImplementedError

    def _all_gather_block(self, block):
        raise NotImplementedError

    # -- scalars ---------------------------------------------------------------------------------------------
    def all_reduce(self, values, op=SUM):
        v = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
        self._all_reduce_host(v, op)
        return v

    def max(self, value):
        return float(self.all_reduce([value], MAX)[0])

    def sum(self, value):
        return float(self.all_reduce([value], SUM)[0])

    def barrier(self):
        self.all_reduce([0.0], SUM)

    def row_counts(self, n_local):
        """Rows held by every rank (one SUM reduction of a one-hot vector)."""
        return self.all_reduce(np.eye(self.world)[self.rank] * int(n_local), SUM).astype(np.int64)

    # -- the data-path collective ------------------------------------------------------------------------------
    def all_gather_rows(self, local, total_rows=None):
        """Concatenation in rank order of every rank's row block `local` [n_local][...] (host array).  Blocks made by
        shard_bounds differ by at most one row; shorter ones are padded for the fixed-size collective and trimmed."""
        local = np.ascontiguousarray(local, dtype=np.float64)
        counts = self.row_counts(local.shape[0])
        n_max = int(counts.max())
        width = int(np.prod(local.shape[1:], dtype=np.int64))
        if n_max == 0 or width == 0:
            return np.zeros((int(counts.sum()),) + local.shape[1:])
        padded = np.zeros((n_max, width))
        padded[:local.shape[0]] = local.reshape(local.shape[0], width)
        out = np.asarray(self._all_gather_block(padded)).reshape(self.world, n_max, width)
        res = trim_gathered(out, counts).reshape((-1,) + local.shape[1:])
        if total_rows is not None and res.shape[0] != total_rows:
            raise ValueError("gathered %d rows, expected %d" % (res.shape[0], total_rows))
        return res


def trim_gathered(gathered, counts):
    """[world][n_max][width] as the fixed-size collective delivers it -> the first counts[r] rows of every rank's block,
    concatenated in rank order (what ``bench.py`` and the discopt line search read after ``all_gather_device`` too)."""
    gathered = np.asarray(gathered)
    return np.concatenate([gathered[r, :int(counts[r])] for r in range(gathered.shape[0])], 0)


class FileTransport(RowCollective):
    """Test transport: the two primitives through files in a directory every rank can see.  No GPU, no native library --
    the product's collective logic (RowCollective) runs unchanged on top of it.  Every collective call is numbered; a
    rank publishes ``<seq>.<rank>.npy`` by atomic rename and polls for the others'; files of call n are removed by their
    owner once call n + 1 has been entered by every rank (its files exist), so a directory serves any number of calls."""

    def __init__(self, directory, world, rank, timeout=120.0):
        self.dir, self.world, self.rank, self.timeout = str(directory), int(world), int(rank), float(timeout)
        self._seq = 0
        self._mine = []

    def _exchange(self, array):
        seq = self._seq
        self._seq += 1
        path = os.path.join(self.dir, "%06d.%d.npy" % (seq, self.rank))
        tmp = path + ".tmp"
        with open(tmp, "wb") as fh:
            np.save(fh, np.ascontiguousarray(array))
        os.replace(tmp, path)
        parts = []
        t0 = time.time()
        for r in range(self.world):
            f = os.path.join(self.dir, "%06d.%d.npy" % (seq, r))
            while not os.path.exists(f):
                if time.time() - t0 > self.timeout:
                    raise TimeoutError("rank %d: rank %d never entered collective %d" % (self.rank, r, seq))
                time.sleep(0.001)
            parts.append(np.load(f))
        # everyone has entered call `seq`, hence has finished reading call seq - 1: this rank's older files can go
        for old in self._mine:
            try:
                os.remove(old)
            except OSError:
                pass
        self._mine = [path]
        return parts

    def _all_reduce_host(self, v, op):
        parts = np.stack(self._exchange(v), 0)
        v[:] = parts.sum(0) if op == SUM else (parts.max(0) if op == MAX else parts.min(0))

    def _all_gather_block(self, block):
        return np.stack(self._exchange(block), 0)

    def close(self):
        """Leave: every rank drops a marker after its last collective; rank 0 waits for all markers and then removes what is left
        (a rank cannot remove its own last files itself: a slower rank may not have read them yet)."""
        marker = os.path.join(self.dir, "done.%d" % self.rank)
        with open(marker + ".tmp", "w") as fh:
            fh.write("%d" % self._seq)
        os.replace(marker + ".tmp", marker)
        if self.rank != 0:
            return
        t0 = time.time()
        for r in range(self.world):
            while not os.path.exists(os.path.join(self.dir, "done.%d" % r)):
                if time.time() - t0 > self.timeout:
                    raise TimeoutError("rank %d never left" % r)
                time.sleep(0.001)
        for name in os.listdir(self.dir):
            if name.startswith("done.") or name.endswith(".npy"):
                try:
                    os.remove(os.path.join(self.dir, name))
                except OSError:
                    pass


class Communicator(RowCollective):
    """RCCL communicator of this process (world size 1 works too: the collectives are then local copies)."""

    def __init__(self, device, world, rank, unique_id):
        self.L = _L().lib()
        self.device, self.world, self.rank = int(device), int(world), int(rank)
        buf = (ctypes.c_uint8 * ID_BYTES).from_buffer_copy(unique_id)
        with _stdout_to_stderr():
            self._h = self.L.tg_comm_create(self.device, self.world, self.rank, ctypes.cast(buf, ctypes.c_void_p))
            if self._h:     # the banner is printed (buffered) during the first collective at the latest
                self.L.tg_comm_barrier(self._h)
        if not self._h:
            raise _L().LibraryError(self.L.tg_last_error().decode())
        self._bufs = {}

    @staticmethod
    def new_unique_id():
        L = _L().lib()
        buf = (ctypes.c_uint8 * ID_BYTES)()
        with _stdout_to_stderr():
            _L().check(L.tg_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
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
        _L().check(self.L.tg_comm_info(self._h, out.ctypes.data_as(_L()._c_ip)))
        return int(out[0]), int(out[1]), int(out[2])

    def wait_stream(self, hip_stream=None):
        """Order the communicator's stream after everything enqueued so far on `hip_stream` (None: the NULL stream; a
        batch's stream: ``mvi.stream``) -- an event, no host synchronisation."""
        _L().check(self.L.tg_comm_wait_stream(self._h, hip_stream))

    def close(self):
        if self._h:
            for p in self._bufs.values():
                self.L.tg_device_free(self.device, p[0])
            self._bufs = {}
            self.L.tg_comm_destroy(self._h)
            self._h = None

    # -- transport primitives (RowCollective) ----------------------------------------------------------------
    def _all_reduce_host(self, v, op):
        _L().check(self.L.tg_comm_all_reduce_host(self._h, v.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), v.size, op))

    def barrier(self):
        _L().check(self.L.tg_comm_barrier(self._h))

    def _all_gather_block(self, block):
        nbytes = block.size * 8
        send = self._buffer("send", nbytes)
        recv = self._buffer("recv", self.world * nbytes)
        _L().check(self.L.tg_memcpy_h2d(self.device, send, block.ctypes.data, nbytes))
        self.all_gather_device(send, recv, nbytes)
        out = np.zeros((self.world,) + block.shape)
        _L().check(self.L.tg_memcpy_d2h(self.device, out.ctypes.data, recv, self.world * nbytes))
        return out

    # -- the data-path collective ------------------------------------------------------------------------------
    def _buffer(self, name, nbytes):
        cur = self._bufs.get(name)
        if cur is None or cur[1] < nbytes:
            if cur is not None:
                self.L.tg_device_free(self.device, cur[0])
            p = self.L.tg_device_alloc(self.device, max(nbytes, 8))
            if not p:
                raise _L().LibraryError(self.L.tg_last_error().decode())
            cur = (p, nbytes)
            self._bufs[name] = cur
        return cur[0]

    def all_gather_device(self, send_ptr, recv_ptr, bytes_per_rank, synchronize=True, after=False):
        """recv [world][bytes_per_rank] <- send [bytes_per_rank] of every rank (device pointers).  `after`: the HIP stream
        that produces `send` (a batch's ``stream``; None for the NULL stream) -- the collective is ordered after it by an
        event; False: the caller has ordered it already (tg_comm_wait_stream) or synchronised."""
        if after is not False:
            self.wait_stream(after)
        _L().check(self.L.tg_comm_all_gather(self._h, send_ptr, recv_ptr, bytes_per_rank))
        if synchronize:
            _L().check(self.L.tg_comm_synchronize(self._h))
