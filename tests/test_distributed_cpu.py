"""world_size > 1 tests of the batch-shard + all-gather path (run on CPU).

The collective LOGIC of the product -- `trep_amd/rccl.py::RowCollective`: pad every rank's row block to the largest shard,
fixed-size all-gather, trim, concatenate in rank order; scalar max / sum; barrier -- is what the RCCL `Communicator` runs on
the GPU box.  Here the same class runs over two CPU transports: gloo (`tests/transports.py::GlooTransport`, world 2) and
files in a directory (`rccl.FileTransport`, worlds 2, 3 and 8 with ragged shards).  Each rank integrates its contiguous
slice of a batch (the host emulation of the kernel source standing in for the GPU) and the ranks all-gather the terminal
states; the result must equal the single-process rollout of the whole batch, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DT = 0.01


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cart_inputs(B, N):
    rng = np.random.default_rng(4)
    Q0 = np.stack([rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)], 1)
    return Q0, rng.standard_normal((B, N, 1))


def _drive_collectives(comm, B, N, out_dir, tag):
    """What a rank of the product does around a rollout, through `comm` (any RowCollective)."""
    from trep_amd import systems, descriptor, rccl
    from emu_harness import EmuBatch
    rank, world = comm.rank, comm.world
    Q0, U = _cart_inputs(B, N)
    lo, hi = rccl.shard_bounds(B, rank, world)
    desc = descriptor.flatten(systems.pend_on_cart())
    e = EmuBatch(desc, max(hi - lo, 1))
    if hi > lo:
        e.initialize_from_configs(0.0, Q0[lo:hi], DT, Q0[lo:hi])
        X = e.rollout(N, DT, U[lo:hi], np.zeros((hi - lo, N, 0)))
        term = X[:, N, :].copy()
    else:
        term = np.zeros((0, desc.n_configs + desc.n_dyn + desc.n_kin))
    gathered = comm.all_gather_rows(term, total_rows=B)
    assert comm.max(1.0 + rank) == float(world)
    assert comm.sum(1.0 + rank) == world * (world + 1) / 2.0
    assert list(comm.row_counts(hi - lo)) == [b - a for a, b in (rccl.shard_bounds(B, r, world) for r in range(world))]
    comm.barrier()
    np.save(os.path.join(out_dir, "%s_%d.npy" % (tag, rank)), gathered)


def _gloo_worker(rank, world, port, B, N, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from transports import GlooTransport
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _drive_collectives(GlooTransport(), B, N, out_dir, "gloo")
    dist.barrier()
    dist.destroy_process_group()


def _file_worker(rank, world, directory, B, N, out_dir, break_trim):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from trep_amd import rccl
    if break_trim:      # a deliberate off-by-one in the product's trim: the test below must notice
        good = rccl.trim_gathered
        rccl.trim_gathered = lambda gathered, counts: good(gathered, [max(int(c) - (1 if r == 0 else 0), 0) + (1 if r == 1 else 0) for r, c in enumerate(counts)])
    comm = rccl.FileTransport(directory, world, rank, timeout=120.0)
    try:
        _drive_collectives(comm, B, N, out_dir, "file")
    finally:
        comm.close()


def _reference_terminal_states(B, N):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from trep_amd import systems, descriptor
    from emu_harness import EmuBatch
    Q0, U = _cart_inputs(B, N)
    e = EmuBatch(descriptor.flatten(systems.pend_on_cart()), B)
    e.initialize_from_configs(0.0, Q0, DT, Q0)
    return e.rollout(N, DT, U, np.zeros((B, N, 0)))[:, N, :]


def test_shard_bounds_cover_batch():
    from trep_amd.distributed import shard_bounds, padded_rows
    for total in (1, 7, 8, 8192, 8193):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(total, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
            assert padded_rows(total, world) == max(sizes)


def test_no_torch_in_the_product_package():
    """north_star: "no PyTorch" -- the ranks' collective is RCCL behind the C ABI; torch is test plumbing only."""
    import re
    pkg = os.path.join(ROOT, "trep_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import torch|from torch)", text, re.M), os.path.join(dirpath, f)


@pytest.mark.timeout(300)
def test_two_rank_gloo_gather_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    B, N, world = 7, 20, 2          # odd batch: ragged shards
    mp.spawn(_gloo_worker, args=(world, _free_port(), B, N, str(tmp_path)), nprocs=world, join=True)
    want = _reference_terminal_states(B, N)
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), "gloo_%d.npy" % r))
        assert g.shape == want.shape and np.array_equal(g, want)


def _run_file_world(tmp_path, world, B, N, break_trim=False):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    d = tmp_path / ("xchg%d_%d" % (world, B))
    d.mkdir()
    procs = [ctx.Process(target=_file_worker, args=(r, world, str(d), B, N, str(tmp_path), break_trim)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    return [p.exitcode for p in procs], d


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,B", [(2, 7), (3, 7), (3, 2), (8, 13)])
def test_ragged_shards_through_the_product_collective(tmp_path, world, B):
    """worlds 2, 3, 8; shards that differ by one row, and (3 ranks, 2 rows) a rank with NO rows."""
    N = 10
    codes, d = _run_file_world(tmp_path, world, B, N)
    assert codes == [0] * world
    want = _reference_terminal_states(B, N)
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), "file_%d.npy" % r))
        assert g.shape == want.shape and np.array_equal(g, want)
    assert not os.listdir(str(d))        # the transport cleans up after itself


def test_8193_rows_over_8_ranks_in_one_process(tmp_path):
    """The driver's 8-GPU shard shape (8193 = 8 x 1024 + 1) through the same logic, the eight 'ranks' as threads of this process."""
    import threading
    from trep_amd import rccl
    world, total, width = 8, 8193, 5
    data = np.arange(total * width, dtype=np.float64).reshape(total, width)
    got = [None] * world

    def work(r):
        comm = rccl.FileTransport(str(tmp_path), world, r)
        lo, hi = rccl.shard_bounds(total, r, world)
        got[r] = comm.all_gather_rows(data[lo:hi], total_rows=total)
        comm.close()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert all(g is not None and np.array_equal(g, data) for g in got)


@pytest.mark.timeout(300)
def test_a_broken_trim_in_the_product_is_caught(tmp_path):
    """The same world-2 run with a deliberate off-by-one patched into `rccl.trim_gathered` (rank 0's block one row short, rank 1's one
    row long): the gathered rows no longer equal the single-process rollout -- i.e. the tests above exercise that code."""
    B, N, world = 7, 10, 2
    codes, _ = _run_file_world(tmp_path, world, B, N, break_trim=True)
    want = _reference_terminal_states(B, N)
    if codes == [0] * world:
        g = np.load(os.path.join(str(tmp_path), "file_0.npy"))
        assert g.shape == want.shape and not np.array_equal(g, want)
    # (a non-zero exit code is the other way the broken trim shows: the row-count check of all_gather_rows raised)


def _rendezvous_worker(rank, world, key, directory, queue):
    import os
    os.environ["TREPAMD_RUN_KEY"] = key
    os.environ["TREPAMD_RENDEZVOUS_DIR"] = directory
    from trep_amd import rccl
    calls = []

    def make_id():
        calls.append(1)
        return bytes(range(128))
    blob, path = rccl.exchange_unique_id(rank, world, make_id, timeout=30.0)
    queue.put((rank, blob, len(calls)))


def test_rccl_unique_id_rendezvous_through_a_file(tmp_path):
    """The torch-free launcher glue of trep_amd/rccl.py without a GPU: rank 0 makes the 128-byte id once and publishes it
    atomically, the other ranks (started first, so they have to poll) read exactly those bytes."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, key = 3, "cpu-test-%d" % os.getpid()
    procs = [ctx.Process(target=_rendezvous_worker, args=(r, world, key, str(tmp_path), q)) for r in (1, 2, 0)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
    assert [g[0] for g in got] == [0, 1, 2]
    assert all(g[1] == bytes(range(128)) for g in got)
    assert [g[2] for g in got] == [1, 0, 0]          # only rank 0 asked the library for an id
    from trep_amd import rccl
    os.environ["TREPAMD_RUN_KEY"] = key
    os.environ["TREPAMD_RENDEZVOUS_DIR"] = str(tmp_path)
    try:
        assert os.path.exists(rccl._rendezvous_path())
    finally:
        del os.environ["TREPAMD_RUN_KEY"], os.environ["TREPAMD_RENDEZVOUS_DIR"]


@pytest.mark.timeout(120)
def test_bench_self_launcher_dry_run(tmp_path):
    """`bench.py --gpus 2` with no launcher in the environment starts its own two rank processes (RANK / LOCAL_RANK /
    WORLD_SIZE / TREPAMD_RUN_KEY) before any GPU call and prints rank 0's line; with --dry-run-ranks the ranks only do the
    communicator-id rendezvous and the shard arithmetic, so this runs without a GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TREPAMD_RUN_KEY")}
    env["TREPAMD_RENDEZVOUS_DIR"] = str(tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-ranks", "--batch", "8193"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=100)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["dry_run"] and out["n_gpus"] == 2 and out["ranks_reported"] == 2 and out["same_id_on_every_rank"]
    assert out["shards"] == [[0, 4097], [4097, 8193]]
    assert out["run_key"].startswith("self")
    # rank 0's real line carries the roofline (with the counter traffic), the fp64 object and the CPU baseline for ANY world size
    for key in ("roofline", "fp64", "cpu_baseline", "discopt", "strong_scaling", "n_gpus", "rccl_ranks"):
        assert key in out["line_keys"], (key, out["line_keys"])
    assert not os.listdir(str(tmp_path))          # the rendezvous files are gone


def test_bench_refuses_a_rank_count_that_contradicts_the_launcher():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run-ranks"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=100)
    assert r.returncode != 0 and "--gpus 4" in r.stderr
