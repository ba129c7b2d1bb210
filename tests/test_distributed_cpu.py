"""world_size-2 gloo test of the batch-shard + all-gather path (runs on CPU).

Each rank integrates its contiguous slice of a batch (with the host emulation of the kernel source
standing in for the GPU) and the ranks all-gather the terminal states; the result must equal the
single-process rollout of the whole batch, bit for bit."""
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


def _worker(rank, world, port, B, N, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from trep_amd import systems, descriptor, distributed
    from emu_harness import EmuBatch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    system = systems.pend_on_cart()
    rng = np.random.default_rng(4)
    Q0 = np.stack([rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)], 1)
    U = rng.standard_normal((B, N, 1))
    lo, hi = distributed.shard_bounds(B, rank, world)
    e = EmuBatch(descriptor.flatten(system), hi - lo)
    e.initialize_from_configs(0.0, Q0[lo:hi], DT, Q0[lo:hi])
    X = e.rollout(N, DT, U[lo:hi], np.zeros((hi - lo, N, 0)))
    gathered = distributed.all_gather_rows(torch.from_numpy(X[:, N, :].copy()))
    tmax = distributed.max_over_ranks(1.0 + rank)
    assert tmax == float(world)
    np.save(os.path.join(out_dir, "gather_%d.npy" % rank), gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_batch():
    from trep_amd.distributed import shard_bounds
    for total in (1, 7, 8, 8192, 8193):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(total, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gloo_gather_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from trep_amd import systems, descriptor
    from emu_harness import EmuBatch
    B, N, world = 7, 20, 2          # odd batch: ragged shards
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, N, str(tmp_path)), nprocs=world, join=True)
    system = systems.pend_on_cart()
    rng = np.random.default_rng(4)
    Q0 = np.stack([rng.uniform(-1, 1, B), rng.uniform(-3, 3, B)], 1)
    U = rng.standard_normal((B, N, 1))
    e = EmuBatch(descriptor.flatten(system), B)
    e.initialize_from_configs(0.0, Q0, DT, Q0)
    X = e.rollout(N, DT, U, np.zeros((B, N, 0)))
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), "gather_%d.npy" % r))
        assert g.shape == (B, X.shape[2])
        assert np.array_equal(g, X[:, N, :])


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
