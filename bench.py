#!/usr/bin/env python3
"""bench.py -- DEL-steps/s x batch for the ~40-DOF puppet (BASELINE.json metric), 1..8 MI355X.

One "step" (in the driver's sense: --steps K, --warmup W) is one pass of the hot path over one
batch: a device-resident rollout of N_ROLLOUT = 200 MidpointVI steps for BATCH = 8192 independent
puppet trajectories per GPU, in ONE kernel launch, inputs (initial state, kinematic string schedule
K[b,k,:]) already resident in HBM, states X[b,k,:] written back to HBM.  value = DEL steps of all
ranks / wall time between barriers.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python bench.py --gpus N --steps K --warmup W         (no launcher: starts its own N rank processes, see self_launch)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: the batch shards trivially, no data-path collective; after each rollout the ranks all-gather the
terminal states [B_local][nX] over RCCL/xGMI (what a discopt line-search consumes).  The ranks never import
torch: the launcher only provides RANK / LOCAL_RANK / WORLD_SIZE, the communicator is RCCL behind the C ABI
(include/trep_amd.h tg_comm_*, trep_amd/rccl.py).  The headline value is WEAK scaling (8192 trajectories per GPU);
the same run also measures BASELINE config 3 as written -- 8192 trajectories in total, B/N per GPU -- and reports
it as the "strong_scaling" object, and config 4 (discopt on 256 puppet seeds, N = 1000) as the "discopt" object.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X datasheet, vector fp64 (secondary figure; the path is fp64-VALU bound)
REFERENCE_TREP_STEPS_PER_S = 345.0  # reference _trep, 1 Xeon 2.1 GHz core, Puppet-40


def build_secondary(args, rank):
    """BASELINE configs[1] (pend-on-cart) and configs[4] (scissor lift): parity-test systems, benchmarked as
    secondary lines with the synthetic inputs of SURVEY.md §8(d)."""
    from trep_amd import systems
    B, N, dt = args.batch, args.rollout_steps, 0.01
    if args.system == "cart":
        system = systems.pend_on_cart()
        rng = np.random.default_rng(20250 + 2 + 1000 * rank)
        Q0 = np.stack([rng.uniform(-1, 1, B), rng.uniform(-np.pi, np.pi, B)], 1)
        U = rng.standard_normal((B, N, 1)) * 2.0
        return system, Q0, U, None, dt
    if args.system == "puppet-basic":
        # examples/puppet-basic.py: constraint-consistent poses captured from the reference's satisfy_constraints
        system = systems.puppet_basic()
        ics = np.load(os.path.join(ROOT, "tests", "golden", "puppet_basic.npz"))["ic_set"]
        Q0 = np.tile(np.roll(ics, rank, axis=0), ((B + len(ics) - 1) // len(ics), 1))[:B]
        return system, Q0, None, None, dt      # only len(ics) = 16 DISTINCT poses: reported as such (distinct_initial_conditions)
    system = systems.scissor_lift(4)
    rng = np.random.default_rng(20250 + 5 + 1000 * rank)
    th = rng.uniform(0.03 * np.pi, 0.12 * np.pi, B)
    Q0 = np.array([systems.scissor_q(system, t) for t in th])
    return system, Q0, None, None, dt


def build_workload(args, rank):
    from trep_amd import systems
    system = systems.puppet()
    B, N, dt = args.batch, args.rollout_steps, 0.01
    nd = system.nQd
    # every trajectory of the batch gets its own seeded initial condition (about 5 s of host
    # forward kinematics for 8192 poses); each rank takes its own seed so shards differ.
    Q0 = systems.puppet_initial_conditions(system, B, seed=20250 + 3 + 1000 * rank)
    K = systems.puppet_string_schedule(system, Q0[:, nd:], N, dt)
    return system, Q0, K, dt


def host_cores():
    """CPUs this process may actually use: affinity mask and cgroup CPU quota, not the machine's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def cpu_baseline(system, Q0, K, dt, budget_s=12.0, threads=None):
    """Oracle (C restatement of the reference) on a bounded sample of the same workload: one oracle object per
    host thread, every thread rolling out its own trajectories of the batch (the C calls release the GIL)."""
    import threading
    from trep_amd import descriptor
    from oracle.oracle import OracleMVI
    threads = max(1, min(threads or min(host_cores(), 64), len(Q0)))
    desc = descriptor.flatten(system)
    oracles = [OracleMVI(desc) for _ in range(threads)]
    counts = [0] * threads
    chunk = 50
    t0 = time.perf_counter()

    def work(i):
        o, b = oracles[i], i
        while time.perf_counter() - t0 < budget_s and b < len(Q0):
            o.initialize_from_configs(0.0, Q0[b], dt, Q0[b])
            k = 0
            while k < K.shape[1] and time.perf_counter() - t0 < budget_s:
                n = min(chunk, K.shape[1] - k)
                o.rollout(n, dt, None, K[b, k:k + n], want_X=False)
                k += n
                counts[i] += n
            b += threads

    pool = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    for t in pool:
        t.start()
    for t in pool:
        t.join()
    el = time.perf_counter() - t0
    steps = sum(counts)
    return {"value": steps / el, "unit": "DEL-steps/s", "cores": threads, "kind": "port",
            "sample": "%d puppet-40 DEL steps of the same rollouts, oracle/libtreporacle.so on %d host threads (%d usable of %d cores), %.1f s"
                      % (steps, threads, host_cores(), os.cpu_count() or 1, el),
            "per_thread_steps_per_s": steps / el / threads,
            "reference_trep_single_core_steps_per_s": REFERENCE_TREP_STEPS_PER_S}


def pmc_traffic(batch, rollout_steps, workload_prefix):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_traffic.json), if one
    exists for exactly this workload; PMC collection needs rocprofv3 around the process, so bench.py
    cannot measure it live (tools/collect_profile.sh + tools/summarize_pmc.py produce the file)."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("global_batch") == batch and d.get("rollout_steps") == rollout_steps and str(d.get("workload", "")).startswith(workload_prefix):
            best = (d["hbm_bytes_per_launch"], os.path.relpath(path, ROOT))
    return best


def pmc_fp64(batch, rollout_steps, workload_prefix):
    """fp64 flop per launch estimated from the committed SQ instruction counters (profiles/*_fp64.json)."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fp64.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("global_batch") == batch and d.get("rollout_steps") == rollout_steps and str(d.get("workload", "")).startswith(workload_prefix):
            best = (d["estimated_fp64_flop_per_launch"], d["valu_f64_wave_instructions_per_launch"], os.path.relpath(path, ROOT))
    return best


def fp64_fma_ceiling():
    """Sustained v_fma_f64 rate of this part with two waves per SIMD (tools/micro/mfma_f64_rate.hip): read from the newest
    profiles/r*_mfma_f64_rate.txt, with the file it came from."""
    import glob
    import re
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mfma_f64_rate.txt")), reverse=True):
        try:
            for line in open(path):
                m = re.match(r"waves/SIMD 2:.*v_fma_f64:.*\(([0-9.]+) TFLOP/s\)\s*$", line)
                if m:
                    return float(m.group(1)), os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def expected_line_keys(args, world):
    """Top-level keys of rank 0's JSON line for this invocation (the same for every world size: the roofline's counter traffic, the fp64
    object and the CPU baseline are rank 0's own figures and do not depend on how many ranks run)."""
    keys = ["metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"]
    if world > 1:
        keys.append("%s_scaling" % ("strong" if args.scaling == "weak" else "weak"))
    if args.system == "puppet":
        keys.append("fp64")
        if not args.no_discopt:
            keys.append("discopt")
    if not args.no_cpu_baseline:
        keys.append("cpu_baseline")
    return keys


def self_launch(args):
    """`bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N rank processes ourselves -- plain child
    processes with RANK / LOCAL_RANK / WORLD_SIZE / TREPAMD_RUN_KEY in their environment, BEFORE this process has made any
    GPU call (it never makes one) -- pass rank 0's JSON line through and exit with the worst return code."""
    import subprocess
    n = args.gpus
    env0 = dict(os.environ)
    env0["WORLD_SIZE"] = str(n)
    env0["TREPAMD_RUN_KEY"] = "self%d_%d" % (os.getpid(), int(time.time() * 1e3) % 10 ** 9)
    env0.setdefault("MASTER_ADDR", "127.0.0.1")
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(env0)
        env["RANK"] = env["LOCAL_RANK"] = str(r)
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * n
    first_failure = None
    while any(c is None for c in codes):
        for i, pr in enumerate(procs):
            if codes[i] is None:
                codes[i] = pr.poll()
                if codes[i] not in (None, 0) and first_failure is None:
                    first_failure = time.time()
        # a rank died: the others hang in a collective -- give them a moment, then stop exactly those children
        if first_failure is not None and time.time() - first_failure > 30.0:
            for i, pr in enumerate(procs):
                if codes[i] is None:
                    pr.kill()
                    codes[i] = pr.wait()
        time.sleep(0.05)
    reader.join(timeout=10.0)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    sys.exit(max(abs(c) for c in codes))


def dry_run_rank(args, world, rank):
    """--dry-run-ranks: what a rank does before its first GPU call -- the file rendezvous of the communicator id and the
    shard arithmetic -- and nothing else.  Used by the CPU test of the self-launcher (no GPU, no RCCL)."""
    import hashlib
    from trep_amd import rccl
    from trep_amd.distributed import shard_bounds, padded_rows
    blob, path = rccl.exchange_unique_id(rank, world, lambda: os.urandom(rccl.ID_BYTES), timeout=60.0)
    digest = hashlib.sha256(blob).hexdigest()[:16]
    mine = "%s.rank%d" % (path, rank)
    with open(mine + ".tmp", "w") as fh:
        fh.write(digest)
    os.replace(mine + ".tmp", mine)
    if rank != 0:
        return
    seen = []
    t0 = time.time()
    for r in range(world):
        f = "%s.rank%d" % (path, r)
        while not os.path.exists(f):
            if time.time() - t0 > 60.0:
                raise TimeoutError("rank %d never reported" % r)
            time.sleep(0.01)
        seen.append(open(f).read())
        os.remove(f)
    os.remove(path)
    shards = [shard_bounds(args.batch, r, world) for r in range(world)]
    print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_reported": len(seen), "same_id_on_every_rank": len(set(seen)) == 1,
                      "shards": shards, "run_key": os.environ.get("TREPAMD_RUN_KEY"), "line_keys": expected_line_keys(args, world)}))


def measure_rollouts(args, system, Q0, U, K, dt, B, N, device, comm, gather_rows):
    """W warm-up + K timed passes of B trajectories x N steps on this rank; returns timing and status figures."""
    import trep_amd
    from trep_amd import _lib
    L = _lib.lib()
    mvi = trep_amd.BatchMidpointVI(system, B, device=device, specialize="auto" if args.specialize else False)
    mvi.predictor = args.predictor
    specialised = mvi._specialized is not None
    nX = mvi.nX
    K_dev = mvi.device_array(K) if K is not None else None
    U_dev = mvi.device_array(U) if U is not None else None
    X_dev = None if args.no_x else mvi.device_empty(B * (N + 1) * nX)
    term = gather = rows = None
    if comm is not None:
        term = mvi.device_empty(gather_rows * nX)            # this rank's terminal states, padded to the largest shard
        gather = mvi.device_empty(comm.world * gather_rows * nX)
        if X_dev is not None:
            idx = (np.arange(B, dtype=np.int64) * (N + 1) + N).astype(np.int32)
            rows = L.tg_device_alloc(device, max(idx.nbytes, 8))
            mvi._owned_dev.append(rows)
            _lib.check(L.tg_memcpy_h2d(device, rows, idx.ctypes.data, idx.nbytes))

    def one_pass():
        mvi.restore()   # device-to-device: every pass integrates the same N-step window
        mvi.rollout_device(N, dt, U_dev, K_dev, X_dev)
        if comm is not None:
            # the collective runs on the communicator's own stream: ordered after its producer by an event (tg_comm_wait_stream)
            if rows is not None:   # term[b] = X[b][N], a NULL-stream kernel (the NULL stream waits for the batch's blocking stream)
                _lib.check(L.tg_copy_rows(device, B, nX, None, rows, X_dev, term))
                comm.all_gather_device(term, gather, gather_rows * nX * 8, synchronize=False, after=None)
            else:                  # --no-x: nothing between rollout and collective, the batch's stream is the producer
                comm.all_gather_device(term, gather, gather_rows * nX * 8, synchronize=False, after=mvi.stream)

    def sync():
        mvi.synchronize()
        if comm is not None:
            _lib.check(L.tg_comm_synchronize(comm._h))
            comm.barrier()

    mvi.initialize_from_configs(0.0, Q0, dt, Q0)
    mvi.snapshot()
    for _ in range(args.warmup):
        one_pass()
    sync()
    mvi.timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    sync()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.max(elapsed)
    n_launch, kernel_ms = mvi.timing(reset=True)
    iters, status = mvi.status()
    kinfo = mvi.kernel_info()
    if specialised and "rollout" not in kinfo["spec_launched"]:
        raise RuntimeError("the specialised rollout kernel was loaded but the launches went through the generic one: %r" % (kinfo,))
    res = {"elapsed": elapsed, "launches": n_launch, "kernel_ms": kernel_ms, "newton_iterations": int(iters.sum()),
           "failed": int((status != 0).sum()), "info": mvi.info(), "specialised": bool(specialised), "nX": mvi.nX, "nU": mvi.nU, "nc": mvi.nc,
           "spec_library": os.path.relpath(kinfo["spec_library"], ROOT) if kinfo["spec_library"] else None}
    mvi.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8192, help="trajectories per GPU (weak scaling) / in total (strong scaling)")
    ap.add_argument("--rollout-steps", type=int, default=200, help="DEL steps per trajectory per pass")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="headline line: weak = --batch per GPU (default), strong = --batch in total (BASELINE config 3 as written); "
                         "with more than one rank the other mode is measured too and reported next to it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-discopt", action="store_true", help="skip the discopt iterations/s measurement (BASELINE config 4)")
    ap.add_argument("--discopt-seeds", type=int, default=256)
    ap.add_argument("--discopt-horizon", type=int, default=1000)
    ap.add_argument("--no-x", action="store_true", help="do not write the state trajectory X to HBM")
    ap.add_argument("--system", choices=["puppet", "puppet-basic", "cart", "scissor"], default="puppet",
                    help="puppet = the BASELINE metric; cart / scissor = secondary lines")
    ap.add_argument("--predictor", choices=["reference", "extrapolate"], default="reference",
                    help="Newton initial guess of the rollout; 'reference' (default) keeps the reference's semantics and "
                         "iteration counts, 'extrapolate' is an opt-in warm start (reported separately, not the headline)")
    ap.add_argument("--no-specialize", dest="specialize", action="store_false",
                    help="run the generic rollout kernel (schedule interpreted at run time) instead of the system-specialised one "
                         "(trep_amd/specialize.py: the same kernel source compiled against the system's schedule)")
    ap.add_argument("--force-dist", action="store_true", help="take the RCCL path even with one rank (self-test)")
    ap.add_argument("--dry-run-ranks", action="store_true",
                    help="ranks only do the communicator-id rendezvous and the shard arithmetic (no GPU call): CPU test of the launcher")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)          # never returns; this process makes no GPU call
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    if args.dry_run_ranks:
        dry_run_rank(args, world, rank)
        return
    want_cpu = not args.no_cpu_baseline and rank == 0 and args.system == "puppet"      # rank 0's host cores, whatever the world size
    if want_cpu:      # compile the checker (a `make` child process) BEFORE this process touches the GPU
        from oracle import oracle as _oracle
        _oracle.build()
    import trep_amd  # noqa: F401

    from trep_amd.distributed import shard_bounds, padded_rows
    N = args.rollout_steps
    U = None
    if args.system == "puppet":
        system, Q0, K, dt = build_workload(args, rank)
    else:
        system, Q0, U, K, dt = build_secondary(args, rank)
    B = args.batch
    if args.specialize:    # hipcc as a child process (about 5 s, cached in trep_amd/_spec): before this process touches the GPU
        from trep_amd import specialize
        specialize.build(system)
    comm = None
    if world > 1 or args.force_dist:      # first GPU call of the process
        from trep_amd import rccl
        comm = rccl.Communicator.from_env()
    rccl_ranks = 1
    if comm is not None:      # what RCCL itself reports (ncclCommCount), not what the launcher said
        rccl_ranks = comm.info()[0]
        if rccl_ranks != world:
            raise RuntimeError("RCCL reports %d ranks, the launcher %d" % (rccl_ranks, world))

    def shard_of_global():    # this rank's slice of ONE global batch of --batch trajectories (strong scaling)
        lo, hi = shard_bounds(B, rank, world)
        return (Q0[lo:hi], None if U is None else U[lo:hi], None if K is None else K[lo:hi], hi - lo,
                padded_rows(B, world))

    modes = [args.scaling] + ([("strong" if args.scaling == "weak" else "weak")] if world > 1 else [])
    results = {}
    for mode in modes:
        if mode == "weak":
            results[mode] = (measure_rollouts(args, system, Q0, U, K, dt, B, N, local_rank, comm, B), world * B, B)
        else:
            q, u, k, b_local, b_max = shard_of_global()
            results[mode] = (measure_rollouts(args, system, q, u, k, dt, b_local, N, local_rank, comm, b_max), B, b_local)

    discopt = None
    if not args.no_discopt and args.system == "puppet":
        import bench_discopt
        d = bench_discopt.measure(args.discopt_seeds, args.discopt_horizon, quasi=1, newton=1, comm=comm, device=local_rank)
        discopt = {k: d[k] for k in ("iters_per_s", "seeds", "horizon", "n_gpus", "seed_iterations_counted", "elapsed_s",
                                     "s_per_batched_quasi_step", "s_per_batched_newton_step", "armijo_failures",
                                     "mean_cost_before_after_per_step", "mean_final_cost_successful_seeds")}
        discopt["unit"] = "DOptimizer.step equivalents (seed-iterations) per second: 1 quasi-Newton + 1 Newton step of every seed, after one untimed warm-up step of each method"
        discopt["scaling"] = "strong (seeds sharded over the ranks)"
        discopt["reference_s_per_newton_step_one_seed"] = d["reference_s_per_newton_step_N1000_one_seed"]

    if rank == 0:
        r, global_batch, b_local = results[args.scaling]
        nX, nU, nc = r["nX"], r["nU"], r["nc"]
        elapsed = r["elapsed"]
        del_steps = float(global_batch) * N * args.steps
        value = del_steps / elapsed
        avg_kernel_s = r["kernel_ms"] / 1e3 / max(r["launches"], 1)
        bytes_per_step = 8.0 * (2 * nX + nU + nc)   # read X_k, U_k; write X_k+1, lambda
        algo_bytes = bytes_per_step * b_local * N
        achieved = algo_bytes / avg_kernel_s / 1e9
        # per-launch counter figures of ONE GPU's launch (profiles/): they apply to this rank's launch whenever its batch is the profiled one
        traffic = pmc_traffic(b_local, N, "Puppet(string_constraints=True)") if args.system == "puppet" else None
        out = {
            "metric": "DEL-steps/sec x batch (%s, fp64)" % ("puppet ~40-DOF" if args.system == "puppet" else args.system),
            "value": value, "unit": "DEL-steps/s", "n_gpus": rccl_ranks, "rccl_ranks": rccl_ranks if comm is not None else None,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("Puppet(string_constraints=True) nq=40 nd=22 nk=18 nc=6" if args.system == "puppet" else
                                    {"cart": "pend-on-cart nd=2 nu=1", "scissor": "scissor-4 nd=9 nc=8",
                                     "puppet-basic": "examples/puppet-basic.py nd=22 nc=6 (fixed-length strings)"}[args.system]) +
                                   ", batch=%d rollouts %s x %d DEL steps, dt=0.01" % (B, "per GPU" if args.scaling == "weak" else "in total", N),
                       "global_batch": global_batch, "rollout_steps": N, "parallelism": "batch-shard x%d" % world,
                       "collective": "RCCL all-gather of terminal states (C ABI tg_comm_*, no torch)" if comm is not None else None,
                       "distinct_initial_conditions": int(len(np.unique(np.ascontiguousarray(Q0), axis=0))) * (world if args.scaling == "weak" and args.system != "puppet-basic" else 1),
                       "team": r["info"]["team"], "lds_bytes_per_trajectory": r["info"]["lds_bytes_per_trajectory"],
                       "newton_iterations_per_step": r["newton_iterations"] / float(b_local * N), "failed_trajectories": r["failed"],
                       "writes_X": not args.no_x, "newton_initial_guess": args.predictor,
                       "kernel_variant": "system-specialised (schedule compiled in)" if r["specialised"] else "generic (schedule interpreted)",
                       "spec_library": r["spec_library"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic[0] if traffic else None,
                         "traffic_source": traffic[1] if traffic else None,
                         "kernel": ("k_spec<0, 0> (system-specialised rollout, team %d)" if r["specialised"] else "k_run<%d, 0> (generic rollout)") % r["info"]["team"],
                         "kernel_avg_ms": 1e3 * avg_kernel_s, "launches": r["launches"],
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "path is fp64-VALU/LDS-latency bound, not HBM bound (SURVEY.md §8d)"},
        }
        for mode in modes[1:]:     # the other scaling mode of the same run
            r2, gb2, bl2 = results[mode]
            out["%s_scaling" % mode] = {"value": float(gb2) * N * args.steps / r2["elapsed"], "unit": "DEL-steps/s",
                                        "global_batch": gb2, "batch_per_gpu": bl2, "ms_per_step": 1e3 * r2["elapsed"] / args.steps,
                                        "kernel_avg_ms": r2["kernel_ms"] / max(r2["launches"], 1),
                                        "note": "BASELINE config 3 as written: 8192 rollouts in total" if mode == "strong" else "fixed work per GPU"}
        fp64 = pmc_fp64(b_local, N, "Puppet(string_constraints=True)") if args.system == "puppet" else None
        if args.system == "puppet":   # secondary figure: the path is compute/latency bound, so also say how far from the fp64 peak
            ceiling, ceiling_src = fp64_fma_ceiling()
            out["fp64"] = None if not fp64 else {
                "estimated_tflops": fp64[0] / avg_kernel_s / 1e12, "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                "measured_v_fma_f64_ceiling_tflops": ceiling, "ceiling_source": ceiling_src,   # tools/micro/mfma_f64_rate.hip, two waves per SIMD
                "frac": fp64[0] / avg_kernel_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                "flop_per_del_step": fp64[0] / (b_local * N), "f64_wave_instructions_per_del_step": fp64[1] / (b_local * N), "source": fp64[2]}
            # The composite form of the Newton matrix (round 4) does the same step with 38 % fewer flops, so the counter-based rate fell while the
            # throughput rose; for a like-for-like reading against earlier rounds: the rate at which this kernel gets through the WORK of the
            # previous formulation (its counter-measured flop per DEL step, profiles/r04_fp64_pair_loop.json)
            ref = os.path.join(ROOT, "profiles", "r04_fp64_pair_loop.json")
            if out["fp64"] and os.path.exists(ref):
                try:
                    d = json.load(open(ref))
                    per_step = d["estimated_fp64_flop_per_launch"] / float(d["global_batch"] * d["rollout_steps"])
                    out["fp64"]["pair_loop_formulation"] = {
                        "flop_per_del_step": per_step, "equivalent_tflops": per_step * b_local * N / avg_kernel_s / 1e12,
                        "equivalent_frac": per_step * b_local * N / avg_kernel_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS, "source": os.path.relpath(ref, ROOT),
                        "note": "flops the (body, item, item) pair loop spends on the same DEL step / this kernel's time: comparable with the fp64.frac of rounds 1-3"}
                except Exception:
                    pass
        if discopt is not None:
            out["discopt"] = discopt
        if want_cpu:
            out["cpu_baseline"] = cpu_baseline(system, Q0, K, dt)
            out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        missing = [k for k in expected_line_keys(args, world) if k not in out]
        if missing:       # the measured line is printed whatever the self-check says; the ranks still meet at the barrier below
            out["line_self_check"] = "missing keys: %s" % missing
            sys.stderr.write("bench.py: line lacks %s\n" % missing)
        print(json.dumps(out))
        sys.stdout.flush()
    else:
        missing = []
    if comm is not None:
        try:
            comm.barrier()
        finally:
            comm.close()
    if missing:
        sys.exit(3)


if __name__ == "__main__":
    main()
