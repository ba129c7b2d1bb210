#!/usr/bin/env python3
"""bench_discopt.py -- secondary metric of BASELINE.json: discopt iterations/s on the ~40-DOF puppet.

Problem of examples/puppet-optimization.py (reference lines 20-24, 27-105, 127-185): the desired
trajectory is the puppet simulated with its four limb strings moving sinusoidally; the initial guess is
the same puppet with the strings held still; cost weights QD=100, QK=1, PD=1, VK=1, RHO=0.1.  One
"iteration" is one DOptimizer.step of one seed.  S seeds (perturbed initial poses, SURVEY section 8d
config 4) are optimised together by BatchDOptimizer, device resident: S*N-way linearisation, S Riccati
sweeps, [Newton: S adjoint sweeps + S*N z-contracted second derivatives], S LQ sweeps, S*M Armijo
projections per chunk.  `--sequential` runs the per-seed DOptimizer (host LQR) instead.

  python bench_discopt.py --seeds 256 --horizon 1000 --quasi 1 --newton 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench_discopt.py --seeds 256 ...        # seeds sharded over N GPUs, costs all-gathered
Prints one JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

REFERENCE_NEWTON_ITER_S = 35.0   # reference, puppet, N~1000: >= 35 s per Newton iteration


def problem(seeds, N, dt, device=0):
    """S puppet problems of examples/puppet-optimization.py: perturbed initial poses,
    desired trajectory = moving strings, initial guess = still strings, the script's cost weights."""
    import trep_amd
    from trep_amd import systems
    system = systems.puppet()
    nd = system.nQd
    Q0 = systems.puppet_initial_conditions(system, seeds, seed=20250 + 4)
    K_move = systems.puppet_string_schedule(system, Q0[:, nd:], N, dt)
    K_still = np.repeat(Q0[:, None, nd:], N, axis=1)
    sim = trep_amd.BatchMidpointVI(system, seeds, device=device)
    sim.initialize_from_state(0.0, Q0, np.zeros((seeds, nd)))
    Xd = sim.rollout(N, dt, None, K_move)
    sim.initialize_from_state(0.0, Q0, np.zeros((seeds, nd)))
    Xi = sim.rollout(N, dt, None, K_still)
    sim.close()
    wq = [100.0] * nd + [1.0] * system.nQk + [1.0] * nd + [1.0] * system.nQk
    return system, Xd, K_move, Xi, K_still, np.diag(wq), np.diag([0.1] * system.nQk)


def measure(seeds, horizon, quasi=1, newton=1, comm=None, device=0, armijo_chunk=None, predictor="reference", overlap="auto",
            stages=False):
    """Times `quasi` + `newton` batched DOptimizer steps of `seeds` puppet problems (sharded over the ranks of `comm`,
    a trep_amd.rccl.Communicator, if given) after one untimed warm-up step per method; returns the result dict (same on every
    rank).  Seeds whose Armijo search is exhausted (where the reference raises) do not count as iterations."""
    import trep_amd
    from trep_amd import discopt, _lib, distributed
    world = comm.world if comm is not None else 1
    rank = comm.rank if comm is not None else 0
    dt, N = 0.01, horizon
    system, Xd, Ud, Xi, Ui, Qc, Rc = problem(seeds, N, dt, device=device)
    lo, hi = distributed.shard_bounds(seeds, rank, world)     # this rank's seeds
    Xd, Ud, Xi, Ui = Xd[lo:hi], Ud[lo:hi], Xi[lo:hi], Ui[lo:hi]
    S = hi - lo
    dsys = discopt.DSystem(trep_amd.MidpointVI(system, device=device), dt * np.arange(N + 1))
    opt = discopt.BatchDOptimizer(dsys, Xd, Ud, Qc, Rc, device=device, armijo_chunk=armijo_chunk, predictor=predictor,
                                  overlap_sweeps={"auto": "auto", "on": True, "off": False}[overlap])
    L = _lib.lib()
    methods = ["quasi"] * quasi + ["newton"] * newton
    opt.set_trajectories(Xi, Ui)
    for m in dict.fromkeys(methods):          # warm-up, not timed: one step of every method used (a first Newton step allocates the
        opt.set_trajectories(Xi, Ui)          # 13 GB of z-contracted second derivatives: 0.36 s of hipMalloc, once per optimizer)
        opt.step(m)
    opt.set_trajectories(Xi, Ui)
    L.tg_device_synchronize(device)
    if comm is not None:
        comm.barrier()
    per = {"quasi": [], "newton": []}
    costs = []
    n_failed = 0
    good_iters = 0
    t0 = time.perf_counter()
    for m in methods:
        ts = time.perf_counter()
        r = opt.step(m)
        L.tg_device_synchronize(device)
        per[m].append(time.perf_counter() - ts)
        ok = ~r.failed
        costs.append([float(np.mean(r.cost0[ok])) if ok.any() else None, float(np.mean(r.cost1[ok])) if ok.any() else None])
        n_failed += int(r.failed.sum())
        good_iters += int(ok.sum())
    elapsed = time.perf_counter() - t0
    final_cost, final_ok = r.cost1, ~r.failed
    if comm is not None:                      # every rank sees every seed's cost (what a supervisor would act on)
        comm.barrier()
        elapsed = comm.max(elapsed)
        gathered = comm.all_gather_rows(np.stack([np.where(final_ok, r.cost1, 0.0), final_ok.astype(float)], 1), total_rows=seeds)
        final_cost, final_ok = gathered[:, 0], gathered[:, 1] > 0.5
        n_failed = int(round(comm.sum(n_failed)))
        good_iters = int(round(comm.sum(good_iters)))
        for m in per:
            per[m] = [comm.max(x) for x in per[m]]
    stage_s = None
    if stages and world == 1:             # one more Newton step, synchronising after every stage
        stage_s = {}

        def timed(name, fn, *a):
            ts = time.perf_counter()
            out = fn(*a)
            L.tg_device_synchronize(device)
            stage_s[name] = stage_s.get(name, 0.0) + time.perf_counter() - ts
            return out
        timed("linearize (S*N DEL solves + deriv1 -> A,B)", opt.linearize)
        timed("projection gain (Riccati)", opt.projection_gain)
        timed("cost + gradients", opt.gradients_and_cost)
        if opt.overlap:      # what a step really runs with S <= 128 seeds: the two independent sweeps on two streams
            timed("projection gain || quasi LQ sweep + tangent rollout (side by side, replaces the two alone)", opt.projection_gain_and_quasi_direction)
        if getattr(opt, "pipeline", False):      # ... and what a Newton step runs instead of the next two stages, one after the other
            timed("projection | quasi | second derivatives | Newton LQ + tangent, pipelined over chunks of the horizon (replaces the side-by-side stage and the next two)",
                  opt.projection_quasi_and_newton_model)
        timed("newton curvature (adjoint + S*N deriv2z)", opt.newton_curvature, None)
        timed("LQ sweep + tangent rollout", lambda: (opt._lq(None, opt.Q, opt.Qf, opt.R, opt.HZ, True, opt.K, opt.C),
                                                      opt.descent_direction(None, "quasi")))
        timed("armijo round 1 (S*%d projections + costs)" % opt.M, opt.armijo_chunk, 0)
    out = {
        "metric": "discopt iterations/s (puppet ~40-DOF, fp64)", "value": good_iters / elapsed, "unit": "iters/s",
        "iters_per_s": good_iters / elapsed, "seeds": seeds, "horizon": N,
        "n_gpus": world, "dtype": "f64", "data": "synthetic", "scaling": "strong",
        "config": {"workload": "puppet-optimization.py problem, nX=80 nU=18, N=%d, %d seeds batched on the device (%d per GPU), %d quasi + %d newton steps each"
                               % (N, seeds, S, quasi, newton), "armijo_chunk": opt.M,
                   "newton_initial_guess": predictor, "sweeps_side_by_side": bool(opt.overlap),
                   "newton_step_pipelined_over_horizon_chunks": bool(getattr(opt, "pipeline", False)), "pipeline_chunks": len(opt._chunks()) if getattr(opt, "pipeline", False) else 0},
        "seed_iterations_counted": good_iters, "elapsed_s": elapsed,
        "mean_final_cost_successful_seeds": float(np.mean(final_cost[final_ok])) if final_ok.any() else None,
        "s_per_batched_quasi_step": float(np.mean(per["quasi"])) if per["quasi"] else None,
        "s_per_batched_newton_step": float(np.mean(per["newton"])) if per["newton"] else None,
        "mean_cost_before_after_per_step": costs,
        "armijo_failures": n_failed,   # seeds where the reference would raise ConvergenceError("Armijo Failed to Converge")
        "stage_seconds": stage_s,
        "reference_s_per_newton_step_N1000_one_seed": REFERENCE_NEWTON_ITER_S,
    }
    opt.close()
    return out


def run_batched(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    comm = None
    if world > 1:   # torch-free: RCCL behind the C ABI (trep_amd/rccl.py)
        from trep_amd import rccl
        comm = rccl.Communicator.from_env()
    device = comm.device if comm is not None else 0
    out = measure(args.seeds, args.horizon, args.quasi, args.newton, comm=comm, device=device,
                  armijo_chunk=args.armijo_chunk, predictor=args.predictor, stages=args.stages, overlap=args.overlap)
    if comm is None or comm.rank == 0:
        print(json.dumps(out))
    if comm is not None:
        comm.barrier()
        comm.close()


def run_sequential(args):
    import trep_amd
    from trep_amd import discopt
    dt, N = 0.01, args.horizon
    system, Xd, Ud, Xi, Ui, Qc, Rc = problem(args.seeds, N, dt)
    dsys = discopt.DSystem(trep_amd.MidpointVI(system), dt * np.arange(N + 1))
    methods = ["quasi"] * args.quasi + ["newton"] * args.newton
    opt = discopt.DOptimizer(dsys, discopt.DCost(Xd[0], Ud[0], Qc, Rc))
    opt.step(0, Xi[0].copy(), Ui[0].copy(), "quasi")   # warm-up, not timed
    per = {"quasi": [], "newton": []}
    iters = 0
    t0 = time.perf_counter()
    for s in range(args.seeds):
        opt.cost = discopt.DCost(Xd[s], Ud[s], Qc, Rc)
        X, U = Xi[s].copy(), Ui[s].copy()
        for i, m in enumerate(methods):
            ts = time.perf_counter()
            (done, X, U, dcost0, cost1) = opt.step(i, X, U, m)
            per[m].append(time.perf_counter() - ts)
            iters += 1
            if done:
                break
    elapsed = time.perf_counter() - t0
    print(json.dumps({
        "metric": "discopt iterations/s (puppet ~40-DOF, fp64)", "value": iters / elapsed, "unit": "iters/s",
        "n_gpus": 1, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "puppet-optimization.py problem, nX=80 nU=18, N=%d, %d seeds one after the other (host LQR)" % (N, args.seeds)},
        "mean_s_per_quasi_step": float(np.mean(per["quasi"])) if per["quasi"] else None,
        "mean_s_per_newton_step": float(np.mean(per["newton"])) if per["newton"] else None,
        "reference_s_per_newton_step_N1000_one_seed": REFERENCE_NEWTON_ITER_S}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=32)
    ap.add_argument("--horizon", type=int, default=200, help="number of DEL steps N in the trajectory")
    ap.add_argument("--quasi", type=int, default=1)
    ap.add_argument("--newton", type=int, default=2)
    ap.add_argument("--armijo-chunk", type=int, default=None)
    ap.add_argument("--stages", action="store_true", help="also report per-stage times of one Newton step")
    ap.add_argument("--overlap", choices=["auto", "on", "off"], default="auto",
                    help="projection gain and quasi-Newton sweep side by side on two streams (auto: with <= 128 seeds per GPU)")
    ap.add_argument("--predictor", choices=["reference", "extrapolate"], default="reference",
                    help="Newton initial guess of the Armijo projections (default: the reference's)")
    ap.add_argument("--sequential", action="store_true")
    args = ap.parse_args()
    (run_sequential if args.sequential else run_batched)(args)


if __name__ == "__main__":
    main()
