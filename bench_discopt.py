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


def sum_over_ranks(value, dist, torch):
    t = torch.tensor([float(value)], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def problem(seeds, N, dt, device=0):
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


def run_batched(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = torch = None
    if world > 1:   # torch first: one HIP runtime in the process
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    import trep_amd
    from trep_amd import discopt, _lib, distributed
    dt, N = 0.01, args.horizon
    system, Xd, Ud, Xi, Ui, Qc, Rc = problem(args.seeds, N, dt, device=local_rank)
    lo, hi = distributed.shard_bounds(args.seeds, rank, world)     # this rank's seeds
    Xd, Ud, Xi, Ui = Xd[lo:hi], Ud[lo:hi], Xi[lo:hi], Ui[lo:hi]
    S = hi - lo
    dsys = discopt.DSystem(trep_amd.MidpointVI(system, device=local_rank), dt * np.arange(N + 1))
    opt = discopt.BatchDOptimizer(dsys, Xd, Ud, Qc, Rc, device=local_rank, armijo_chunk=args.armijo_chunk,
                                  predictor=args.predictor)
    L = _lib.lib()
    methods = ["quasi"] * args.quasi + ["newton"] * args.newton
    opt.set_trajectories(Xi, Ui)
    opt.step(methods[0])                      # warm-up (allocations, code objects), not timed
    opt.set_trajectories(Xi, Ui)
    L.tg_device_synchronize(local_rank)
    if dist is not None:
        dist.barrier()
    per = {"quasi": [], "newton": []}
    costs = []
    n_failed = 0
    t0 = time.perf_counter()
    for m in methods:
        ts = time.perf_counter()
        r = opt.step(m)
        L.tg_device_synchronize(local_rank)
        per[m].append(time.perf_counter() - ts)
        costs.append([float(np.nanmean(r.cost0)), float(np.nanmean(r.cost1))])
        n_failed += int(r.failed.sum())
    elapsed = time.perf_counter() - t0
    final_cost = r.cost1
    if dist is not None:                      # every rank sees every seed's cost (what a supervisor would act on)
        dist.barrier()
        elapsed = distributed.max_over_ranks(elapsed, device="cuda")
        final_cost = distributed.all_gather_rows(torch.from_numpy(np.nan_to_num(r.cost1)[:, None]).cuda()).cpu().numpy()[:, 0]
        n_failed = int(sum_over_ranks(n_failed, dist, torch))
    stages = None
    if args.stages and world == 1:             # one more Newton step, synchronising after every stage
        stages = {}

        def timed(name, fn, *a):
            ts = time.perf_counter()
            out = fn(*a)
            L.tg_device_synchronize(local_rank)
            stages[name] = stages.get(name, 0.0) + time.perf_counter() - ts
            return out
        timed("linearize (S*N DEL solves + deriv1 -> A,B)", opt.linearize)
        timed("projection gain (Riccati)", opt.projection_gain)
        timed("cost + gradients", opt.gradients_and_cost)
        timed("newton curvature (adjoint + S*N deriv2z)", opt.newton_curvature, None)
        timed("LQ sweep + tangent rollout", lambda: (opt._lq(None, opt.Q, opt.Qf, opt.R, opt.HZ, True, opt.K, opt.C),
                                                      opt.descent_direction(None, "quasi")))
        timed("armijo round 1 (S*%d projections + costs)" % opt.M, opt.armijo_chunk, 0)
    iters = args.seeds * len(methods)
    out = {
        "metric": "discopt iterations/s (puppet ~40-DOF, fp64)", "value": iters / elapsed, "unit": "iters/s",
        "n_gpus": world, "dtype": "f64", "data": "synthetic", "scaling": "strong",
        "config": {"workload": "puppet-optimization.py problem, nX=80 nU=18, N=%d, %d seeds batched on the device (%d per GPU), %d quasi + %d newton steps each"
                               % (N, args.seeds, S, args.quasi, args.newton), "armijo_chunk": opt.M,
                   "newton_initial_guess": args.predictor},
        "mean_final_cost_all_seeds": float(np.mean(final_cost)),
        "s_per_batched_quasi_step": float(np.mean(per["quasi"])) if per["quasi"] else None,
        "s_per_batched_newton_step": float(np.mean(per["newton"])) if per["newton"] else None,
        "mean_cost_before_after_per_step": costs,
        "armijo_failures": n_failed,   # seeds where the reference would raise ConvergenceError("Armijo Failed to Converge")
        "stage_seconds": stages,
        "reference_s_per_newton_step_N1000_one_seed": REFERENCE_NEWTON_ITER_S,
    }
    opt.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


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
    ap.add_argument("--predictor", choices=["reference", "extrapolate"], default="reference",
                    help="Newton initial guess of the Armijo projections (default: the reference's)")
    ap.add_argument("--sequential", action="store_true")
    args = ap.parse_args()
    (run_sequential if args.sequential else run_batched)(args)


if __name__ == "__main__":
    main()
