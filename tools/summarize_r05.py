#!/usr/bin/env python3
"""Copy / condense the round-5 measurements collected by tools/collect_r05.sh (gpurun_out/r05) into profiles/r05_*."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r05")
DST = os.path.join(ROOT, "profiles")
ROLLOUT = ("k_spec<0>", "k_spec<0,", "k_spec<(int)0", "k_run<64, 0")


def load(name):
    return json.load(open(os.path.join(SRC, name)))


def mean_counter(path, kernels=ROLLOUT):
    acc = {}
    for r in csv.DictReader(open(path)):
        if any(k in r["Kernel_Name"] for k in kernels):
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


for a, b in (("bench.json", "r05_bench.json"), ("bench_generic.json", "r05_bench_generic_kernel.json"),
             ("kernel_stats_rollout.csv", "r05_kernel_stats.csv"), ("kernel_stats.csv", "r05_kernel_stats_with_discopt.csv"),
             ("kernel_stats_discopt.csv", "r05_kernel_stats_discopt.csv"), ("pmc_fetch.csv", "r05_pmc_fetch_rollout.csv"),
             ("pmc_write.csv", "r05_pmc_write_rollout.csv"), ("mfma_f64_rate.txt", "r05_mfma_f64_rate.txt")):
    shutil.copy(os.path.join(SRC, a), os.path.join(DST, b))
bench = load("bench.json")
fetch, nf = mean_counter(os.path.join(SRC, "pmc_fetch.csv"))
write, nw = mean_counter(os.path.join(SRC, "pmc_write.csv"))
traffic = {
    "kernel": "k_spec<0, 0> (system-specialised rollout kernel)", "workload": bench["config"]["workload"],
    "global_batch": bench["config"]["global_batch"], "rollout_steps": bench["config"]["rollout_steps"],
    "FETCH_SIZE_kB_raw": fetch["FETCH_SIZE"], "WRITE_SIZE_kB": write["WRITE_SIZE"], "dispatches_averaged": [nf["FETCH_SIZE"], nw["WRITE_SIZE"]],
    "read_bytes_corrected": 2.0 * fetch["FETCH_SIZE"] * 1024.0, "write_bytes": write["WRITE_SIZE"] * 1024.0,
    "hbm_bytes_per_launch": 2.0 * fetch["FETCH_SIZE"] * 1024.0 + write["WRITE_SIZE"] * 1024.0,
    "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE exact; separate --pmc passes",
}
json.dump(traffic, open(os.path.join(DST, "r05_traffic.json"), "w"), indent=1)
c = load("fp64/fp64.json")
lanes = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
f64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_TRANS_F64"]
flop = (2 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_ADD_F64"]) * 64.0 * lanes
json.dump({"kernel": "k_spec<0, 0> (rollout)", "workload": bench["config"]["workload"], "global_batch": bench["config"]["global_batch"],
           "rollout_steps": bench["config"]["rollout_steps"], "counters_per_launch": c,
           "valu_f64_wave_instructions_per_launch": f64, "valu_wave_instructions_per_launch": c["SQ_INSTS_VALU"],
           "mean_active_lane_fraction": lanes, "estimated_fp64_flop_per_launch": flop,
           "note": "flop = (2*FMA + MUL + ADD) wave-instructions x 64 lanes x mean active-lane fraction of all VALU instructions; separate rocprofv3 --pmc pass (tools/collect_fp64.sh)"},
          open(os.path.join(DST, "r05_fp64.json"), "w"), indent=1)
sq = {}
for i in (1, 2, 3):
    m, _ = mean_counter(os.path.join(SRC, "sq", "p%d.csv" % i))
    sq.update(m)
wc = sq["SQ_WAVE_CYCLES"]
sq_out = {"kernel": "k_spec<0, 0> (rollout), B=8192 x 200 steps", "counters_per_launch": sq,
          "derived": {"valu_busy_fraction_of_simd_time": 4.0 * sq["SQ_ACTIVE_INST_VALU"] / (4.0 * sq["SQ_BUSY_CYCLES"] * 4) if False else None,
                      "wave_waiting_fraction (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": sq["SQ_WAIT_ANY"] / wc,
                      "wave_issue_stalled_fraction (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)": sq["SQ_WAIT_INST_ANY"] / wc,
                      "valu_instructions_per_del_step": sq["SQ_INSTS_VALU"] / (8192.0 * 200), "salu_per_del_step": sq["SQ_INSTS_SALU"] / (8192.0 * 200),
                      "lds_instructions_per_del_step": sq["SQ_INSTS_LDS"] / (8192.0 * 200),
                      "lanes_active_per_valu_instruction": sq["SQ_THREAD_CYCLES_VALU"] / (64.0 * sq["SQ_ACTIVE_INST_VALU"]),
                      "lds_bank_conflict_cycles_fraction": sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"]}}
sq_out["derived"].pop("valu_busy_fraction_of_simd_time")
json.dump(sq_out, open(os.path.join(DST, "r05_sq_counters.json"), "w"), indent=1)
sweep = {}
for s in (32, 64, 128, 256):
    d = load("discopt_%d.json" % s)
    sweep[str(s)] = {k: d[k] for k in ("value", "s_per_batched_quasi_step", "s_per_batched_newton_step", "stage_seconds", "armijo_failures", "config")}
    seq = os.path.join(SRC, "discopt_%d_unpipelined.json" % s)      # the same with the Newton step's sweeps one after the other (TREPAMD_NEWTON_PIPELINE=0: round 4's schedule)
    if os.path.exists(seq):
        try:
            q = json.load(open(seq))
            sweep[str(s)]["newton_step_not_pipelined"] = {k: q[k] for k in ("value", "s_per_batched_quasi_step", "s_per_batched_newton_step", "stage_seconds")}
        except Exception:
            pass
json.dump({"what": "bench_discopt.py --horizon 1000 --quasi 1 --newton 1 --stages on ONE MI355X at 32 / 64 / 128 / 256 seeds: what each of 8 / 4 / 2 / 1 GPUs holds when BASELINE config 4 (256 seeds) is sharded",
           "seeds": sweep}, open(os.path.join(DST, "r05_seed_sweep.json"), "w"), indent=1)
json.dump({"specialised": load("bench_derivs.json"), "generic": load("bench_derivs_generic.json")}, open(os.path.join(DST, "r05_derivs.json"), "w"), indent=1)
json.dump({k: load("bench_%s.json" % k) for k in ("cart", "scissor", "puppet-basic")}, open(os.path.join(DST, "r05_secondary.json"), "w"), indent=1)
json.dump({"mfma": load("lq_mfma.json"), "note": "us per Riccati (lqr) / affine (lq) step at nX = 80, nU = 18; *_dsystem: tg_lq_problem::ds_* set (k_tv_lq_ds), the others the dense kernel k_tv_lq_mfma on the same matrices"},
          open(os.path.join(DST, "r05_lq_sweep.json"), "w"), indent=1)
print("bench: %.4g steps/s, kernel %.2f ms, discopt %.1f it/s" % (bench["value"], bench["roofline"]["kernel_avg_ms"], bench["discopt"]["iters_per_s"]))
print("traffic %.3f GB per launch (algorithmic %.3f GB)" % (traffic["hbm_bytes_per_launch"] / 1e9, bench["roofline"]["algorithmic_bytes_per_launch"] / 1e9))
print("fp64: %.3g flop per launch -> %.2f TFLOP/s" % (flop, flop / (bench["roofline"]["kernel_avg_ms"] * 1e-3) / 1e12))
print(json.dumps(sq_out["derived"], indent=1))
for s in sweep:
    print(s, round(sweep[s]["value"], 1), sweep[s]["s_per_batched_quasi_step"], sweep[s]["s_per_batched_newton_step"])
pd = os.path.join(SRC, "prof_derivs")
if os.path.exists(os.path.join(pd, "deriv2.txt")):
    with open(os.path.join(DST, "r05_phase_profile_derivs.txt"), "w") as fh:
        fh.write("# specialised derivative kernels with helper waves, diagnostic -DTG_PROFILE build: cycles of WAVE 0 of trajectory 0\n"
                 "# (tools/gpu_prof_derivs.sh; a first-round trajectory: cold caches inflate the first phases)\n")
        fh.write(open(os.path.join(pd, "deriv2.txt")).read())
        fh.write(open(os.path.join(pd, "deriv1.txt")).read())
cls = {}
for i in (1, 2):
    pth = os.path.join(SRC, "pmc_cls%d.csv" % i)
    if os.path.exists(pth):
        m, _ = mean_counter(pth)
        cls.update(m)
if cls:
    per = {k: v / (8192.0 * 200) for k, v in cls.items()}
    json.dump({"kernel": "k_spec<0, 0> (rollout), B=8192 x 200 steps", "wave_instructions_per_del_step": per,
               "note": "dynamic instruction classes (SQ counters, two PMC passes); VALU minus (fp64 of rNN_fp64.json + INT32 + INT64 + CVT + F32) = moves, selects, lane reads, DPP"},
              open(os.path.join(DST, "r05_valu_classes.json"), "w"), indent=1)
for a, b in (("gj_bench.txt", "r05_gj_bench.txt"), ("bbd_bench.txt", "r05_bbd_bench.txt"), ("phase_profile_lq.txt", "r05_phase_profile_lq.txt"), ("step_latency.json", "r05_step_latency.json"), ("phase_profile.txt", "r05_phase_profile.txt")):
    if os.path.exists(os.path.join(SRC, a)):
        shutil.copy(os.path.join(SRC, a), os.path.join(DST, b))
