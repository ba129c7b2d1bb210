#!/usr/bin/env python3
"""fp64 rooflines of the kernels other than the rollout from tools/collect_r05_kernels.sh's passes (gpurun_out/r05/kern_*): per kernel the
fp64 VALU and matrix-core flop per dispatch (SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 x 64 lanes x the average lane activity of the kernel's VALU
instructions; SQ_INSTS_VALU_MFMA_MOPS_F64 x 512), its average duration (rocprofv3 --kernel-trace --stats of the same command), the rates
against the 78.6 TFLOP/s fp64 vector and matrix peaks of the part, VALU lane activity, LDS bank-conflict fraction, wave wait fractions.
    python tools/summarize_r05_kernels.py  ->  profiles/r05_fp64_derivs.json, profiles/r05_fp64_lq.json"""
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r05")
PEAK = 78.6e12      # MI355X fp64, vector and matrix alike (datasheet)
CUS = 256
NAMES = {"k_spec<3, 0>": "deriv1 -> A, B (k_spec<3, 0>, two waves per trajectory)", "k_spec<4, 0>": "deriv2z (k_spec<4, 0>, two waves per trajectory)",
         "k_run<64, 6, false>": "dynamics_deriv1 (k_run<64, 6>, generic)", "k_run<64, 5, false>": "dynamics (k_run<64, 5>, generic)",
         "k_tv_lq_ds<5, 20>": "LQ / Riccati sweep (k_tv_lq_ds<5, 20>, one workgroup per seed)", "k_tangent_rows<20, 6, 10>": "tangent rollout (k_tangent_rows<20, 6, 10>)",
         "k_cost_mfma<5>": "quadratic cost (k_cost_mfma<5>)", "k_spec<0, 0>": "rollout launches of this workload (k_spec<0, 0>)"}


def short(name):
    for k in NAMES:
        if k in name:
            return k
    return None


def load(wl):
    p1 = json.load(open(os.path.join(SRC, "kern_%s_p1.json" % wl)))
    p2 = json.load(open(os.path.join(SRC, "kern_%s_p2.json" % wl)))
    stats = {}
    for r in csv.DictReader(open(os.path.join(SRC, "kern_%s_stats.csv" % wl))):
        stats[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]))
    out = {}
    for name, c in p1.items():
        k = short(name)
        if not k:
            continue
        st = next((v for n, v in stats.items() if k in n), None)
        c2 = next((v for n, v in p2.items() if k in n), {})
        if not st:
            continue
        lanes = c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_ACTIVE_INST_VALU"] * 64.0, 1.0)
        f64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_TRANS_F64"]
        valu_flop = (c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + 2.0 * c["SQ_INSTS_VALU_FMA_F64"]) * 64.0 * lanes
        mfma_flop = c["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0
        t = st[1] * 1e-9
        e = {"kernel": NAMES[k], "dispatches_timed": st[0], "dispatches_counted": c["dispatches"], "avg_duration_ms": st[1] * 1e-6,
             "valu_wave_instructions": c["SQ_INSTS_VALU"], "fp64_valu_wave_instructions": f64, "fp64_share_of_valu": f64 / max(c["SQ_INSTS_VALU"], 1.0),
             "lanes_active_per_valu_instruction": lanes,
             "fp64_valu_flop": valu_flop, "fp64_matrix_flop": mfma_flop,
             "valu_tflops": valu_flop / t / 1e12, "matrix_tflops": mfma_flop / t / 1e12,
             "frac_of_fp64_vector_peak": valu_flop / t / PEAK, "frac_of_fp64_matrix_peak": mfma_flop / t / PEAK,
             "frac_of_fp64_peak_both": (valu_flop + mfma_flop) / t / PEAK}
        if c2:
            e["lds_bank_conflict_fraction"] = c2.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c2.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)
            wc = max(c2.get("SQ_WAVE_CYCLES", 0.0), 1.0)
            e["wave_waiting_fraction"] = c2.get("SQ_WAIT_ANY", 0.0) / wc
            e["wave_issue_stalled_fraction"] = c2.get("SQ_WAIT_INST_ANY", 0.0) / wc
            e["matrix_core_busy_cycles"] = c2.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        out[k] = e
    return out


def main():
    d = load("derivs")
    B = 65536
    for k in ("k_spec<3, 0>", "k_spec<4, 0>", "k_run<64, 6, false>", "k_run<64, 5, false>"):
        if k in d:
            d[k]["units_per_s"] = B / (d[k]["avg_duration_ms"] * 1e-3)
            d[k]["flop_per_unit"] = (d[k]["fp64_valu_flop"] + d[k]["fp64_matrix_flop"]) / B
    json.dump({"what": "tools/bench_derivs.py --batch 65536 (puppet) under rocprofv3: one --kernel-trace --stats pass and two --pmc passes (tools/collect_r05_kernels.sh); "
                       "fp64 peaks of the part: 78.6 TFLOP/s vector, 78.6 TFLOP/s matrix", "kernels": d},
              open(os.path.join(ROOT, "profiles", "r05_fp64_derivs.json"), "w"), indent=1)
    q = load("discopt")
    for k, e in q.items():
        if k.startswith("k_tv_lq") or k.startswith("k_tangent"):      # one workgroup per seed: against the CUs it occupies
            e["per_cu"] = {"seeds": 32, "note": "32 seeds -> 32 of 256 CUs busy", "matrix_gflops_per_cu": e["fp64_matrix_flop"] / (e["avg_duration_ms"] * 1e-3) / 32 / 1e9,
                           "valu_gflops_per_cu": e["fp64_valu_flop"] / (e["avg_duration_ms"] * 1e-3) / 32 / 1e9, "cu_fp64_peak_gflops": PEAK / CUS / 1e9}
            e["per_cu"]["frac_of_a_cu_matrix_peak"] = e["per_cu"]["matrix_gflops_per_cu"] / e["per_cu"]["cu_fp64_peak_gflops"]
            e["per_cu"]["frac_of_a_cu_vector_peak"] = e["per_cu"]["valu_gflops_per_cu"] / e["per_cu"]["cu_fp64_peak_gflops"]
    json.dump({"what": "bench_discopt.py --seeds 32 --horizon 1000 --quasi 1 --newton 1 (puppet, config 4's per-GPU shard) under rocprofv3, passes as in r05_fp64_derivs.json; "
                       "the sweeps run one workgroup per seed, so their rate is also given per occupied CU", "kernels": q},
              open(os.path.join(ROOT, "profiles", "r05_fp64_lq.json"), "w"), indent=1)
    for name, t in (("derivs", d), ("discopt", q)):
        print(name)
        for k, e in t.items():
            print("  %-28s %8.3f ms  VALU %6.2f TF (%.3f)  matrix %6.2f TF (%.3f)  fp64 share %.2f  lanes %.2f  bank conflicts %.2f  wait %.2f  stall %.2f" %
                  (k, e["avg_duration_ms"], e["valu_tflops"], e["frac_of_fp64_vector_peak"], e["matrix_tflops"], e["frac_of_fp64_matrix_peak"], e["fp64_share_of_valu"],
                   e["lanes_active_per_valu_instruction"], e.get("lds_bank_conflict_fraction", -1), e.get("wave_waiting_fraction", -1), e.get("wave_issue_stalled_fraction", -1)))
            if "per_cu" in e:
                print("      per occupied CU: matrix %.1f GFLOP/s (%.3f of a CU's peak), VALU %.1f (%.3f)" % (e["per_cu"]["matrix_gflops_per_cu"], e["per_cu"]["frac_of_a_cu_matrix_peak"], e["per_cu"]["valu_gflops_per_cu"], e["per_cu"]["frac_of_a_cu_vector_peak"]))


if __name__ == "__main__":
    main()
