"""Executed-work calibration of k_step_bdf1<32,false> from rocprofv3 PMC passes (what bench.py's EXEC table holds).

    python tools/roofline_from_pmc.py gpurun_out/<tag>  [out.json]        (out.json: profiles/roofline_calibration.json, what bench.py loads)

Inputs (written by tools/gpu_session.sh): pmc_f64/ and pmc_f64_tol3/ = counter_collection.csv of the bench command at two Newton
tolerances (different iterations-per-step mixes) with SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64, SQ_INSTS_VALU_MFMA_MOPS_F64,
SQ_INSTS_VALU, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, plus the bench JSON line of each pass (measured iteration / halving counts).
Model: counter(launch) = front_evals * FRONT + newton_iters * NEWTON (+ nothing else: q/qdot load/store is ~10 instructions per
step), with front_evals = rollout-steps + newton_iters + ls_halvings.  Two launches (the 100-step timed ones) give the 2x2
system per counter; the 10-step warm-up launches of the same passes are the cross-check.
flops = 64 x (ADD_F64 + MUL_F64 + 2 FMA_F64) + 512 x MFMA_MOPS_F64  (wave-instructions x 64 lanes; one MOP = 512 flops)."""
import csv
import glob
import json
import sys

import numpy as np

COUNTERS = ("SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
            "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")


def read_pass(root, name):
    f = glob.glob("%s/pmc_%s/*/*counter_collection.csv" % (root, name))[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_step_bdf1" not in r["Kernel_Name"]:      # (k_step_bdf1<32,false,false,true>, the FULLCHAIN instantiation, for the 32-chain)
            continue
        per.setdefault(int(r["Dispatch_Id"]), {}).setdefault(r["Counter_Name"], 0.0)
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    ids = sorted(per)
    line = json.loads([ln for ln in open("%s/pmc_%s.json" % (root, name)) if ln.startswith("{")][0])
    return per[ids[0]], per[ids[-1]], line          # warm-up launch, timed launch, bench JSON


def flops(c):
    return 64.0 * (c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + 2.0 * c["SQ_INSTS_VALU_FMA_F64"]) + 512.0 * c["SQ_INSTS_VALU_MFMA_MOPS_F64"]


def main():
    root = sys.argv[1]
    rows, rhs = [], {k: [] for k in COUNTERS + ("flops",)}
    info = []
    for name in ("f64", "f64_tol3"):
        warm, timed, line = read_pass(root, name)
        B, K = line["config"]["batch_per_gpu"], line["steps"]
        r = line["roofline"]
        iters = r["newton_iters"]
        halv = r["ls_halvings_per_step"] * B * K
        # MFMA_MOPS is exact per iteration (15 MFMAs x 4 MOPS): use it to recover the integer iteration count of the profiled launch
        iters_pmc = timed["SQ_INSTS_VALU_MFMA_MOPS_F64"] / 60.0
        fronts = B * K + iters_pmc + halv
        rows.append([fronts, iters_pmc])
        for k in COUNTERS:
            rhs[k].append(timed[k])
        rhs["flops"].append(flops(timed))
        info.append({"pass": name, "tol": line["config"]["newton_tol"], "newton_iters_bench": iters, "newton_iters_from_mfma_mops": iters_pmc,
                     "ls_halvings": halv, "front_evals": fronts, "kernel_ms_profiled_run": r["kernel_ms"], "flops_timed_launch": flops(timed),
                     "warmup_launch_flops": flops(warm)})
    A = np.array(rows)
    out = {"model": "counter = front_evals * FRONT + newton_iters * NEWTON, per wavefront-instruction totals over the launch", "passes": info,
           "condition_number": float(np.linalg.cond(A)), "per_wave": {}}
    for k, v in rhs.items():
        x = np.linalg.solve(A, np.array(v))
        out["per_wave"][k] = {"front": float(x[0]), "newton": float(x[1])}
    t = info[0]
    sec = t["kernel_ms_profiled_run"] * 1e-3
    out["timed_launch"] = {
        "executed_tflops": t["flops_timed_launch"] / sec / 1e12, "frac_of_78.6": t["flops_timed_launch"] / sec / 78.6e12,
        "valu_insts_per_wave": rhs["SQ_INSTS_VALU"][0] / 1024.0, "wave_cycles_per_wave_counter_units": rhs["SQ_WAVE_CYCLES"][0] / 1024.0,
        "valu_per_newton_iter_incl_front": rhs["SQ_INSTS_VALU"][0] / t["newton_iters_from_mfma_mops"]}
    # HBM bytes per launch (separate FETCH_SIZE / WRITE_SIZE passes, KB as rocprofv3 reports them), the fingerprint of the kernel the
    # library of this session was built with, and where the numbers come from
    import os

    def kb(name, counter):
        f = glob.glob("%s/pmc_%s/*/*counter_collection.csv" % (root, name))
        if not f:
            return None
        per = {}
        for r in csv.DictReader(open(f[0])):
            if "k_step_bdf1" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                per[int(r["Dispatch_Id"])] = per.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
        return per[max(per)] if per else None
    fe, wr = kb("fetch", "FETCH_SIZE"), kb("write", "WRITE_SIZE")
    if fe is not None and wr is not None:
        out["hbm_kb_per_launch"] = {"fetch": fe, "write": wr, "fetch_k20": kb("fetch_k20", "FETCH_SIZE"), "write_k20": kb("write_k20", "WRITE_SIZE")}
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out["fingerprint"] = json.load(open(os.path.join(here, "redmax_amd", "kernel_fingerprint.json")))
    out["source"] = "tools/roofline_from_pmc.py %s" % root
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


main()
