"""Executed-work calibration of every bench workload from rocprofv3 PMC passes (what bench.py's `roofline` objects are built from).

    python tools/roofline_from_pmc.py gpurun_out/<tag>  [out.json]        (out.json: profiles/roofline_calibration.json, what bench.py loads)

Inputs (written by tools/gpu_session.sh <tag> pmc): for every workload WL in chain, tree64, tree64x (1024 rollouts: the kernels that
read the constants from global memory), ground, adjoint, chain128 (one workgroup per tree) the directories pmc_WL_<pass>/ with the
counter_collection.csv of `python bench.py --workload ... --repeats 0 --no-side-legs --no-cpu-baseline` and pmc_WL_<pass>.json,
the bench line of that very run (measured evaluation / iteration counts).  Passes (separate runs: 8 SQ slots, FETCH_SIZE and
WRITE_SIZE do not fit one pass; --kernel-trace only, as the pool requires):
    f64     SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64  SQ_INSTS_VALU_MFMA_MOPS_F64  SQ_INSTS_VALU  SQ_WAVE_CYCLES  SQ_BUSY_CYCLES
    sq      SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES
    fetch   FETCH_SIZE        write   WRITE_SIZE
    f64_tol3  (chain, tree64, tree64x) the f64 pass at --tol 1e-3: another iterations-per-step mix

What is stored per workload:
  launch     the counter TOTALS of the timed launch (the last step call of the run: with --repeats 0 nothing follows it) together with
             its signature (steps, warm-up, batch, tol) and its measured Newton iteration / evaluation counts.  The workloads are
             deterministic, so a bench run with the same signature reproduces the iteration count exactly and bench.py uses the totals
             as they are (no model): flops = 64 x (ADD + MUL + 2 FMA) + 512 x MFMA_MOPS, every wave-wide instruction counted with all
             64 lanes.
  per_wave   where the instruction counts per stage are static (chain, tree64, tree64x): counter = front_evals * FRONT +
             newton_iters * NEWTON solved from the two tolerance passes (the solution comes out integral to 3 digits: the static
             instruction counts of the two code paths).  Valid for any --steps / --warmup of that workload.
  fingerprints  opcode hashes of the kernels the workload launched (redmax_amd/kernel_fingerprint.json of the library the session
             ran): bench.py refuses the entry when the built kernels differ.
"""
import csv
import glob
import json
import os
import sys

import numpy as np

F64 = ("SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
       "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")

# kernels of the timed step call, per workload: name fragments as rocprofv3 prints them (demangled)
KERNELS = {
    "chain": ("k_step_bdf1_pair32", "k_step_bdf1<32"), "tree64": ("k_step_bdf1<64",), "tree64x": ("k_step_bdf1<64",), "ground": ("k_ground32", "k_step_pair<"),
    "adjoint": ("k_adjoint_fwd<16", "k_adjoint_bwd<16"), "chain128": ("k_big_step",),
}


def read_pass(root, wl, name):
    """(counter totals of the timed call {counter: value}, kernel names of that call, bench JSON) or None"""
    files = glob.glob("%s/pmc_%s_%s/*/*counter_collection.csv" % (root, wl, name))
    jf = "%s/pmc_%s_%s.json" % (root, wl, name)
    if not files or not os.path.exists(jf):
        return None
    lines = [ln for ln in open(jf) if ln.startswith("{")]
    if not lines:
        return None
    per, names = {}, {}
    for r in csv.DictReader(open(files[0])):
        if not any(k in r["Kernel_Name"] for k in KERNELS[wl]):
            continue
        d = int(r["Dispatch_Id"])
        per.setdefault(d, {}).setdefault(r["Counter_Name"], 0.0)
        per[d][r["Counter_Name"]] += float(r["Counter_Value"])
        names[d] = r["Kernel_Name"]
    if not per:
        return None
    ids = sorted(per)
    # the timed call = the trailing dispatches with pairwise different kernels (chain: 1; ground: lean + contact kernel; adjoint: forward +
    # backward), as long as the same sequence of kernels also ended the call before it
    take = [ids[-1]]
    for d in reversed(ids[:-1]):
        if names[d] in [names[t] for t in take]:
            break
        take.append(d)
    tot = {}
    for d in take:
        for k, v in per[d].items():
            tot[k] = tot.get(k, 0.0) + v
    return tot, sorted(set(names[d] for d in take)), json.loads(lines[0])


def flops(c):
    return 64.0 * (c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + 2.0 * c["SQ_INSTS_VALU_FMA_F64"]) + 512.0 * c["SQ_INSTS_VALU_MFMA_MOPS_F64"]


def signature(line):
    c = line["config"]
    return {"steps": line["steps"], "warmup": line["warmup"], "batch": c["batch_per_gpu"], "tol": c.get("newton_tol"), "links": c.get("links")}


def counts(line):
    """(front evaluations, Newton iterations) of the timed launch as the bench line reports them"""
    r = line.get("roofline") or {}
    return r.get("front_evals"), r.get("newton_iters")


def main():
    root = sys.argv[1]
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fp_all = json.load(open(os.path.join(here, "redmax_amd", "kernel_fingerprint.json")))
    by_name = {v["name"]: (k, v) for k, v in fp_all.items()}
    out = {"schema": 2, "source": "tools/roofline_from_pmc.py %s" % root, "workloads": {}}
    if len(sys.argv) > 2 and os.path.exists(sys.argv[2]):      # a session that profiled some of the workloads: the others keep their entries
        try:
            old = json.load(open(sys.argv[2]))
            out["workloads"] = dict(old.get("workloads") or {})
            out["source"] = "%s; %s" % (old.get("source", ""), out["source"])
        except (OSError, ValueError):
            pass
    for wl in KERNELS:
        p = read_pass(root, wl, "f64")
        if p is None:
            continue
        tot, knames, line = p
        fronts, iters = counts(line)
        ent = {"kernels": knames, "signature": signature(line), "front_evals": fronts, "newton_iters": iters,
               "kernel_ms_profiled_run": (line.get("roofline") or {}).get("kernel_ms"),
               "launch": {k: tot[k] for k in F64 if k in tot}}
        ent["launch"]["flops"] = flops(tot)
        sq = read_pass(root, wl, "sq")
        if sq is not None:
            ent["launch_sq"] = sq[0]
        for nm, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            q = read_pass(root, wl, nm)
            if q is not None and ctr in q[0]:
                ent.setdefault("hbm_kb_per_launch", {})[nm] = q[0][ctr]
        # fingerprints of the kernels this workload launched: demangled rocprof names -> symbols of the library
        fps = {}

        def base(nm):      # "void (anonymous namespace)::k<1, true>(args)" -> "k<1, true>"
            return nm.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
        for kn in knames:
            for n, (s, v) in by_name.items():
                if base(n) == base(kn):
                    fps[s] = v["opcode_sha16"]
        if len(fps) != len(knames):
            print("WARNING: %s: %d kernels launched, %d matched in kernel_fingerprint.json" % (wl, len(knames), len(fps)))
        ent["fingerprints"] = fps
        p3 = read_pass(root, wl, "f64_tol3")
        if p3 is not None and fronts and iters:
            tot3, _, line3 = p3
            f3, i3 = counts(line3)
            A = np.array([[fronts, iters], [f3, i3]], dtype=float)
            if "pair32" in " ".join(knames):
                # the two-point kernel: every Newton iteration executes exactly one evaluation of the front (its line-search trial), so
                # front_evals - newton_iters (rejected trials + the first evaluation of each rollout's launch) is the second regressor:
                # counts = (front_evals - newton_iters) x FRONT + newton_iters x NEWTON, NEWTON = a whole iteration INCLUDING its front
                A = np.array([[fronts - iters, iters], [f3 - i3, i3]], dtype=float)
                ent["per_wave_basis"] = "fronts_beyond_iters"
            ent["model_condition_number"] = float(np.linalg.cond(A))
            ent["model_passes"] = [{"tol": signature(line)["tol"], "front_evals": fronts, "newton_iters": iters},
                                   {"tol": signature(line3)["tol"], "front_evals": f3, "newton_iters": i3}]
            pw = {}
            for k in F64:
                x = np.linalg.solve(A, np.array([tot[k], tot3[k]]))
                pw[k] = {"front": float(x[0]), "newton": float(x[1])}
            x = np.linalg.solve(A, np.array([flops(tot), flops(tot3)]))
            pw["flops"] = {"front": float(x[0]), "newton": float(x[1])}
            ent["per_wave"] = pw
        out["workloads"][wl] = ent
        ms = ent["kernel_ms_profiled_run"]
        ent["kernels"] = [base(k) for k in knames]
        print("%-9s %-40s flops/launch %.4g  VALU/wave %.0f%s" % (wl, ",".join(ent["kernels"])[:40], ent["launch"]["flops"],
              tot["SQ_INSTS_VALU"] / max(signature(line)["batch"], 1),
              ("  executed %.2f TF in the profiled run" % (ent["launch"]["flops"] / (ms * 1e-3) / 1e12)) if ms else ""))
    print(json.dumps({k: v.get("per_wave", {}).get("SQ_INSTS_VALU") for k, v in out["workloads"].items()}, indent=1))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


main()
