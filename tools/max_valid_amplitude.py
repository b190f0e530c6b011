"""Which initial-state range does the REFERENCE ALGORITHM survive on the headline workload?   (CPU, literal oracle only)

BASELINE.md / SURVEY.md 8(d) proposed q ~ U(-pi/4, pi/4), qdot ~ U(-1, 1) for the 1024 x 100-step batch of the 32-link chain; on
those states the reference's own Newton (oracle/redmax_oracle.c, the literal restatement) prints "Newton diverged" within a few
steps, so the bench uses U(-a, a) with a = 0.1 for both.  This script FINDS the largest a instead of choosing it: it bisects
a in [lo, hi] with

    valid(a)  :=  no trajectory-step of the B x K batch ends in "Newton diverged" (driverRedMaxBDF1.m:118-121), and every step that
                  ends in "Newton did not converge" (:150-153) leaves |g| < GSTALL = 1e-6

(on plain doubles the reference's |g| < 1e-9 test is at the resolution of M ulp(q) for this 320 cm cgs chain: ~0.3 % of the steps
run to iterMax with |g| ~ 1e-9 .. 1e-8 at ANY amplitude - DESIGN.md 5 -; those are converged solutions for every purpose, a step that
stalls at |g| ~ 1e2 is not).  The batch is advanced in chunks of a few steps so that an invalid amplitude is given up at its first
"diverged".  States: redmax_amd.scenes.syntheticStates (seeded by global rollout index, rollout 0 = the deterministic q = 0.1 state).

    python tools/max_valid_amplitude.py --batch 1024 --steps 100 --lo 0.1 --hi 0.785 --iters 6 [--threads N] [--out file.json]

One valid probe of the full batch costs ~10^5 rollout-steps of the literal oracle (about 1.2 per second and core)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc                        # noqa: E402
from redmax_amd.scenes import sceneChain, syntheticStates   # noqa: E402

GSTALL = 1e-6


def probe(desc, nr, h, amp, batch, steps, chunk, threads, first=0):
    q, qd = syntheticStates(nr, batch, first=first, sq=amp, sv=amp)
    q = np.ascontiguousarray(q)
    qd = np.ascontiguousarray(qd)
    t0 = time.time()
    res = {"amp": amp, "batch": batch, "steps_done": 0, "diverged_steps": 0, "not_converged_steps": 0, "worst_exit_g": 0.0, "newton_iters": 0}
    k = 0
    while k < steps:
        c = min(chunk, steps - k)
        out = orc.batch_step_bdf1(desc, q, qd, h, c, nthreads=threads, counters=True)
        k += c
        res["steps_done"] = k
        res["diverged_steps"] += int(out["diverged"].sum())
        res["not_converged_steps"] += int((out["bad"] - out["diverged"]).sum())
        res["worst_exit_g"] = max(res["worst_exit_g"], float(out["worst_exit_g"].max()))
        res["newton_iters"] += int(out["newton_iters"].sum())
        if res["diverged_steps"] or res["worst_exit_g"] >= GSTALL or not np.isfinite(q).all():
            res["first_bad_rollouts"] = [int(i) + first for i in np.nonzero((out["diverged"] > 0) | (out["worst_exit_g"] >= GSTALL))[0][:8]]
            break
    res["valid"] = bool(res["diverged_steps"] == 0 and res["worst_exit_g"] < GSTALL and np.isfinite(q).all() and k >= steps)
    res["seconds"] = round(time.time() - t0, 1)
    res["max_abs_q_end"] = float(np.abs(q).max())
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--links", type=int, default=32)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--chunk", type=int, default=5)
    ap.add_argument("--lo", type=float, default=0.1)
    ap.add_argument("--hi", type=float, default=np.pi / 4)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--probe", type=float, nargs="*", help="only probe these amplitudes (no bisection)")
    ap.add_argument("--budget-s", type=float, default=1e9, help="stop bisecting when this much time is spent")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    orc.build()
    sc = sceneChain(a.links)
    sc.init()
    desc = sc.desc()
    orc.set_newton()                       # the reference's constants: tol 1e-9, dxMax 1e3, iterMax 10 nr, 20 halvings
    log = []
    t0 = time.time()

    def run(amp):
        r = probe(desc, sc.nr, sc.h, amp, a.batch, a.steps, a.chunk, a.threads)
        log.append(r)
        print(json.dumps(r), flush=True)
        return r["valid"]

    result = {"criterion": "no 'Newton diverged' step and every not-converged step ends with |g| < %g, literal oracle at the reference's "
                           "Newton constants, %d rollouts x %d BDF1 steps of the %d-link chain, h = %g" % (GSTALL, a.batch, a.steps, a.links, sc.h)}
    if a.probe:
        for amp in a.probe:
            run(amp)
    else:
        lo, hi = a.lo, a.hi
        lo_ok = run(lo)
        if not lo_ok:
            print("lower end %g is not valid: nothing to bisect" % lo)
        else:
            for _ in range(a.iters):
                if time.time() - t0 > a.budget_s:
                    break
                mid = round(0.5 * (lo + hi), 4)
                if run(mid):
                    lo = mid
                else:
                    hi = mid
            result["largest_valid_found"] = lo
            result["smallest_invalid_found"] = hi
    result["probes"] = log
    result["cores"] = a.threads or os.cpu_count()
    result["seconds"] = round(time.time() - t0, 1)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(result, f, indent=1)
    print(json.dumps({k: v for k, v in result.items() if k != "probes"}))


main()
