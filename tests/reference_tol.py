"""Behaviour of the BDF1 rollout of BASELINE.json configs[1] at the reference's own Newton constant (tol = 1e-9,
driverRedMaxBDF1.m:95), per step, on every implementation available:

    python tests/reference_tol.py [--rollouts 64] [--steps 100] [--no-gpu] [--no-literal] [--json out.json]

gpu / gpu_plain                  the HIP library through the C ABI, one step per call: rmx_opts.compensated = 1 (default) / 0
tensor_free / tensor_free_plain  oracle/redmax_tensorfree.c (the algorithm the GPU executes, scalar C), the same two modes
literal                          oracle/redmax_oracle.c (the literal restatement of the .m files, O(n^3) tensor path)

Per implementation: Newton iterations and line-search halvings per trajectory-step, fraction of trajectory-steps whose Newton ended
"did not converge" / "diverged", and the same split over the first / second half of the rollout (the roundoff floor of |g| is
reached later in the rollout).  tests/test_gpu_reference_tol.py and tests/test_oracle_tensorfree.py assert the bands; run as a script
this prints the numbers behind them."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_stats(B, K, h=1e-2, tol=1e-9, gpu=True, literal=True, tensor_free=True, first=0, stride=1):
    from oracle import oracle as orc
    from redmax_amd import sceneChain, syntheticStates
    sc = sceneChain(32)
    sc.init()
    q = np.empty((B, 32))
    qd = np.empty((B, 32))
    for i in range(B):
        q[i], qd[i] = (a[0] for a in syntheticStates(32, 1, first=first + stride * i))
    desc = sc.desc()
    nthreads = os.cpu_count() or 1
    out = {}

    def pack(it, ls, bad, qf, qdf):
        it, ls, bad = np.asarray(it), np.asarray(ls), np.asarray(bad)
        half = K // 2
        return {"iters_per_step": float(it.mean()), "halvings_per_step": float(ls.mean()), "bad_frac": float((bad != 0).mean()),
                "bad_frac_first_half": float((bad[:half] != 0).mean()), "bad_frac_second_half": float((bad[half:] != 0).mean()),
                "rollouts_with_a_bad_step": int((bad != 0).any(axis=0).sum()), "iters": it, "halvings": ls, "bad": bad, "q": qf, "qd": qdf}

    for comp in ((1, 0) if gpu else ()):
        from redmax_amd import BatchSim
        sim = BatchSim(sc, batch=B)
        sim.opts.tol = tol
        sim.opts.compensated = comp
        sim.set_state(q, qd)
        it, ls, bad = (np.zeros((K, B), dtype=np.int64) for _ in range(3))
        for s in range(K):
            o = sim.step_bdf1(1, h=h, stats=True)
            it[s], ls[s], bad[s] = o["newton_iters"], o["ls_halvings"], o["status"] & 15
        qf, qdf = sim.get_state()
        sim.close()
        out["gpu" if comp else "gpu_plain"] = pack(it, ls, bad, qf, qdf)
    for comp in ((True, False) if tensor_free else ()):
        qc, qdc = np.ascontiguousarray(q.copy()), np.ascontiguousarray(qd.copy())
        it, ls, bad = (np.zeros((K, B), dtype=np.int64) for _ in range(3))
        for s in range(K):
            o = orc.tensorfree_batch_step_bdf1(desc, qc, qdc, h, 1, nthreads=nthreads, tol=tol, compensated=comp)
            it[s], ls[s], bad[s] = o["newton_iters"], o["ls_halvings"], o["status"] & 15
        out["tensor_free" if comp else "tensor_free_plain"] = pack(it, ls, bad, qc, qdc)
    if literal:
        orc.set_newton(tol=tol)
        qc, qdc = np.ascontiguousarray(q.copy()), np.ascontiguousarray(qd.copy())
        it, ls, bad = (np.zeros((K, B), dtype=np.int64) for _ in range(3))
        for s in range(K):
            o = orc.batch_step_bdf1(desc, qc, qdc, h, 1, nthreads=nthreads, counters=True)
            it[s], ls[s], bad[s] = o["newton_iters"], o["ls_halvings"], o["bad"]
        orc.set_newton()
        out["literal"] = pack(it, ls, bad, qc, qdc)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rollouts", type=int, default=64)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--tol", type=float, default=1e-9)
    ap.add_argument("--no-gpu", action="store_true")
    ap.add_argument("--no-literal", action="store_true")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    r = run_stats(a.rollouts, a.steps, tol=a.tol, gpu=not a.no_gpu, literal=not a.no_literal)
    keys = ("iters_per_step", "halvings_per_step", "bad_frac", "bad_frac_first_half", "bad_frac_second_half", "rollouts_with_a_bad_step")
    for name, v in r.items():
        print("%-12s " % name + "  ".join("%s=%.4g" % (k, v[k]) for k in keys))
    names = list(r)
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            a_, b_ = r[names[i]], r[names[j]]
            eq = np.linalg.norm(a_["q"] - b_["q"], axis=1) / np.linalg.norm(b_["q"], axis=1)
            print("%s vs %s: final q rel err max %.3g median %.3g; per-step bad-set agreement %.4f" % (
                names[i], names[j], eq.max(), np.median(eq), float(((a_["bad"] != 0) == (b_["bad"] != 0)).mean())))
    if a.json:
        json.dump({n: {k: v[k] for k in keys} for n, v in r.items()}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
