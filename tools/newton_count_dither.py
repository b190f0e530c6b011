"""Can ANY evaluator agree with the literal oracle's Newton counts on >= 99 % of trajectory-steps at the reference's tol = 1e-9
?  The literal CPU oracle against ITSELF: the same rollouts of the headline workload, once from the synthetic initial
states and once from those states moved by one unit in the last place of every coordinate (a perturbation below anything a different -
equally valid - summation order introduces).  The fraction of (rollout, step) pairs with equal Newton iteration counts is the ceiling
for any other implementation of the same mathematics.  CPU only (oracle/): run here or on the GPU box.
    python tools/newton_count_dither.py [rollouts] [steps] [tol ...]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from redmax_amd import sceneChain, syntheticStates  # noqa: E402


def counts(desc, q, qd, h, K, tol, threads):
    orc.set_newton(tol=tol)
    qc, qdc = np.ascontiguousarray(q.copy()), np.ascontiguousarray(qd.copy())
    per = np.zeros((K, q.shape[0]), dtype=np.int64)
    for s in range(K):
        per[s] = orc.batch_step_bdf1(desc, qc, qdc, h, 1, nthreads=threads, counters=True)["newton_iters"]
    orc.set_newton()
    return per, qc


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 23
    tols = [float(a) for a in sys.argv[3:]] or [1e-9, 1e-8]
    sc = sceneChain(32)
    sc.init()
    desc = sc.desc()
    q, qd = syntheticStates(sc.nr, nb)
    rng = np.random.default_rng(7)
    sgn = rng.choice([-1.0, 1.0], size=q.shape)
    q1 = np.nextafter(q, q + sgn)                     # one unit in the last place, random direction per coordinate
    threads = min(os.cpu_count() or 1, nb)
    out = {"workload": "32-link chain, BDF1, h = 1e-2, synthetic states U(-0.1, 0.1)", "rollouts": nb, "steps": K, "perturbation": "1 ulp of every q", "tols": {}}
    for tol in tols:
        a, qa = counts(desc, q, qd, sc.h, K, tol, threads)
        b, qb = counts(desc, q1, qd, sc.h, K, tol, threads)
        same = a == b
        out["tols"]["%g" % tol] = {"trajectory_steps": int(a.size), "equal_counts": int(same.sum()), "frac": round(float(same.mean()), 5),
                                    "iters_a": int(a.sum()), "iters_b": int(b.sum()), "max_abs_diff_in_a_step": int(np.abs(a - b).max()),
                                    "final_q_relerr_max": float(np.max(np.linalg.norm(qa - qb, axis=1) / np.linalg.norm(qa, axis=1)))}
        print("tol %g: literal oracle vs itself from states 1 ulp apart: equal Newton counts on %d of %d trajectory-steps (%.2f %%); iterations %d vs %d; "
              "final q differs by %.1e" % (tol, same.sum(), a.size, 100 * same.mean(), a.sum(), b.sum(), out["tols"]["%g" % tol]["final_q_relerr_max"]), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
