"""The two-wave kernels for full 64-node trees (rmx_kernels.hip RMX_PART 5) against the one-wave kernels: same bits, time per 100 steps.
RMX_W2_MAX is read when a model is created, so the two runs live in one process.  Usage: w2_check.py [batch ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redmax_amd import _abi  # noqa: E402
if os.environ.get("RMX_W2_LIB"):          # a variant library (tools/build_variant.py) instead of the in-tree one
    _abi.LIB_PATH = os.environ["RMX_W2_LIB"]
from redmax_amd import BatchSim, sceneTree  # noqa: E402


def run(B, w2, integ="bdf1", tol=1e-9):
    os.environ["RMX_W2_MAX"] = "100000" if w2 else "0"
    sc = sceneTree(64)
    sc.init()
    q0, _ = sc.getQ()
    q = np.empty((B, sc.nr))
    qd = np.empty((B, sc.nr))
    for b in range(B):
        rng = np.random.default_rng(20240 + b)
        q[b] = q0 + rng.uniform(-0.05, 0.05, sc.nr)
        qd[b] = rng.uniform(-0.1, 0.1, sc.nr)
    sim = BatchSim(sc, batch=B)
    sim.opts.tol = tol
    sim.set_state(q, qd)
    step = sim.step_bdf1 if integ == "bdf1" else sim.step_bdf2
    step(5, h=1e-2)
    best = None
    for _ in range(3):
        out = step(100, h=1e-2, stats=True)
        best = out["ms"] if best is None else min(best, out["ms"])
    qf, qdf = sim.get_state()
    return best, qf, qdf, out["newton_iters"].copy(), out["status"].copy()


def main():
    batches = [int(a) for a in sys.argv[1:]] or [64, 256, 512]
    for integ in os.environ.get("RMX_W2_INTEG", "bdf1 bdf2").split():
        for B in batches:
            t1, q1, qd1, it1, st1 = run(B, False, integ)
            t2, q2, qd2, it2, st2 = run(B, True, integ)
            same = np.array_equal(q1, q2) and np.array_equal(qd1, qd2) and np.array_equal(it1, it2) and np.array_equal(st1, st2)
            print("tree64 %s B=%d: one wave %.3f ms, two waves %.3f ms per 100 steps (x%.3f); bit-identical %s; pivoted %d"
                  % (integ, B, t1, t2, t1 / t2, same, int(((st1 & 16) != 0).sum())), flush=True)
            if not same:
                print("   max|dq| %.3e  iters differ %d" % (np.abs(q1 - q2).max(), int((it1 != it2).sum())))


if __name__ == "__main__":
    main()
