"""configs[3] (16-DOF chain, 512 rollouts, 20 steps): the adjoint launch pair with a second wavefront per rollout for M, D (RMX_PART 8, the
default for batches of up to one rollout per two SIMDs) against one wavefront per rollout (RMX_ADJ_HELP=0): P, dP/dp, counters bit for
bit, kernel milliseconds; BDF1 and BDF2, a second batch size and a smaller tree.
    python tools/adj_help_check.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redmax_amd import BatchSim  # noqa: E402
from redmax_amd.scenes import sceneAdjointChain  # noqa: E402


def run(n, B, K, bdf2, helper, reps=9):
    os.environ["RMX_ADJ_HELP"] = "1" if helper else "0"
    sc = sceneAdjointChain(n)
    sc.init()
    p = 0.1 * np.random.default_rng(0).standard_normal((B, sc.nr))
    sim = BatchSim(sc, batch=B)
    q0, qd0 = sc.getQ()
    ms = []
    for _ in range(reps):
        sim.set_state(q0[None, :], qd0[None, :])
        f = sim.adjoint_bdf2 if bdf2 else sim.adjoint_bdf1
        P, dPdp, info = f(K, sc.h, dict(sc.task, t=K * sc.h), p, stats=True)
        ms.append(info["ms"])
    sim.close()
    return P, dPdp, info, min(ms), float(np.median(ms))


def main():
    for n, B, K, bdf2 in ((16, 512, 20, False), (16, 512, 20, True), (16, 200, 33, False), (10, 512, 20, False), (16, 16, 7, True)):
        a = run(n, B, K, bdf2, False)
        b = run(n, B, K, bdf2, True)
        same = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2]["newton_iters"], b[2]["newton_iters"]) and \
            np.array_equal(a[2]["status"], b[2]["status"])
        print("n=%2d B=%3d K=%2d %s: same bits %s; kernel ms one wave min %.4f median %.4f, with the helper wave min %.4f median %.4f (x%.3f)" %
              (n, B, K, "BDF2" if bdf2 else "BDF1", same, a[3], a[4], b[3], b[4], a[4] / b[4]), flush=True)


if __name__ == "__main__":
    main()
