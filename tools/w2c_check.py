"""The two-wave kernel of the full 32-link chain (rmx_kernels.hip RMX_PART 6: shards of 128 .. 512 rollouts) against the one-wave headline
kernel: time per 100 steps, agreement of the final states, Newton counts; and the same with the run-ahead switched off (same bits).
RMX_W2_MAX is read when a model is created.  Usage: w2c_check.py [batch ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redmax_amd import _abi  # noqa: E402
if os.environ.get("RMX_W2_LIB"):          # a variant library (tools/build_variant.py) instead of the in-tree one
    _abi.LIB_PATH = os.environ["RMX_W2_LIB"]
from redmax_amd import BatchSim, sceneChain, syntheticStates  # noqa: E402


def run(B, w2, ahead="1"):
    os.environ["RMX_W2_MAX"] = "100000" if w2 else "0"
    os.environ["RMX_W2_RUNAHEAD"] = ahead
    sc = sceneChain(32)
    sc.init()
    q, qd = syntheticStates(sc.nr, B)
    sim = BatchSim(sc, batch=B)
    sim.set_state(q, qd)
    sim.step_bdf1(5, h=1e-2)
    best = None
    for _ in range(3):
        out = sim.step_bdf1(100, h=1e-2, stats=True)
        best = out["ms"] if best is None else min(best, out["ms"])
    qf, qdf = sim.get_state()
    return best, qf, qdf, out["newton_iters"].copy(), out["status"].copy()


def main():
    for B in [int(a) for a in sys.argv[1:]] or [128, 256, 512]:
        t1, q1, qd1, it1, st1 = run(B, False)
        t2, q2, qd2, it2, st2 = run(B, True)
        t3, q3, qd3, it3, st3 = run(B, True, "0")
        print("chain32 B=%d: one wave %.3f ms, two waves %.3f ms per 100 steps (x%.3f; run-ahead off %.3f ms, same bits %s); max|dq| vs one wave %.2e, "
              "Newton iterations %d vs %d (rollouts with another count: %d), status equal %s"
              % (B, t1, t2, t1 / t2, t3, np.array_equal(q2, q3) and np.array_equal(it2, it3), np.abs(q1 - q2).max(), it1.sum(), it2.sum(),
                 int((it1 != it2).sum()), np.array_equal(st1, st2)), flush=True)


if __name__ == "__main__":
    main()
