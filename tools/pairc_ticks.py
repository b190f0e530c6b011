"""Per-rollout shader-clock ticks (rmx_step_ticks) and Newton counts of the one-point and the two-point kernel of the full 32-link chain on
the bench states: where a launch's time is decided (the slowest wavefront) and what the two-point kernel gains per rollout.
Usage: pairc_ticks.py [B] [K]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redmax_amd import BatchSim, sceneChain, syntheticStates  # noqa: E402


def run(B, K, pair):
    os.environ["RMX_PAIRC"] = "1" if pair else "0"
    os.environ["RMX_W2_MAX"] = "0"
    sc = sceneChain(32)
    sc.init()
    q, qd = syntheticStates(sc.nr, B)
    sim = BatchSim(sc, batch=B)
    sim.set_state(q, qd)
    sim.step_bdf1(5, h=1e-2)
    q0, qd0 = sim.get_state()
    best = None
    for _ in range(3):
        sim.set_state(q0, qd0)
        out = sim.step_bdf1(K, h=1e-2, stats=True)
        t = sim.step_ticks().astype(np.float64)
        if best is None or out["ms"] < best[0]:
            best = (out["ms"], t, out["newton_iters"].copy())
    sim.close()
    return best


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    (m1, t1, it1), (m2, t2, it2) = run(B, K, False), run(B, K, True)
    for name, m, t, it in (("one point", m1, t1, it1), ("two points", m2, t2, it2)):
        print("%-10s kernel %.4f ms; ticks per rollout: p50 %.0f p90 %.0f p99 %.0f max %.0f (rollout %d), rollout 0: %.0f; ticks per Newton iteration: "
              "median %.1f, rollout 0 %.1f; iterations per step: mean %.3f, rollout 0 %.3f" %
              (name, m, np.percentile(t, 50), np.percentile(t, 90), np.percentile(t, 99), t.max(), int(t.argmax()), t[0], np.median(t / it), t[0] / it[0],
               it.mean() / K, it[0] / K))
    r = t1 / t2
    print("ticks one point / two points per rollout: median x%.4f, min x%.4f, max x%.4f, rollout 0 x%.4f; slowest rollout x%.4f" %
          (np.median(r), r.min(), r.max(), r[0], t1.max() / t2.max()))


if __name__ == "__main__":
    main()
