"""Distribution of Newton work per trajectory for the bench workload (run on the GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redmax_amd import BatchSim, sceneChain, syntheticStates  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sc = sceneChain(32)
sc.init()
q, qd = syntheticStates(32, B)
sim = BatchSim(sc, batch=B)
sim.opts.tol = 1e-8
sim.set_state(q, qd)
sim.step_bdf1(10, h=1e-2)
out = sim.step_bdf1(100, h=1e-2, stats=True)
it = out["newton_iters"]
print("kernel ms %.3f; iters/traj mean %.1f min %d max %d p99 %.0f; pivoted-fallback trajectories %d" % (
    out["ms"], it.mean(), it.min(), it.max(), np.percentile(it, 99), int(((out["status"] & 16) != 0).sum())))
print("us per iteration of the slowest trajectory: %.2f ; of the mean: %.2f" % (1e3 * out["ms"] / it.max(), 1e3 * out["ms"] / it.mean()))
for Bb in (256, 512, 2048, 4096):
    if Bb > B:
        qq, qqd = syntheticStates(32, Bb)
    else:
        qq, qqd = q[:Bb], qd[:Bb]
    s2 = BatchSim(sc, batch=Bb)
    s2.opts.tol = 1e-8
    s2.set_state(qq, qqd)
    s2.step_bdf1(10, h=1e-2)
    o2 = s2.step_bdf1(100, h=1e-2)
    print("B=%d: %.3f ms per 100 steps -> %.2f M rollout-steps/s" % (Bb, o2["ms"], Bb * 100 / o2["ms"] / 1e3))
