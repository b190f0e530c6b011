"""Repeatable kernel timing of the bench workload (min / median of R launches of 100 steps after 10 warm-up steps)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redmax_amd import BatchSim, sceneChain, syntheticStates  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 7
sc = sceneChain(32)
sc.init()
q, qd = syntheticStates(32, 1024)
sim = BatchSim(sc, batch=1024)
sim.opts.tol = float(os.environ.get("RMX_TOL", "1e-9"))
ms = []
for r in range(R):
    sim.set_state(q, qd)
    sim.step_bdf1(10, h=1e-2)
    ms.append(sim.step_bdf1(100, h=1e-2)["ms"])
ms = np.array(ms)
print("kernel ms per 100 steps: min %.3f median %.3f max %.3f  -> %.2f M rollout-steps/s (median)" % (ms.min(), np.median(ms), ms.max(), 102.4 / np.median(ms)))
