"""Cost per Newton iteration of the four step kernels of the 32-link chain in free flight (no corner touches the ground):
plain / contact-capable (CT) instantiation x BDF1 / BDF2.  Identical rollouts (one state copied 1024 times)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redmax_amd import BatchSim, sceneChain, sceneChainGround, syntheticStates  # noqa: E402

B, K = 1024, 100
for name, sc in (("plain", sceneChain(32)), ("CT", sceneChainGround(32))):
    sc.init()
    q, qd = syntheticStates(sc.nr, B, sq=5e-4, sv=0.1)
    q[:], qd[:] = 1e-3 * np.sin(np.arange(sc.nr)), 0.05 * np.cos(np.arange(sc.nr))   # near horizontal: no corner reaches z = -2 in 100 steps
    for integ in ("bdf1", "bdf2"):
        sim = BatchSim(sc, batch=B)
        sim.opts.tol = 1e-8
        sim.set_state(q, qd)
        for rep in range(2):
            sim.stats_reset()
            sim.set_state(q, qd)
            o = (sim.step_bdf1 if integ == "bdf1" else sim.step_bdf2)(K, h=5e-4, stats=True)
        it, ls = o["newton_iters"][0], o["ls_halvings"][0]
        print("%-5s %s: %.2f ms per %d steps, %d Newton iterations, %d halvings -> %.1f us per iteration" % (name, integ, o["ms"], K, it, ls, o["ms"] * 1e3 / it))
        sim.close()
