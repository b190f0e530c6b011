"""BASELINE.json configs[4]: 32-link chain over frictional ground (ForceGroundCuboid on every body), BDF2, B=1024.
The chain starts near horizontal 2 units above the ground and falls onto it: steps 1..~100 are free flight (the contact
code is skipped wave-uniformly), later steps run through impact and stick/slip sliding."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redmax_amd import BatchSim, sceneChainGround, syntheticStates  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sc = sceneChainGround(32)
sc.init()
q, qd = syntheticStates(sc.nr, B, sq=5e-4, sv=0.1)   # tip deflection ~0.3: every chain starts above the ground
q[0], qd[0] = sc.getQ()                                    # trajectory 0: the scene's own initial state
for tol in (1e-9, 1e-8):
    sim = BatchSim(sc, batch=B)
    sim.opts.tol = tol
    sim.set_state(q, qd)
    for seg in range(4):
        out = sim.step_bdf2(100, h=sc.h, stats=True, history=True)
        it, st = out["newton_iters"], out["status"]
        print("chain32+ground B=%d tol=%.0e steps %3d-%3d: %.2f ms per 100 steps -> %.2f M rollout-steps/s; iters/step mean %.2f "
              "max-traj %.2f; ls/step %.3f; diverged %d maxiter %d pivoted %d; V range %.3g" % (
                  B, tol, 100 * seg, 100 * seg + 99, out["ms"], B * 100 / out["ms"] / 1e3, it.mean() / 100, it.max() / 100,
                  out["ls_halvings"].mean() / 100, int(((st & 1) != 0).sum()), int(((st & 2) != 0).sum()),
                  int(((st & 16) != 0).sum()), out["V"].max() - out["V"].min()))
        sim.stats_reset()
    qf, _ = sim.get_state()
    print("   finite:", bool(np.isfinite(qf).all()), " max|q| %.2f" % np.abs(qf).max())
    sim.close()
