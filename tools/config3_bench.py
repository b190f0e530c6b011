"""BASELINE.json configs[2] on one GPU's share: 64-joint branching tree (revolute/prismatic), BDF1, B=512."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from redmax_amd import BatchSim, sceneTree  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sc = sceneTree(64)
sc.init()
q0, qd0 = sc.getQ()
q = np.empty((B, sc.nr))
qd = np.empty((B, sc.nr))
for b in range(B):
    rng = np.random.default_rng(20240 + b)
    q[b] = q0 + rng.uniform(-0.05, 0.05, sc.nr)
    qd[b] = rng.uniform(-0.1, 0.1, sc.nr)
for tol in (1e-9, 1e-8):
    sim = BatchSim(sc, batch=B)
    sim.opts.tol = tol
    sim.set_state(q, qd)
    sim.step_bdf1(5, h=1e-2)
    out = sim.step_bdf1(100, h=1e-2, stats=True)
    it = out["newton_iters"]
    st = out["status"]
    print("tree64 B=%d tol=%.0e: %.2f ms per 100 steps -> %.2f M rollout-steps/s; iters/step mean %.2f max-traj %.2f; ls/step %.3f; diverged %d maxiter %d pivoted %d" % (
        B, tol, out["ms"], B * 100 / out["ms"] / 1e3, it.mean() / 100, it.max() / 100, out["ls_halvings"].mean() / 100,
        int(((st & 1) != 0).sum()), int(((st & 2) != 0).sum()), int(((st & 16) != 0).sum())))
    qf, _ = sim.get_state()
    print("   finite:", bool(np.isfinite(qf).all()), " max|q| %.2f" % np.abs(qf).max())
